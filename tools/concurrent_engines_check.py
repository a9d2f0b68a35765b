import sys, threading, numpy as np
sys.path.insert(0, '/root/repo')
from summertts_amd import engine, synth_blob as sb
cfg = sb.full_cfg("hifigan_sdp"); blob = sb.make_blob(cfg, 1234); ids = sb.synthetic_ids(96, cfg.vocab, salt=2)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
engines = [engine.Synthesizer(blob) for _ in range(3)]
for e in engines:
    e.debug_set("front_mode", mode); e.set_profiling(True)
want = engines[0].infer_ids(ids, 0, 1.0)
bad = []
def work(k):
    for it in range(20):
        o = engines[k].infer_ids(ids, 0, 1.0)
        if not np.array_equal(o, want):
            bad.append((k, it, int(np.abs(o.astype(np.int32) - want.astype(np.int32)).max()), int((o != want).sum()), engines[k].profile()["conv_math_fallbacks"]))
th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
print("front_mode", mode, "mismatches", bad, "fallbacks", [e.profile()["conv_math_fallbacks"] for e in engines], flush=True)
