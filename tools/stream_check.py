"""Streaming decode on the full-size synthetic model: bit-exactness against the one-pass PCM and time to
first audio per chunk size (run on a GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summertts_amd import engine as eng, synth_blob as sb

def main():
    for kind in sys.argv[1:] or ["hifigan_sdp", "mbb_fix"]:
        cfg = sb.full_cfg(kind)
        syn = eng.Synthesizer(sb.make_blob(cfg, 1234))
        syn.set_conv_mode(int(os.environ.get("STREAM_CONV_MODE", "0")))
        ids = sb.synthetic_ids(128, cfg.vocab, salt=0)
        full = syn.infer_ids(ids)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); syn.infer_ids(ids); t.append(time.perf_counter() - t0)
        print(f"{kind}: halo {syn.stream_halo_frames()} frames, {len(full)} samples, one-pass latency {1e3*np.median(t):.2f} ms")
        for chunk in (16, 32, 64, 128, 256):
            syn.infer_ids_stream(ids, chunk)
            best = None
            for _ in range(5):
                chunks, times = syn.infer_ids_stream(ids, chunk)
                if best is None or times[0] < best[0]: best = (times[0], times[-1], len(chunks))
            got = np.concatenate(chunks)
            d = np.abs(got.astype(np.int32) - full.astype(np.int32))
            print(f"   chunk {chunk:4d} frames: {best[2]:3d} chunks, first audio {1e3*best[0]:.2f} ms, last {1e3*best[1]:.2f} ms, "
                  f"max |diff| {int(d.max())} LSB on {int((d > 0).sum())} samples")

if __name__ == "__main__":
    main()
