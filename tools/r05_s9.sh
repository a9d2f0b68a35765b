# session 8: ResBlock chains of a decoder stage as two launch sequences on two streams (heaviest chain | the others): parity, per-stage masks, A/B
O=gpurun_out/r05s9
mkdir -p $O
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
for rep in 1 2; do
  for m in 0 5 65 37 21 13; do
    timeout 200 python bench.py $Q --debug-set chain_split=$m 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c1 chain_split=$m', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  done
done
for c in 4; do for m in 0 5; do
  timeout 300 python bench.py $Q --config $c --steps 6 --warmup 2 --debug-set chain_split=$m 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c$c chain_split=$m', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4))" >> $O/ab.txt
done; done
cat $O/ab.txt
python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from conftest import golden_files_v2, load_golden_v2
from summertts_amd import engine
for name in ("full_hifigan_sdp_T128.npz", "full_mbb_fix_T96.npz", "real_hifigan_sdp_T96.npz"):
    path = [p for p in golden_files_v2("full_") + golden_files_v2("real_") if p.endswith(name)][0]
    g, cfg, blob, utts, stride = load_golden_v2(path)
    syn = engine.Synthesizer(blob)
    u, ids, sid, ls, dur, pcm_ref, wave = utts[0]
    base = syn.infer_ids(ids, sid, ls)
    for m in (5, 65, 37):
        syn.debug_set("chain_split", m)
        got = [syn.infer_ids(ids, sid, ls) for _ in range(3)]
        print(name, "mask", m, "identical to grouped launches:", all(np.array_equal(x, base) for x in got), "max lsb vs reference", int(np.abs(got[0].astype(int) - pcm_ref.astype(int)).max()))
    syn.close()
PY
