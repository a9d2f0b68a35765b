#!/usr/bin/env python3
"""A/B of the 128-channel decoder stage as grouped launches (trunk_mode 1) vs one persistent launch (trunk_mode 2): per-step decoder
time, launches, and the PCM difference.   python tools/trunk_mode_ab.py [phonemes] [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from summertts_amd import engine, synth_blob as sb
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg = sb.full_cfg("hifigan_sdp")
blob = sb.make_blob(cfg, 1234)
syn = engine.Synthesizer(blob)
ids = sb.synthetic_ids(T, cfg.vocab, salt=0)
syn.set_profiling(True)
pcm = {}
for rnd in range(2):
    for mode in (1, 2):
        syn.set_conv_math("bf16x3")
        syn.debug_set("trunk_mode", mode)
        for _ in range(3):
            syn.run_batch([ids])
        dec, mf, n = 0.0, 0.0, 0
        t0 = time.perf_counter()
        for _ in range(reps):
            syn.run_batch([ids])
            p = syn.profile()
            dec += p["ms_decoder"]; mf += p["ms_decoder_mfma"]; n = p["decoder_mfma_launches"]
        el = (time.perf_counter() - t0) / reps * 1e3
        pcm[mode] = syn.pcm_host().copy()
        print(f"trunk_mode {mode}: {el:.3f} ms/step  decoder {dec / reps:.3f} ms  trunk {mf / reps:.3f} ms  matrix-core launches {n}", flush=True)
print("PCM identical:", bool(np.array_equal(pcm[1], pcm[2])), " max |diff| LSB:", int(np.abs(pcm[1].astype(int) - pcm[2].astype(int)).max()))
syn.close()
