"""Actual error of the GPU path against the compiled reference at a realistic size (full-size HiFi-GAN model, 72
phonemes) under each arithmetic setting of the trunk convs (sts_set_conv_math): split-bf16 (default), exact-fp32 MFMA with
the Winograd-domain layer kernels, split-bf16 in every eligible conv -- and, with STS_NO_WINO=1 in the environment (read
once per process), the exact-fp32 direct-form layer kernels:   python tools/err_check.py; STS_NO_WINO=1 python tools/err_check.py"""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref
from summertts_amd import engine, synth_blob as sb

cfg = sb.full_cfg("hifigan_sdp"); blob = sb.make_blob(cfg, 1234)
ids = sb.synthetic_ids(72, cfg.vocab, salt=3)
o = pyref.RefModel(blob).infer_ids(ids, 0, 1.0)
syn = engine.Synthesizer(blob); syn.set_record_taps(True)
for math in ("bf16x3", "f32", "bf16x3_all", "f16x2"):
    syn.set_conv_math(math); syn.run_batch([ids])
    w = syn.tap("wave")[0]; d = w - o["wave"]
    p = syn.pcm_host().astype(np.int32) - o["pcm"].astype(np.int32)
    label = math + (" (direct-form fp32 layers)" if math == "f32" and os.environ.get("STS_NO_WINO") else "")
    print("%-12s wave rmse %.2e max %.2e (ref rms %.3f) | pcm max %d LSB on %d of %d | durations equal: %s" % (
        label, np.sqrt((d * d).mean()), np.abs(d).max(), o["wave"].std(), np.abs(p).max(), (p != 0).sum(), p.size,
        bool((syn.durations(len(ids)) == o["durations"]).all())))
