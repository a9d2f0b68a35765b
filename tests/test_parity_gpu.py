"""GPU suite (MI355X): the HIP path against the oracle, through the C ABI."""
import numpy as np
import pytest

from conftest import (TAP_MAXABS_TOL, AMP_WAVE_MAXABS_TOL, AMP_WAVE_RMSE_TOL, assert_pcm_close, assert_pcm_close_wrapped, assert_wave_close,
                      golden_files, golden_files_v2, load_golden, load_golden_v2)
from oracle import pyref
from summertts_amd import engine, synth_blob as sb

pytestmark = pytest.mark.gpu

TINY = ["hifigan_sdp", "hifigan_fix", "mbb_fix", "ms_sdp", "istft_fix", "ms_hifigan_sdp", "ms_hifigan_fix", "odd"]

CONV_CASES = [  # Cin, Cout, k, pad, dil, L, stride_transposed, depthwise
    (64, 64, 3, 1, 1, 300, 0, False), (32, 32, 11, 25, 5, 700, 0, False), (96, 192, 1, 0, 1, 77, 0, False),
    (64, 128, 5, 2, 1, 129, 0, False), (48, 40, 7, 3, 1, 200, 0, False), (20, 24, 3, 3, 3, 50, 0, False),
    (64, 32, 8, 2, 1, 100, 4, False), (128, 64, 16, 4, 1, 65, 8, False), (24, 12, 7, 2, 1, 33, 3, False),
    (16, 16, 3, 9, 9, 120, 0, True), (64, 72, 7, 3, 1, 500, 0, False), (256, 256, 3, 1, 1, 1000, 0, False),
    (32, 1, 7, 3, 1, 1000, 0, False), (4, 1, 63, 31, 1, 400, 0, False), (192, 384, 5, 2, 1, 31, 0, False),
    (128, 128, 7, 9, 3, 1500, 0, False), (128, 128, 11, 25, 5, 777, 0, False), (64, 64, 9, 8, 2, 241, 0, False),
    (256, 256, 13, 6, 1, 480, 0, False), (32, 32, 3, 6, 6, 239, 0, False),
    # long single-output-channel FIRs (conv_cout1_kernel: 16-byte staging of whole tiles, scalar last tile and halo columns)
    (32, 1, 7, 3, 1, 5003, 0, False), (20, 1, 5, 2, 1, 4500, 0, False), (8, 1, 7, 0, 1, 4700, 0, False),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=str)
def test_conv_kernels_against_torch_fp32(case):
    import torch
    import torch.nn.functional as F
    ci, co, k, pad, dil, L, st, dw = case
    rng = np.random.default_rng(ci * 131 + co)
    x = rng.standard_normal((ci, L)).astype(np.float32)
    w = (rng.standard_normal((co, k, 1 if dw else ci)) / np.sqrt(k * (1 if dw else ci))).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    xt = torch.from_numpy(x)[None]
    if st:
        ref = F.conv_transpose1d(xt, torch.from_numpy(w).permute(2, 0, 1).contiguous(), torch.from_numpy(b), stride=st, padding=pad)[0].numpy()
    elif dw:
        ref = F.conv1d(xt, torch.from_numpy(w).permute(0, 2, 1).contiguous(), torch.from_numpy(b), padding=pad, dilation=dil, groups=ci)[0].numpy()
    else:
        ref = F.conv1d(xt, torch.from_numpy(w).permute(0, 2, 1).contiguous(), torch.from_numpy(b), padding=pad, dilation=dil)[0].numpy()
    mfma_ok = not dw and ci >= 32 and co >= 32
    modes = [1] + ([0, 2, 3, 4, 5, 6, 7, 8, 9] if mfma_ok else [0])
    outs = {}
    for mode in modes:
        y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=mode)
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5, (case, mode, np.abs(y - ref).max())   # fp32, K <= 2816 terms
        outs[mode] = y
    if mfma_ok:   # every LDS-staged matrix-core tile shape walks K in the same order: bit-identical results
        for mode in (3, 4, 5, 6, 7):
            assert np.array_equal(outs[mode], outs[2]), (case, mode)
    # Winograd-domain kernel (segmented F(2,3)): same tolerance as the direct kernels
    if mfma_ok and not st and k >= 2 and dil in (1, 2, 3, 5, 6) and (k - 1) * dil <= 64:
        y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=12)
        assert y.shape == ref.shape
        assert np.abs(y - ref).max() <= 2e-5, (case, "winograd", np.abs(y - ref).max())
        y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, in_slope=0.1, in_act=1, mode=12)
        assert np.abs(y - engine.debug_conv1d(x, w, b, pad, dil, st, dw, in_slope=0.1, in_act=1, mode=0)).max() <= 2e-5
    # fused input leaky-relu
    y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, in_slope=0.1, in_act=1, mode=0)
    y1 = engine.debug_conv1d(np.where(x < 0, x * np.float32(0.1), x).astype(np.float32), w, b, pad, dil, st, dw, mode=0)
    assert np.array_equal(y, y1)


CONV_MATHS = ["bf16x3", "f32", "bf16x3_all", "f16x2"]   # sts_set_conv_math 0 (default) / 1 / 2 / 3: same tolerances for all four

BF3_CASES = [c for c in CONV_CASES if not c[7] and c[0] >= 32 and c[0] % 16 == 0 and c[1] >= 32] + [
    (512, 256, 16, 4, 1, 70, 8, False), (128, 96, 3, 1, 1, 4000, 0, False), (32, 32, 7, 9, 3, 5000, 0, False)]


@pytest.mark.parametrize("case", BF3_CASES, ids=str)
def test_bf3_conv_is_as_accurate_as_the_fp32_matrix_core_kernel(case):
    """Split-bf16 conv (conv_bf3.hip: fp32 operands as 3 bf16 terms, 6 products, fp32 accumulate) against a float64
    convolution: same 2e-5 bound as the fp32 kernels, and an RMS error no worse than 1.5x the exact-fp32 MFMA kernel's."""
    import torch
    import torch.nn.functional as F
    ci, co, k, pad, dil, L, st, dw = case
    rng = np.random.default_rng(ci * 977 + co + k)
    x = (rng.standard_normal((ci, L)) * rng.uniform(0.05, 3.0, (ci, 1))).astype(np.float32)
    w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    xt = torch.from_numpy(x).double()[None]
    if st:
        ref = F.conv_transpose1d(xt, torch.from_numpy(w).double().permute(2, 0, 1).contiguous(), torch.from_numpy(b).double(), stride=st, padding=pad)[0].numpy()
    else:
        ref = F.conv1d(xt, torch.from_numpy(w).double().permute(0, 2, 1).contiguous(), torch.from_numpy(b).double(), padding=pad, dilation=dil)[0].numpy()
    y32 = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=2 + 4)          # exact-fp32 MFMA kernel, 32 x 128 tile
    e32 = np.sqrt(np.mean((y32 - ref) ** 2))
    outs = []
    # mode = 20 + tile code (13: automatic).  The shipped library carries the tiles its automatic choice uses -- 0 / 3 / 4 and, for
    # transposed convs, the phase-merged 22 / 23 --; the lab build (-DSTS_EXPERIMENTS) every code.  41-43: phase-merged rows
    modes = (13, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 41, 42, 43) if engine.lab_build() else (13, 20, 23, 24, 42, 43)
    for mode in modes:
        y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=mode)
        assert y.shape == ref.shape
        err = np.abs(y - ref)
        assert err.max() <= 2e-5, (case, mode, err.max())
        assert np.sqrt(np.mean(err ** 2)) <= 1.5 * e32 + 1e-9, (case, mode, np.sqrt(np.mean(err ** 2)), e32)
        outs.append(y)
    for y in outs[2:]:     # every explicit tile shape walks K in the same order (outs[0] = automatic choice: may split K)
        assert np.array_equal(y, outs[1]), case
    # K split over two wave groups of a workgroup (tile code 20): different summation order, same bounds
    y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=40)
    err = np.abs(y - ref)
    assert err.max() <= 2e-5 and np.sqrt(np.mean(err ** 2)) <= 1.5 * e32 + 1e-9, (case, "kg2", err.max())
    # fused input leaky-relu is applied before the split
    y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, in_slope=0.1, in_act=1, mode=13)
    y1 = engine.debug_conv1d(np.where(x < 0, x * np.float32(0.1), x).astype(np.float32), w, b, pad, dil, st, dw, mode=13)
    assert np.array_equal(y, y1)


H2_MODES = (50, 60, 63, 64, 80, 82, 83)     # two-term fp16 form: automatic tile, tile codes 0 / 3 / 4 / 20 (K groups) / 22 / 23 (phase-merged)


@pytest.mark.parametrize("case", BF3_CASES, ids=str)
def test_f16x2_conv_against_float64(case):
    """Two-term fp16 conv (conv_bf3.hip MATH 1: x = hi + 2^-11 lo', weights scaled per conv, three products) against a float64
    convolution: the 2e-5 bound of every other conv kernel, an RMS error within 2x the exact-fp32 MFMA kernel's, and no
    dependence on the operands' magnitude -- weights scaled by 2^-20 .. 2^9 and inputs by 2^-8 .. 2^10 give the scaled result
    to the same relative accuracy.  (Activations have an ABSOLUTE error floor instead, 2^-36 per value: a tensor that is
    entirely below ~1e-5 -- the 2^-20 case -- is only good to ~1e-5 relative.)"""
    import torch
    import torch.nn.functional as F
    ci, co, k, pad, dil, L, st, dw = case
    rng = np.random.default_rng(ci * 977 + co + k)
    x = (rng.standard_normal((ci, L)) * rng.uniform(0.05, 3.0, (ci, 1))).astype(np.float32)
    w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)

    def ref64(x, w, b):
        xt = torch.from_numpy(x).double()[None]
        if st:
            return F.conv_transpose1d(xt, torch.from_numpy(w).double().permute(2, 0, 1).contiguous(), torch.from_numpy(b).double(), stride=st, padding=pad)[0].numpy()
        return F.conv1d(xt, torch.from_numpy(w).double().permute(0, 2, 1).contiguous(), torch.from_numpy(b).double(), padding=pad, dilation=dil)[0].numpy()

    ref = ref64(x, w, b)
    y32 = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=2 + 4)
    e32 = np.sqrt(np.mean((y32 - ref) ** 2))
    outs = []
    for mode in H2_MODES:
        y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, mode=mode)
        assert y.shape == ref.shape
        err = np.abs(y - ref)
        assert err.max() <= 2e-5, (case, mode, err.max())
        assert np.sqrt(np.mean(err ** 2)) <= 2.0 * e32 + 1e-9, (case, mode, np.sqrt(np.mean(err ** 2)), e32)
        outs.append(y)
    # operand magnitudes: the result scales, its relative error does not
    zero = np.zeros_like(b)
    base = ref64(x, w, zero)
    rms = np.sqrt(np.mean(base ** 2))
    for sx, sw, tol in ((2.0 ** -8, 1.0, 1e-6), (1.0, 2.0 ** -20, 1e-6), (2.0 ** 10, 1.0, 1e-6), (2.0 ** -6, 2.0 ** 9, 1e-6), (2.0 ** 10, 2.0 ** -18, 1e-6),
                        (2.0 ** -20, 1.0, 3e-5)):
        y = engine.debug_conv1d((x * np.float32(sx)).astype(np.float32), (w * np.float32(sw)).astype(np.float32), zero, pad, dil, st, dw, mode=50)
        rel = np.sqrt(np.mean((y / (sx * sw) - base) ** 2)) / rms
        assert rel <= tol, (case, sx, sw, rel)
    # a conv of vanishing weights (max |w| ~ 2^-105: the per-conv scale is capped at 2^100) stays finite and accurate
    y = engine.debug_conv1d(x, (w * np.float32(2.0 ** -100)).astype(np.float32), zero, pad, dil, st, dw, mode=50)
    assert np.isfinite(y).all() and np.sqrt(np.mean((y.astype(np.float64) * 2.0 ** 100 - base) ** 2)) / rms <= 1e-5
    # fused input leaky-relu is applied before the split
    y = engine.debug_conv1d(x, w, b, pad, dil, st, dw, in_slope=0.1, in_act=1, mode=50)
    y1 = engine.debug_conv1d(np.where(x < 0, x * np.float32(0.1), x).astype(np.float32), w, b, pad, dil, st, dw, mode=50)
    assert np.array_equal(y, y1)


@pytest.mark.parametrize("stats", ["gaussian", "realistic"])
def test_f16x2_falls_back_when_an_activation_leaves_the_fp16_range(stats):
    """A model whose decoder activations exceed 65504 (conv_pre weights scaled up) trips the overflow word of the two-term fp16
    kernels; the engine repeats the call in the split-bf16 form, counts it, and returns exactly the split-bf16 result.  (Round 5: also on
    the checkpoint-like weight statistics -- per-channel gains, outliers, biases --, where the range logic had never been driven to its limit.)"""
    import dataclasses
    cfg = dataclasses.replace(sb.full_cfg("hifigan_sdp"), stats=stats)
    blob = sb.make_blob(cfg, 5)
    ids = sb.synthetic_ids(20, cfg.vocab)
    syn = engine.Synthesizer(blob)
    syn.set_profiling(True)
    syn.set_forced_durations([3] * len(ids))
    syn.set_conv_math("f16x2")
    syn.run_batch([ids])
    assert syn.profile()["conv_math_fallbacks"] == 0
    syn.close()
    # the decoder's input conv is the first conv after the text encoder and the generator header in the blob (synth_blob.make_blob)
    w = sb._W(5, stats)
    w.ints(cfg.is_ms, cfg.lang, cfg.dur_type, cfg.dec_type)
    sb._text_encoder(w, cfg)
    sb._gen_hdr(w, cfg)
    start = w.n + 6
    assert tuple(blob[w.n:w.n + 3].astype(int)) == (cfg.up_init, cfg.inter, 7)
    big = blob.copy()
    big[start:start + cfg.up_init * 7 * cfg.inter] *= np.float32(3.0e6)
    syn = engine.Synthesizer(big)
    syn.set_profiling(True)
    syn.set_forced_durations([3] * len(ids))
    syn.set_conv_math("bf16x3")
    syn.run_batch([ids])
    want = syn.pcm_host().copy()
    syn.set_conv_math("f16x2")
    syn.set_forced_durations([3] * len(ids))          # (one run consumes them -- and the repeat inside the call needs them again)
    syn.run_batch([ids])
    assert syn.profile()["conv_math_fallbacks"] == 1
    assert np.array_equal(syn.pcm_host(), want)
    syn.run_batch([ids])
    assert syn.profile()["conv_math_fallbacks"] == 2
    # two repeats in a row: the engine stops trying (no third repeat, same result) until the setting is made again
    syn.set_forced_durations([3] * len(ids))
    syn.run_batch([ids])
    assert syn.profile()["conv_math_fallbacks"] == 2
    assert np.array_equal(syn.pcm_host(), want)
    # streaming: the word is checked before the first chunk leaves -- the call starts over in the split-bf16 form
    syn.set_conv_math("bf16x3")
    syn.set_forced_durations([3] * len(ids))
    ref_chunks, _ = syn.infer_ids_stream(ids, 16)
    syn.set_conv_math("f16x2")
    syn.set_forced_durations([3] * len(ids))
    chunks, _ = syn.infer_ids_stream(ids, 16)          # (set_conv_math above re-armed the two-term form)
    assert syn.profile()["conv_math_fallbacks"] == 3
    assert np.array_equal(np.concatenate(chunks), np.concatenate(ref_chunks))
    syn.close()


def test_bf3_conv_random_shapes_against_float64():
    """Seeded sweep of odd shapes through the split-bf16 conv (automatic tile, the K-split tile, a phase-merged tile): output
    channels that are not a multiple of the tile, sequences shorter than one tile or than the halo, every dilation / kernel
    size the staged window admits, polyphase transposed convs with partial last phases -- against a float64 convolution."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(20260925)
    cases = []
    for _ in range(28):
        ci = int(rng.choice([32, 48, 64, 96, 128, 192, 256]))
        co = int(rng.integers(1, 300))
        k = int(rng.integers(1, 14))
        dil = int(rng.choice([1, 1, 2, 3, 5]))
        while (k - 1) * dil > 64:
            k -= 1
        L = int(rng.choice([1, 2, 7, 31, 33, 127, 129, 257, 700, 3001]))
        pad = int(rng.integers(0, (k - 1) * dil + 1))
        if L + 2 * pad - dil * (k - 1) <= 0:
            pad = dil * (k - 1)
        cases.append((ci, co, k, pad, dil, L, 0))
    for _ in range(10):
        ci = int(rng.choice([32, 64, 128, 256, 512]))
        co = int(rng.choice([8, 32, 40, 64, 128, 256]))
        st = int(rng.choice([2, 3, 4, 8]))
        k = int(st * rng.integers(1, 3) + rng.integers(0, st))
        L = int(rng.choice([1, 5, 64, 130, 669]))
        pad = int(rng.integers(0, max(1, (k - st) // 2 + 1)))
        if (L - 1) * st - 2 * pad + k <= 0:
            pad = 0
        cases.append((ci, co, k, pad, 1, L, st))
    for case in cases:
        ci, co, k, pad, dil, L, st = case
        x = (rng.standard_normal((ci, L)) * rng.uniform(0.05, 3.0, (ci, 1))).astype(np.float32)
        w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci)).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        xt = torch.from_numpy(np.where(x < 0, x * np.float32(0.1), x)).double()[None]
        if st:
            ref = F.conv_transpose1d(xt, torch.from_numpy(w).double().permute(2, 0, 1).contiguous(), torch.from_numpy(b).double(), stride=st, padding=pad)[0].numpy()
        else:
            ref = F.conv1d(xt, torch.from_numpy(w).double().permute(0, 2, 1).contiguous(), torch.from_numpy(b).double(), padding=pad, dilation=dil)[0].numpy()
        for mode in (13, 40, 42 if st else 24):
            y = engine.debug_conv1d(x, w, b, pad, dil, st, False, in_slope=0.1, in_act=1, mode=mode)
            assert y.shape == ref.shape, (case, mode, y.shape, ref.shape)
            assert np.abs(y - ref).max() <= 2e-5, (case, mode, np.abs(y - ref).max())


def test_conv_math_setting_is_validated_and_reported():
    """sts_set_conv_math: 0 / 1 / 2 accepted, anything else EINVAL; the profile reports bf16 matrix-core FLOPs only when the
    trunk ran on split operands (6 x the algorithmic FLOPs), and none under the exact-fp32 setting."""
    cfg = sb.full_cfg("mbb_fix")
    blob = sb.make_blob(cfg, 99)
    ids = sb.synthetic_ids(24, cfg.vocab)
    syn = engine.Synthesizer(blob)
    with pytest.raises(engine.StsError):
        syn.set_conv_math(4)
    with pytest.raises(engine.StsError):
        syn.set_conv_math(-1)
    syn.set_profiling(True)
    pcm = {}
    for math in CONV_MATHS:
        syn.set_conv_math(math)
        syn.run_batch([ids])
        pcm[math] = syn.pcm_host().astype(np.int32).copy()
        pr = syn.profile()
        if math == "f32":
            assert pr["flops_decoder_bf16_issued"] == 0.0
        else:
            products = 3.0 if math == "f16x2" else 6.0
            assert pr["flops_decoder_mfma"] > 0 and abs(pr["flops_decoder_bf16_issued"] / pr["flops_decoder_mfma"] - products) < 1e-6
            assert pr["conv_math_fallbacks"] == 0
    assert np.abs(pcm["bf16x3"] - pcm["f32"]).max() <= 1 and np.abs(pcm["bf16x3_all"] - pcm["f32"]).max() <= 1
    assert np.abs(pcm["f16x2"] - pcm["f32"]).max() <= 1
    syn.close()


def test_conv_math_default_follows_the_environment():
    """STS_CONV_MATH=f32 in the environment makes the exact-fp32 MFMA path the default of a new engine (read at engine
    construction, so checked in a child process)."""
    import os, subprocess, sys
    code = ("import numpy as np; from summertts_amd import engine, synth_blob as sb;"
            "cfg = sb.full_cfg('mbb_fix'); syn = engine.Synthesizer(sb.make_blob(cfg, 7)); syn.set_profiling(True);"
            "syn.run_batch([sb.synthetic_ids(16, cfg.vocab)]); p = syn.profile();"
            "print('BF16', p['flops_decoder_bf16_issued'] > 0, 'F32', p['flops_decoder_mfma_executed'] > 0)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for val, expect in (("f32", "BF16 False F32 True"), ("bf16x3", "BF16 True F32 False"), (None, "BF16 True F32 False")):
        env = dict(os.environ)
        env.pop("STS_CONV_MATH", None)
        if val:
            env["STS_CONV_MATH"] = val
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert expect in out.stdout, (val, out.stdout[-300:], out.stderr[-300:])


@pytest.mark.parametrize("math", CONV_MATHS)
@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1])
def test_hip_matches_reference_golden(path, math):
    g, cfg, blob = load_golden(path)
    syn = engine.Synthesizer(blob)
    syn.set_conv_math(math)
    syn.set_record_taps(True)
    n = syn.run_batch([g["ids"]], [int(g["sid"])], [float(g["length_scale"])])
    dur = syn.durations(len(g["ids"]))
    assert (dur == g["durations"]).all(), "durations differ from the reference"
    assert_wave_close(syn.tap("wave")[0], g["wave"], "hip vs reference golden")
    assert_pcm_close(syn.pcm_host(), g["pcm"], "hip vs reference golden")
    for k in ("m", "z", "logw"):
        assert np.abs(syn.tap(k) - g[k]).max() <= (1e-3 if k == "logw" else TAP_MAXABS_TOL), k   # logw: the inverse spline amplifies fp32 noise (see below)
    assert int(n[0]) == g["pcm"].size
    syn.close()


@pytest.mark.parametrize("mode", [0, 1], ids=["auto", "generic"])
@pytest.mark.parametrize("kind", TINY)
def test_hip_matches_oracle_all_stages(kind, mode):
    cfg = sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 4321)
    ids = sb.synthetic_ids(29, cfg.vocab, salt=3)
    o = pyref.PortModel(blob).infer_ids(ids, 1, 1.1, taps=True)
    syn = engine.Synthesizer(blob)
    assert syn.info.blob_floats_consumed == blob.size
    syn.set_conv_mode(mode)
    syn.set_record_taps(True)
    syn.run_batch([ids], [1], [1.1])                       # free-running durations
    assert (syn.durations(len(ids)) == o["durations"]).all()
    for k in ("x_enc", "m", "logw", "z_p", "z"):
        # logw: the inverse rational-quadratic spline amplifies fp32 noise by up to 1/min_derivative
        tol = 1e-3 if k == "logw" else TAP_MAXABS_TOL
        err = np.abs(syn.tap(k) - o[k]).max()
        assert err <= tol, (k, err)
    assert_wave_close(syn.tap("wave")[0], o["wave"], kind)
    assert_pcm_close(syn.pcm_host(), o["pcm"], kind)
    syn.close()


@pytest.mark.parametrize("kind", ["hifigan_sdp", "mbb_fix", "ms_hifigan_fix", "ms_sdp"])
def test_batch_equals_single_utterances_bit_exact(kind):
    cfg = sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 11)
    syn = engine.Synthesizer(blob)
    lens = [9, 31, 5, 17, 1, 24]
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=i) for i, t in enumerate(lens)]
    sid = [i % syn.get_speaker_num() for i in range(len(lens))]
    ls = [1.0, 0.9, 1.2, 1.0, 1.1, 1.05]
    # with the kernel variant pinned, an utterance's samples do not depend on what it is batched with
    # (same K order per output element); in automatic mode the dispatcher may pick another variant for
    # another batch geometry, which is fp32-noise different: <= 1 LSB.
    syn.set_conv_mode(6)
    batch = syn.infer_batch(ids, sid, ls)
    dur_b = syn.durations(sum(lens))
    off = 0
    for i in range(len(lens)):
        one = syn.infer_ids(ids[i], sid[i], ls[i])
        assert np.array_equal(one, batch[i]), (kind, i)
        assert np.array_equal(syn.durations(lens[i]), dur_b[off:off + lens[i]])
        off += lens[i]
    syn.set_conv_mode(0)
    auto = syn.infer_batch(ids, sid, ls)
    for i in range(len(lens)):
        assert_pcm_close(auto[i], batch[i], f"{kind} auto vs pinned, utterance {i}")
    syn.close()


def test_forced_durations_edge_cases_and_errors():
    cfg = sb.tiny_cfg("hifigan_fix")
    blob = sb.make_blob(cfg, 3)
    port = pyref.PortModel(blob)
    syn = engine.Synthesizer(blob)
    ids = sb.synthetic_ids(7, cfg.vocab)
    for fd in ([1, 0, 3, 2, 0, 1, 4], [0] * 7, [50, 1, 1, 1, 1, 1, 1]):
        o = port.infer_ids(ids, 0, 1.0, forced_dur=fd)
        syn.set_forced_durations(fd)
        assert_pcm_close(syn.infer_ids(ids, 0, 1.0), o["pcm"], str(fd))
    for T in (1, 2, 4):      # shorter than the attention window + 1
        i2 = sb.synthetic_ids(T, cfg.vocab)
        assert_pcm_close(syn.infer_ids(i2, 0, 1.0), port.infer_ids(i2, 0, 1.0)["pcm"], f"T={T}")
    with pytest.raises(engine.StsError):
        syn.infer_ids([0, cfg.vocab], 0, 1.0)              # id outside the vocabulary
    with pytest.raises(engine.StsError):
        syn.infer_ids([], 0, 1.0)
    syn.close()
    # out-of-range speaker id -> 0 (SynthesizerTrn.cpp:366-369)
    cfg = sb.tiny_cfg("ms_hifigan_fix")
    syn = engine.Synthesizer(sb.make_blob(cfg, 3))
    assert syn.get_speaker_num() == cfg.spk_num
    a = syn.infer_ids(sb.synthetic_ids(8, cfg.vocab), 0, 1.0)
    b = syn.infer_ids(sb.synthetic_ids(8, cfg.vocab), 99, 1.0)
    c = syn.infer_ids(sb.synthetic_ids(8, cfg.vocab), 1, 1.0)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    syn.close()


@pytest.mark.parametrize("kind", ["hifigan_sdp", "mbb_fix"])
def test_full_size_properties(kind):
    """BASELINE.json's full-size model (synthetic weights): size-independent properties + a bounded
    comparison with the oracle (the real reference where oracle/_ref travelled, else the restatement)."""
    cfg = sb.full_cfg(kind)
    blob = sb.make_blob(cfg, 1234)
    syn = engine.Synthesizer(blob)
    ids128 = sb.synthetic_ids(128, cfg.vocab)
    a = syn.infer_ids(ids128, 0, 1.0)
    dur = syn.durations(128)
    assert a.size == int(dur.sum()) * cfg.hop_total and a.size > 128 * cfg.hop_total
    assert np.array_equal(a, syn.infer_ids(ids128, 0, 1.0)), "not deterministic"
    slow = syn.infer_ids(ids128, 0, 1.5)
    assert (syn.durations(128) >= dur).all() and slow.size > a.size   # lengthScale only stretches
    # batch invariance at full size
    ids = [sb.synthetic_ids(40, cfg.vocab, salt=1), ids128, sb.synthetic_ids(77, cfg.vocab, salt=2)]
    batch = syn.infer_batch(ids)
    assert_pcm_close(batch[1], a, "batched vs single (automatic kernel choice)")
    syn.set_conv_mode(6)            # pinned kernel variant: bit-exact batch invariance
    pinned = syn.infer_batch(ids)
    assert np.array_equal(pinned[0], syn.infer_ids(ids[0]))
    assert np.array_equal(pinned[2], syn.infer_ids(ids[2]))
    syn.set_conv_mode(0)
    # streaming decode (sts_infer_ids_stream): windows are smaller launches, so the automatic kernel choice may
    # differ from the one-pass call (split-K sums K in another order): within 1 LSB; pinned kernel: bit-exact
    chunks, _ = syn.infer_ids_stream(ids128, 64)
    assert len(chunks) == -(-int(dur.sum()) // 64)
    assert_pcm_close(np.concatenate(chunks), a, "streamed vs one pass (automatic kernel choice)")
    syn.set_conv_mode(6)
    one = syn.infer_ids(ids128)
    # the split-K kernel family (pinned) is an independent implementation of every conv: it must agree with the
    # grouped / fused-layer kernels the automatic path used for `a` (the 128-channel fused kernel only engages
    # at this size)
    assert_pcm_close(one, a, "split-K kernels vs grouped/fused kernels at full size")
    for chunk in (48, 200):
        chunks, _ = syn.infer_ids_stream(ids128, chunk)
        assert np.array_equal(np.concatenate(chunks), one), chunk
    syn.set_conv_mode(0)
    # generic VALU kernels and matrix-core kernels agree to fp32 noise
    syn.set_conv_mode(1)
    gen = syn.infer_ids(ids[0])
    syn.set_conv_mode(0)
    assert_pcm_close(gen, batch[0], "generic vs mfma")
    # bounded oracle comparison (T=12 keeps the CPU side to seconds)
    small = sb.synthetic_ids(12, cfg.vocab, salt=9)
    model = pyref.RefModel(blob) if pyref.have_ref() else pyref.PortModel(blob)
    o = model.infer_ids(small, 0, 1.0)
    syn.set_record_taps(True)
    syn.run_batch([small])
    assert (syn.durations(12) == o["durations"]).all()
    assert_wave_close(syn.tap("wave")[0], o["wave"], "full-size vs oracle")
    assert_pcm_close(syn.pcm_host(), o["pcm"], "full-size vs oracle")
    syn.close()


def test_long_utterance_and_big_ragged_batch():
    """test/main.cpp joins a whole file into ONE utterance: T in the thousands must work (attention keeps a
    T-long probability row in LDS), and so must a large ragged batch in one call."""
    cfg = sb.tiny_cfg("hifigan_fix")
    blob = sb.make_blob(cfg, 5)
    syn = engine.Synthesizer(blob)
    port = pyref.PortModel(blob)
    ids = sb.synthetic_ids(1500, cfg.vocab, salt=2)
    o = port.infer_ids(ids, 0, 1.0)
    pcm = syn.infer_ids(ids, 0, 1.0)
    assert (syn.durations(len(ids)) == o["durations"]).all()
    assert_pcm_close(pcm, o["pcm"], "T=1500")
    rng = np.random.default_rng(0)
    lens = rng.integers(1, 60, size=200)
    batch_ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=i) for i, t in enumerate(lens)]
    out = syn.infer_batch(batch_ids)
    assert len(out) == 200
    for i in (0, 57, 199):
        assert_pcm_close(out[i], port.infer_ids(batch_ids[i], 0, 1.0)["pcm"], f"utt {i} of 200")
    with pytest.raises(engine.StsError):
        syn.infer_ids(sb.synthetic_ids(60000, cfg.vocab), 0, 1.0)   # beyond the attention kernel's LDS row
    syn.close()


def test_malformed_blobs_are_rejected():
    cfg = sb.tiny_cfg("mbb_fix")
    blob = sb.make_blob(cfg, 5)
    for bad in (blob[: blob.size // 2], blob[:10], np.concatenate([[0, 0, 0, 9], blob[4:]]).astype(np.float32),
                np.concatenate([[0, 0, 7, 0], blob[4:]]).astype(np.float32)):
        with pytest.raises(engine.StsError) as ei:
            engine.Synthesizer(np.ascontiguousarray(bad, np.float32))
        assert "sts error -2" in str(ei.value)
    # trailing bytes after the acoustic sections (the frontend sections of a real blob) are fine
    syn = engine.Synthesizer(np.concatenate([blob, np.zeros(1000, np.float32)]))
    assert syn.info.blob_floats_consumed == blob.size
    syn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["hifigan_sdp", "mbb_fix", "ms_sdp", "istft_fix", "ms_hifigan_sdp"])
def test_streaming_equals_one_pass(kind):
    """sts_infer_ids_stream: chunks decoded with receptive-field halos concatenate to the one-pass
    PCM for every decoder family and for chunk sizes below / around / above the halo: bit for bit with a pinned
    kernel variant, within 1 LSB under the automatic choice (a chunk is a different launch size, and the
    Winograd-domain layer kernel pairs output positions relative to the window start)."""
    cfg = sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 99)
    syn = engine.Synthesizer(blob)
    ids = sb.synthetic_ids(41, cfg.vocab, salt=5)
    sid = 1 if cfg.is_ms else 0
    halo = syn.stream_halo_frames()
    assert 1 <= halo < 400
    o = pyref.PortModel(blob).infer_ids(ids, sid, 1.3)       # the oracle's PCM: what "one pass" is compared with, not only itself
    for mode in (0, 6):
        syn.set_conv_mode(mode)
        full = syn.infer_ids(ids, sid=sid, length_scale=1.3)
        assert (syn.durations(len(ids)) == o["durations"]).all()
        assert_pcm_close(full, o["pcm"], f"one pass vs oracle, {kind}, conv mode {mode}")
        for chunk in (1, 7, halo, 3 * halo + 1, 100000):
            chunks, times = syn.infer_ids_stream(ids, chunk, sid=sid, length_scale=1.3)
            got = np.concatenate(chunks)
            assert got.shape == full.shape, (kind, chunk, got.shape, full.shape)
            if mode == 6:
                assert np.array_equal(got, full), (kind, chunk, int(np.abs(got.astype(np.int32) - full).max()))
            else:
                assert_pcm_close(got, full, f"streamed vs one pass, {kind}, chunk {chunk}")
            assert_pcm_close(got, o["pcm"], f"streamed vs oracle, {kind}, chunk {chunk}")
            assert all(t1 >= t0 for t0, t1 in zip(times, times[1:]))
    syn.set_conv_mode(0)
    # early stop from the callback
    chunks, _ = syn.infer_ids_stream(ids, 8, sid=sid, length_scale=1.3, on_chunk=lambda pcm, off, t: True)
    assert len(chunks) == 1
    # errors: bad chunk size
    with pytest.raises(engine.StsError):
        syn.infer_ids_stream(ids, 0)


def test_request_pool_matches_direct_calls():
    """sts_pool: requests submitted from several threads come back with the PCM a direct
    sts_infer_ids call produces (within 1 LSB: a request may run alone or folded into a packed batch), tickets
    complete in any wait order, batching really happens, and a bad request only fails itself."""
    import threading
    cfg = sb.tiny_cfg("ms_hifigan_sdp")
    blob = sb.make_blob(cfg, 21)
    syn = engine.Synthesizer(blob)
    reqs = [(sb.synthetic_ids(9 + 5 * (i % 7), cfg.vocab, salt=i), i % cfg.spk_num, 1.0 + 0.1 * (i % 3)) for i in range(24)]
    want = [syn.infer_ids(ids, sid=s, length_scale=ls) for ids, s, ls in reqs]
    port = pyref.PortModel(blob)
    oracle = [port.infer_ids(ids, s, ls)["pcm"] for ids, s, ls in reqs]         # every request's PCM according to the oracle
    for i, (w, o) in enumerate(zip(want, oracle)):
        assert_pcm_close(w, o, f"direct call {i} vs oracle")
    pool = engine.Pool(blob, n_engines=2, max_batch=6)
    tickets = [None] * len(reqs)

    def producer(lo, hi):
        for i in range(lo, hi):
            tickets[i] = pool.submit(*reqs[i])
    ths = [threading.Thread(target=producer, args=(k * 8, k * 8 + 8)) for k in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert len(set(tickets)) == len(reqs) and all(t > 0 for t in tickets)
    for i in reversed(range(len(reqs))):          # wait in reverse order
        got = pool.wait(tickets[i])
        assert_pcm_close(got, want[i], f"pool request {i}")
        assert_pcm_close(got, oracle[i], f"pool request {i} vs oracle")
    batches, done = pool.stats()
    assert done == len(reqs) and batches < len(reqs), (batches, done)     # some requests shared a batch
    # a bad id fails alone; its neighbours still complete
    good1 = pool.submit(*reqs[0])
    bad = pool.submit([0, cfg.vocab + 5, 1], 0, 1.0)
    good2 = pool.submit(*reqs[1])
    assert_pcm_close(pool.wait(good1), want[0], "neighbour of a bad request")
    with pytest.raises(engine.StsError):
        pool.wait(bad)
    assert_pcm_close(pool.wait(good2), want[1], "neighbour of a bad request")
    with pytest.raises(engine.StsError):
        pool.wait(12345678)                      # unknown ticket
    pool.close()
    syn.close()


def test_realistic_size_against_the_real_reference():
    """The kernels that only engage at realistic launch sizes (grouped ResBlock launches, the fused layer kernels
    incl. the 128-channel one, the LDS-staged upsamplers) against the reference's own Eigen path: full-size
    HiFi-GAN model, 72 phonemes (~380 frames, ~97 k samples; ~10-20 s of CPU time on the GPU box's host cores)."""
    if not pyref.have_ref():
        pytest.skip("oracle/_ref (the compiled reference) did not travel; the plain-C restatement is too slow at this size")
    cfg = sb.full_cfg("hifigan_sdp")
    blob = sb.make_blob(cfg, 1234)
    ids = sb.synthetic_ids(72, cfg.vocab, salt=3)
    o = pyref.RefModel(blob).infer_ids(ids, 0, 1.0)
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    syn.run_batch([ids])
    assert (syn.durations(72) == o["durations"]).all()
    assert int(o["durations"].sum()) >= 330          # large enough for the 128-channel fused kernel (>= 512 workgroups)
    assert_wave_close(syn.tap("wave")[0], o["wave"], "realistic size vs reference")
    assert_pcm_close(syn.pcm_host(), o["pcm"], "realistic size vs reference")
    syn.close()


@pytest.mark.parametrize("math", CONV_MATHS)
@pytest.mark.parametrize("path", golden_files_v2("full_"), ids=lambda p: p.split("/")[-1])
def test_full_size_configs_match_reference_golden(path, math):
    """BASELINE configs[2]-[4] at FULL model size against outputs of the real reference (tools/make_golden_full.py ran it):
    MB-iSTFT (PQMF) and MS-iSTFT decoders at 96 phonemes (the grouped / fused / Winograd kernels engage), the multi-speaker
    HiFi-GAN model (gin 256: cond paths of the duration predictor, the flow's WaveNet and the decoder) with three speaker
    ids, and ragged 8-utterance batches (64..256 phonemes, the tile choices of a batched launch) of which two members
    each carry reference outputs.  Durations equal, PCM within 1 LSB, float waveform within the stated tolerance."""
    g, cfg, blob, utts, stride = load_golden_v2(path)
    syn = engine.Synthesizer(blob)
    assert syn.info.blob_floats_consumed == blob.size
    syn.set_conv_math(math)     # split-bf16 trunk (default) / exact-fp32 MFMA / split-bf16 in every eligible conv: one tolerance
    syn.set_record_taps(True)
    if "batch_lens" in g:
        lens = [int(t) for t in g["batch_lens"]]
        sids = [int(v) for v in g["batch_sids"]]
        ids = [sb.synthetic_ids(t, cfg.vocab, salt=u) for u, t in enumerate(lens)]
        n_out = syn.run_batch(ids, sids, [1.0] * len(ids))
        pcm = syn.pcm_host()
        wave = syn.tap("wave")[0]
        dur = syn.durations(sum(lens))
        soff = np.concatenate([[0], np.cumsum(n_out)])
        toff = np.concatenate([[0], np.cumsum(lens)])
        for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
            assert np.array_equal(ids_u, ids[u]) and sid_u == sids[u]
            assert (dur[toff[u]:toff[u + 1]] == dur_u).all(), f"utterance {u}: durations differ from the reference"
            assert_pcm_close(pcm[soff[u]:soff[u + 1]], pcm_u, f"{path} utterance {u} of the batch")
            assert_wave_close(wave[soff[u]:soff[u + 1]][::stride], wave_u, f"{path} utterance {u} of the batch")
    else:
        for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
            syn.run_batch([ids_u], [sid_u], [ls_u])
            assert (syn.durations(len(ids_u)) == dur_u).all(), f"utterance {u}: durations differ from the reference"
            assert_pcm_close(syn.pcm_host(), pcm_u, f"{path} utterance {u}")
            assert_wave_close(syn.tap("wave")[0][::stride], wave_u, f"{path} utterance {u}")
    syn.close()


@pytest.mark.parametrize("path", [p for p in golden_files_v2("full_") + golden_files_v2("loud_") + golden_files_v2("real_") if "tiny" not in p],
                         ids=lambda p: p.split("/")[-1])
def test_pre_split_trunk_path_matches_reference_golden(path):
    """Round 6: the decoder stages of 128 k channels on pre-split, channel-minor activations (conv_h2p.hip: entry split, planes / x16
    epilogues, LDS-DMA staging).  The engine takes that path by itself from ~8 tiles per CU on; here it is FORCED (sts_debug_set h2p = 2) for
    every full-size fixture the real reference produced -- single utterances (ragged tile edges, one tile per CU) and the 8- / 32- / 64-
    utterance batches, Gaussian and realistic weight statistics, near-full-scale outputs -- under the same unscaled tolerances, and the
    staged kernels (h2p = 0) run the same fixture next to it.  /root/reference/src/modules/ResBlock1.cpp:55-69, Generator_hifigan.cpp:151-175."""
    g, cfg, blob, utts, stride = load_golden_v2(path)
    syn = engine.Synthesizer(blob)
    syn.set_conv_math("f16x2")
    syn.set_record_taps(True)
    syn.set_profiling(True)
    for h2p in (2, 0):
        syn.debug_set("h2p", h2p)
        if "batch_lens" in g:
            lens = [int(t) for t in g["batch_lens"]]
            sids = [int(v) for v in g["batch_sids"]]
            ids = [sb.synthetic_ids(t, cfg.vocab, salt=u) for u, t in enumerate(lens)]
            n_out = syn.run_batch(ids, sids, [1.0] * len(ids))
            assert syn.profile()["conv_math_fallbacks"] == 0
            pcm = syn.pcm_host()
            wave = syn.tap("wave")[0]
            dur = syn.durations(sum(lens))
            soff = np.concatenate([[0], np.cumsum(n_out)])
            toff = np.concatenate([[0], np.cumsum(lens)])
            for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
                assert (dur[toff[u]:toff[u + 1]] == dur_u).all(), f"utterance {u}: durations differ from the reference"
                assert_pcm_close(pcm[soff[u]:soff[u + 1]], pcm_u, f"{path} utterance {u} of the batch (h2p={h2p})")
                assert_wave_close(wave[soff[u]:soff[u + 1]][::stride], wave_u, f"{path} utterance {u} of the batch (h2p={h2p})")
        else:
            for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
                syn.run_batch([ids_u], [sid_u], [ls_u])
                assert syn.profile()["conv_math_fallbacks"] == 0
                assert (syn.durations(len(ids_u)) == dur_u).all(), f"utterance {u}: durations differ from the reference"
                assert_pcm_close(syn.pcm_host(), pcm_u, f"{path} utterance {u} (h2p={h2p})")
                assert_wave_close(syn.tap("wave")[0][::stride], wave_u, f"{path} utterance {u} (h2p={h2p})")
    syn.close()


def test_explicit_multistream_filter_bias_full_size():
    """a full-size MS-iSTFT blob whose learned synthesis filter carries an explicit bias of 0.0375 (model.hip: fir_bias, added by the
    fused iSTFT tail and by synth_fir): the HIP path against the oracle (pinned on the compiled reference by
    tests/test_oracle_cpu.py::test_port_matches_live_reference_with_an_explicit_multistream_filter_bias), fused and unfused tail, and the bias
    is really there.  /root/reference/src/nn_op/nn_conv1d.cpp:40-46, Generator_MS.cpp."""
    from test_oracle_cpu import ms_blob_with_explicit_filter_bias
    cfg, blob, at = ms_blob_with_explicit_filter_bias()
    ids = sb.synthetic_ids(9, cfg.vocab, salt=3)
    want = pyref.PortModel(blob).infer_ids(ids, 0, 1.0)
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    for fused in (1, 0):
        syn.debug_set("tail_fused", fused)
        syn.run_batch([ids], [0], [1.0])
        assert_pcm_close(syn.pcm_host(), want["pcm"], f"ms_fix explicit filter bias (tail_fused={fused})")
        assert_wave_close(syn.tap("wave")[0], want["wave"], f"ms_fix explicit filter bias (tail_fused={fused})")
    wave = syn.tap("wave")[0].copy()
    syn.close()
    blob0 = blob.copy(); blob0[at] = 0.0
    syn0 = engine.Synthesizer(blob0)
    syn0.set_record_taps(True)
    syn0.run_batch([ids], [0], [1.0])
    assert np.abs((wave - syn0.tap("wave")[0]) - np.float32(0.0375)).max() <= 1e-6
    syn0.close()


UPS_CASES = [  # Cin, Cout, k, pad, stride, L      (HiFi-GAN: k16 s8 / k4 s2; MB-iSTFT: k16 s4; tiny models: k8 s4; odd lengths -> ragged tiles)
    (512, 256, 16, 4, 8, 668), (256, 128, 16, 4, 8, 1301), (128, 64, 4, 1, 2, 3001), (64, 32, 4, 1, 2, 4099), (256, 128, 16, 6, 4, 700),
    (64, 32, 8, 2, 4, 77), (32, 32, 16, 4, 8, 5), (96, 64, 4, 1, 2, 1), (64, 96, 11, 3, 4, 130),
]


@pytest.mark.parametrize("case", UPS_CASES, ids=str)
def test_upsampler_row_interleaved_phases_equal_the_phase_major_form(case):
    """Round 6: transposed convs of stride 2 / 4 / 8 through the split-operand kernels with the phases interleaved along the packed rows
    (ConvArgs::rowph: whole-sector stores).  Same products in the same order per output value: BIT-identical to the phase-major packing for every
    tile shape that does not split K, both arithmetics, and within the usual bound of a float64 transposed conv.
    /root/reference/src/nn_op/nn_conv1d_transposed.cpp:24-53."""
    import torch
    import torch.nn.functional as F
    ci, co, k, pad, st, L = case
    rng = np.random.default_rng(ci * 31 + co + k + L)
    x = (rng.standard_normal((ci, L)) * rng.uniform(0.05, 3.0, (ci, 1))).astype(np.float32)
    w = (rng.standard_normal((co, k, ci)) / np.sqrt(k * ci / st)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    ref = F.conv_transpose1d(torch.from_numpy(x).double()[None], torch.from_numpy(w).double().permute(2, 0, 1).contiguous(), torch.from_numpy(b).double(),
                             stride=st, padding=pad)[0].numpy()
    for auto, codes in ((13, (20, 23, 42, 43)), (50, (60, 63, 82, 83))):       # split-bf16 / two-term fp16: automatic tile, 128 x 128, 64 x 128, the summed-input tiles
        y0 = engine.debug_conv1d(x, w, b, pad, 1, st, False, in_slope=0.1, in_act=1, mode=codes[0])
        for mode in codes:
            y = engine.debug_conv1d(x, w, b, pad, 1, st, False, in_slope=0.1, in_act=1, mode=100 + mode)
            assert y.shape == ref.shape
            assert np.array_equal(y, y0), (case, mode, np.abs(y - y0).max())
        y = engine.debug_conv1d(x, w, b, pad, 1, st, False, in_slope=0.1, in_act=1, mode=100 + auto)      # (may pick the K-split tile: other order)
        xa = np.where(x < 0, x * np.float32(0.1), x).astype(np.float32)
        refa = F.conv_transpose1d(torch.from_numpy(xa).double()[None], torch.from_numpy(w).double().permute(2, 0, 1).contiguous(),
                                  torch.from_numpy(b).double(), stride=st, padding=pad)[0].numpy()
        assert np.abs(y - refa).max() <= 2e-5 and np.abs(y0 - refa).max() <= 2e-5, (case, auto, np.abs(y - refa).max())
    y = engine.debug_conv1d(x, w, None, pad, 1, st, False, mode=150)       # no bias
    assert np.abs(y - (ref - b[:, None])).max() <= 2e-5


@pytest.mark.parametrize("kind", ["hifigan_sdp", "mbb_fix", "ms_hifigan_sdp"])
def test_engine_upsamplers_row_interleaved_on_off_identical(kind):
    """The engine's upsamplers with and without the row-interleaved packing (sts_debug_set ups_rowph): the same PCM, bit for bit, at one
    utterance (K-split tile on the first upsampler in both forms) and in a ragged batch, under both split-operand arithmetics."""
    cfg = sb.full_cfg(kind)
    blob = sb.make_blob(cfg, 11)
    syn = engine.Synthesizer(blob)
    batch = [sb.synthetic_ids(t, cfg.vocab, salt=u) for u, t in enumerate((37, 9, 64))]
    sids = [0, 1, 2] if cfg.is_ms else [0, 0, 0]
    for math in ("f16x2", "bf16x3"):
        syn.set_conv_math(math)
        out = {}
        for on in (1, 0):
            syn.debug_set("ups_rowph", on)
            syn.run_batch([batch[0]], [sids[0]], [1.0])
            one = syn.pcm_host().copy()
            syn.run_batch(batch, sids, [1.0, 0.9, 1.1])
            out[on] = (one, syn.pcm_host().copy())
        assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1]), (kind, math)
    syn.close()


@pytest.mark.parametrize("C,k,dil,L", [(128, 3, 1, 777), (128, 7, 3, 1500), (128, 11, 5, 300), (256, 3, 1, 1029), (256, 11, 5, 97), (128, 3, 1, 5), (512, 3, 1, 130)])
def test_pre_split_conv_against_float64(C, k, dil, L):
    """One conv through split_planes + conv_h2p_group, every tile code, all three output forms (fp32 [C][L], the channel-minor fp32 copy, the
    two fp16 planes of lrelu(out)): as accurate against float64 as the staged two-term kernel (the k order inside a 16-channel chunk differs,
    so the two are not bit-identical), the channel-minor copy bit-equal to the plain one, the planes within 2^-21 of lrelu(out).
    /root/reference/src/nn_op/nn_conv1d.cpp:118-199."""
    rng = np.random.default_rng(C + k + L)
    x = rng.standard_normal((C, L)).astype(np.float32) * 1.5
    w = (rng.standard_normal((C, k, C)) / np.sqrt(k * C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((C, L)).astype(np.float32)
    xa = np.where(x < 0, x * np.float32(0.1), x).astype(np.float64)
    pad = dil * (k - 1) // 2
    xp = np.pad(xa, ((0, 0), (pad, pad)))
    r64 = sum(w[:, t, :].astype(np.float64) @ xp[:, t * dil:t * dil + L] for t in range(k)) + b[:, None] + res
    staged = engine.debug_conv1d(x, w, b, pad, dil, 0, False, 0.1, 1, mode=60) + res
    e_staged = float(np.abs(staged - r64).max())
    for tile in range(11):
        y, y16, yp, _ = engine.debug_conv_h2p(x, w, b, dil, res, 0.1, 0.1, tile=tile, members=2 if tile % 2 else 1)
        lre = np.where(y < 0, y * np.float32(0.1), y)
        assert np.abs(y - r64).max() <= 1.5 * e_staged + 1e-7, (tile, float(np.abs(y - r64).max()), e_staged)
        assert np.array_equal(y16, y), tile
        assert np.abs(yp - lre).max() <= 5e-7 * max(1.0, float(np.abs(lre).max())), tile


@pytest.mark.parametrize("C,k,dil,L", [(128, 3, 1, 777), (128, 3, 5, 333), (128, 5, 2, 211), (128, 7, 3, 1500), (128, 11, 5, 300), (256, 11, 1, 190), (256, 7, 5, 131), (128, 3, 1, 5)])
def test_winograd_domain_lab_conv_against_float64(C, k, dil, L):
    """conv_h2w.hip (round 6, lab: not dispatched by the engine -- it measures 0.82-0.84 x the direct pre-split kernel's speed):
    segmented F(2,3) / F(2,2) with two-term fp16 MFMAs on the transformed operands, every tap count it is instantiated for, dilations 1-5, with
    the residual.  Its error against float64 must not exceed the direct pre-split kernel's (tools/wino_f16x2_numerics.py predicts it is lower),
    and its channel-minor output of lrelu(out) must equal lrelu of its fp32 output.  /root/reference/src/nn_op/nn_conv1d.cpp:118-199."""
    rng = np.random.default_rng(C + k + L + dil)
    x = rng.standard_normal((C, L)).astype(np.float32) * 1.5
    w = (rng.standard_normal((C, k, C)) / np.sqrt(k * C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    res = rng.standard_normal((C, L)).astype(np.float32)
    xa = np.where(x < 0, x * np.float32(0.1), x).astype(np.float64)
    pad = dil * (k - 1) // 2
    xp = np.pad(xa, ((0, 0), (pad, pad)))
    r64 = sum(w[:, t, :].astype(np.float64) @ xp[:, t * dil:t * dil + L] for t in range(k)) + b[:, None] + res
    yd, *_ = engine.debug_conv_h2p(x, w, b, dil, res, 0.1, 0.1, tile=0)
    y, y16, _ = engine.debug_conv_h2w(x, w, b, dil, res, 0.1, 0.1, members=2)
    e_dir = float(np.sqrt(((yd - r64) ** 2).mean())); e_w = float(np.sqrt(((y - r64) ** 2).mean()))
    assert e_w <= 1.05 * e_dir + 1e-8, (e_w, e_dir)
    assert np.abs(y - r64).max() <= 1.5 * np.abs(yd - r64).max() + 1e-7
    assert np.array_equal(y16, np.where(y < 0, y * np.float32(0.1), y))


@pytest.mark.parametrize("path", golden_files_v2("loud_"), ids=lambda p: p.split("/")[-1])
def test_near_full_scale_utterances_discriminate_the_trunk_arithmetics(path):
    """the bench-shaped models peak at |o| ~ 0.05 of full scale, where "int16 within 1 LSB" is a ~5e-4-relative
    test that cannot tell 22-bit from 24-bit operands.  These fixtures are FULL-size utterances (the bench's own 128 phonemes of
    hifigan_sdp; 96 phonemes of mbb_fix) whose tail gain puts the waveform at peak |o| 0.7-0.8 (rms 0.25-0.34) WITHOUT saturating,
    outputs made by the compiled reference.  Every trunk arithmetic runs them under the unscaled tolerances; the per-arithmetic LSB
    histogram is printed and written to gpurun_out/lsb_histograms/ (the committed copy: profiles/r04_lsb_histograms.json), and the
    default two-term fp16 form must not have a single > 1-LSB sample that the exact-fp32 MFMA path does not have."""
    import json
    import os
    g, cfg, blob, utts, stride = load_golden_v2(path)
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    syn.set_profiling(True)
    hist = {}
    for math in CONV_MATHS:
        syn.set_conv_math(math)
        for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
            syn.run_batch([ids_u], [sid_u], [ls_u])
            assert syn.profile()["conv_math_fallbacks"] == 0
            assert (syn.durations(len(ids_u)) == dur_u).all(), f"{math}: durations differ from the reference"
            pcm = syn.pcm_host().astype(np.int64)
            d = np.abs(pcm - pcm_u.astype(np.int64))
            wave = syn.tap("wave")[0][::stride].astype(np.float64)
            err = wave - wave_u.astype(np.float64)
            hist[math] = {"samples": int(d.size), "lsb0": int((d == 0).sum()), "lsb1": int((d == 1).sum()), "lsb_gt1": int((d > 1).sum()),
                          "max_lsb": int(d.max()), "wave_rmse": float(np.sqrt((err ** 2).mean())), "wave_maxabs": float(np.abs(err).max()),
                          "peak": float(np.abs(wave_u).max()), "rms": float(np.sqrt((wave_u.astype(np.float64) ** 2).mean()))}
    name = os.path.basename(path)[:-4]
    print(f"\n{name}: peak |o| {hist['f32']['peak']:.3f}, rms {hist['f32']['rms']:.3f}, {hist['f32']['samples']} samples")
    for math in CONV_MATHS:
        h = hist[math]
        print(f"  {math:11s} exact {h['lsb0']:7d}  1 LSB {h['lsb1']:6d}  > 1 LSB {h['lsb_gt1']:3d}   waveform rmse {h['wave_rmse']:.3e}  max-abs {h['wave_maxabs']:.3e}")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "lsb_histograms")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(hist, f, indent=1)
    assert hist["f32"]["peak"] > 0.6 and hist["f32"]["rms"] > 0.2, "the fixture is not near full scale"
    for math in CONV_MATHS:
        h = hist[math]
        assert h["max_lsb"] <= 1, (name, math, h)
        assert h["wave_rmse"] <= 2e-6 and h["wave_maxabs"] <= 1e-5, (name, math, h)      # conftest's unscaled tolerances at peak ~0.8
    assert hist["f16x2"]["lsb_gt1"] <= hist["f32"]["lsb_gt1"]
    # the two-term form may move a few more samples across an integer boundary than exact fp32, not a different order of magnitude
    assert hist["f16x2"]["lsb1"] <= 2 * hist["f32"]["lsb1"] + 50, (hist["f16x2"]["lsb1"], hist["f32"]["lsb1"])
    syn.close()


@pytest.mark.parametrize("path", golden_files_v2("real_"), ids=lambda p: p.split("/")[-1])
def test_realistic_weight_statistics_match_reference_golden(path):
    """every fixture, bench line and overflow test of rounds 1-4 ran on ONE weight distribution (i.i.d.
    N(0, gain / sqrt(fan_in)), LayerNorm ~ identity).  These fixtures use synth_blob.py's stats="realistic" -- per-output-channel
    log-normal gains (sigma 1), one weight in a thousand 20x larger, biases on every conv (also the tails upstream builds without),
    LayerNorm gamma ~ U(0.5, 2) and beta ~ N(0, 0.3) -- i.e. what the two-term fp16 arithmetic's range logic (per-conv power-of-two weight
    scale into [2^13, 2^14), the 60 000 activation limit, the pin-to-bf16x3 state machine) meets in a weight-normed checkpoint.  FULL-size
    HiFi-GAN, MB-iSTFT, multi-speaker HiFi-GAN and -- for the first time at full size -- the plain iSTFT decoder
    (Generator_Istft.cpp:149-198), outputs made by the compiled reference, peak |o| 0.57-0.66; plus the tiny models.  All four arithmetics,
    ONE unscaled tolerance: durations equal, PCM <= 1 LSB, waveform RMSE <= 2e-6 / max-abs <= 1e-5, latent z within 5e-5 of its scale.
    The LSB histograms and the fallback counters go to gpurun_out/lsb_histograms/ (committed copy: profiles/r05_lsb_histograms.json)."""
    import json
    import os
    g, cfg, blob, utts, stride = load_golden_v2(path)
    assert cfg.stats == "realistic"
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    syn.set_profiling(True)
    hist = {}
    zs = int(g["z_stride"])
    for math in CONV_MATHS:
        syn.set_conv_math(math)
        h = {"samples": 0, "lsb0": 0, "lsb1": 0, "lsb_gt1": 0, "max_lsb": 0, "wave_rmse": 0.0, "wave_maxabs": 0.0, "z_maxabs": 0.0, "peak": 0.0}
        se = 0.0
        for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
            syn.run_batch([ids_u], [sid_u], [ls_u])
            assert (syn.durations(len(ids_u)) == dur_u).all(), f"{math}: durations differ from the reference"
            d = np.abs(syn.pcm_host().astype(np.int64) - pcm_u.astype(np.int64))
            err = syn.tap("wave")[0][::stride].astype(np.float64) - wave_u.astype(np.float64)
            z_ref = g[f"z_{u}"]
            ze = float(np.abs(syn.tap("z")[:, ::zs] - z_ref).max())
            assert ze <= TAP_MAXABS_TOL * max(1.0, float(np.abs(z_ref).max())), (math, ze)
            h["samples"] += int(d.size); h["lsb0"] += int((d == 0).sum()); h["lsb1"] += int((d == 1).sum()); h["lsb_gt1"] += int((d > 1).sum())
            h["max_lsb"] = max(h["max_lsb"], int(d.max())); se += float((err ** 2).sum()); h["wave_maxabs"] = max(h["wave_maxabs"], float(np.abs(err).max()))
            h["z_maxabs"] = max(h["z_maxabs"], ze); h["peak"] = max(h["peak"], float(np.abs(wave_u).max()))
        h["wave_rmse"] = float(np.sqrt(se / max(1, sum(w.size for *_, w in utts))))
        p_ = syn.profile()
        h["conv_math_fallbacks"] = int(p_["conv_math_fallbacks"]); h["conv_math_pinned"] = int(p_["conv_math_pinned"])
        hist[math] = h
    name = os.path.basename(path)[:-4]
    print(f"\n{name}: peak |o| {hist['f32']['peak']:.3f}, {hist['f32']['samples']} samples")
    for math in CONV_MATHS:
        h = hist[math]
        print(f"  {math:11s} exact {h['lsb0']:7d}  1 LSB {h['lsb1']:6d}  > 1 LSB {h['lsb_gt1']:3d}   waveform rmse {h['wave_rmse']:.3e}  max-abs {h['wave_maxabs']:.3e}  "
              f"z max-abs {h['z_maxabs']:.2e}  fallbacks {h['conv_math_fallbacks']}")
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "lsb_histograms")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(hist, f, indent=1)
    for math in CONV_MATHS:
        h = hist[math]
        assert h["max_lsb"] <= 1, (name, math, h)
        assert h["wave_rmse"] <= 2e-6 and h["wave_maxabs"] <= 1e-5, (name, math, h)
    assert hist["f16x2"]["conv_math_fallbacks"] == 0 and hist["f16x2"]["conv_math_pinned"] == 0, "the two-term fp16 form left its range on realistic weights"
    assert hist["f16x2"]["lsb1"] <= 2 * hist["f32"]["lsb1"] + 50, (hist["f16x2"]["lsb1"], hist["f32"]["lsb1"])     # never worse than 2x f32's one-LSB count
    syn.close()


@pytest.mark.parametrize("math", CONV_MATHS)
@pytest.mark.parametrize("path", golden_files_v2("amp_"), ids=lambda p: p.split("/")[-1])
def test_amplitude_edge_matches_reference_golden(path, math):
    """High amplitudes against the real reference: HiFi-GAN outputs driven into tanh saturation (|o| up to exactly 1.0 ->
    pcm 32737) and MB-iSTFT / MS / iSTFT outputs beyond +-1.0, where the reference's unclipped (int16_t)(o * 32737) wraps
    around modulo 2^16 (SynthesizerTrn.cpp:393-396; the engine reproduces the x86 cast: devmath.hpp pcm_cast).  Waveform
    tolerance scales with the peak (fp32 noise is relative); PCM within 1 LSB modulo 2^16."""
    g, cfg, blob, utts, stride = load_golden_v2(path)
    syn = engine.Synthesizer(blob)
    syn.set_conv_math(math)        # (round 4: every trunk arithmetic, not only the default)
    syn.set_record_taps(True)
    for u, ids_u, sid_u, ls_u, dur_u, pcm_u, wave_u in utts:
        syn.run_batch([ids_u], [sid_u], [ls_u])
        assert (syn.durations(len(ids_u)) == dur_u).all()
        wave = syn.tap("wave")[0][::stride].astype(np.float64)
        peak = max(1.0, float(np.abs(wave_u).max()))
        err = wave - wave_u.astype(np.float64)
        assert np.sqrt((err ** 2).mean()) <= AMP_WAVE_RMSE_TOL * peak and np.abs(err).max() <= AMP_WAVE_MAXABS_TOL * peak, \
            (path, float(np.sqrt((err ** 2).mean())), float(np.abs(err).max()), peak)
        assert_pcm_close_wrapped(syn.pcm_host(), pcm_u, path)
        if "wrap" in path:
            assert (np.abs(wave_u) > 1.0009).mean() > 0.1          # the fixture really exercises the wrap-around
            wrapped = np.abs(wave_u) * 32737 >= 32768
            assert wrapped.mean() > 0.1 and (syn.pcm_host()[wrapped] == pcm_u[wrapped]).mean() > 0.8   # the rest: 1 LSB (mod 2^16)
        else:
            assert np.abs(wave_u).max() > 0.85
    syn.close()


def test_pcm_cast_beyond_int32_matches_the_x86_reference_build():
    """|o * 32737| >= 2^31 (and NaN): the reference build's cvttss2si yields 0x80000000, whose low 16 bits are 0; the
    engine's pcm_cast reproduces that instead of the GPU's saturating conversion (which would give -1 for +overflow)."""
    import dataclasses
    cfg = dataclasses.replace(sb.tiny_cfg("mbb_fix"), mag_bias=30.0)       # |X_k| ~ e^30: far beyond int32 after scaling
    blob = sb.make_blob(cfg, 5)
    ids = sb.synthetic_ids(9, cfg.vocab)
    o = (pyref.RefModel(blob) if pyref.have_ref() else pyref.PortModel(blob)).infer_ids(ids, 0, 1.0)
    assert (np.abs(o["wave"]) * 32737 >= 2 ** 31).mean() > 0.5
    pcm = engine.Synthesizer(blob).infer_ids(ids, 0, 1.0)
    big = np.abs(o["wave"].astype(np.float64)) * 32737 >= 2.0 ** 31 * 1.001
    assert (pcm[big] == 0).all() and (o["pcm"][big] == 0).all()


def test_multi_device_entry_matches_single_engine():
    """sts_multi_* (SURVEY 8b/8e): one process, an engine per listed device (here the one GPU twice), utterances sharded
    longest-first by phoneme count, PCM back in input order."""
    cfg = sb.tiny_cfg("ms_hifigan_sdp")
    blob = sb.make_blob(cfg, 21)
    syn = engine.Synthesizer(blob)
    lens = [9, 31, 5, 17, 2, 24, 11]
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=i) for i, t in enumerate(lens)]
    sid = [i % cfg.spk_num for i in range(len(lens))]
    ls = [1.0 + 0.05 * i for i in range(len(lens))]
    want = [syn.infer_ids(a, s, l) for a, s, l in zip(ids, sid, ls)]
    port = pyref.PortModel(blob)
    oracle = [port.infer_ids(a, s, l)["pcm"] for a, s, l in zip(ids, sid, ls)]
    md = engine.MultiDevice(blob, [0, 0, 0])
    assert md.device_count() == 3
    slot = md.shard_of(lens)
    from summertts_amd import sharding
    shards = sharding.shard_utterances(lens, 3)               # the python sharder of the one-process-per-GPU path: same rule
    for k, sh in enumerate(shards):
        assert all(slot[u] == k for u in sh)
    got = md.infer_batch(ids, sid, ls)
    for i in range(len(lens)):
        assert_pcm_close(got[i], want[i], f"multi-device utterance {i}")
        assert_pcm_close(got[i], oracle[i], f"multi-device utterance {i} vs oracle")
    one = md.infer_batch(ids[:1], sid[:1], ls[:1])            # fewer utterances than devices: idle devices take no part
    assert_pcm_close(one[0], want[0], "single utterance on a 3-slot handle")
    with pytest.raises(engine.StsError):
        md.infer_batch([[0, cfg.vocab + 1]])                  # a bad id fails the call, outputs released
    with pytest.raises(engine.StsError):
        engine.MultiDevice(blob, [0, 99])                     # device that does not exist
    md.close()
    syn.close()


def test_wide_dilation_model_takes_the_checked_paths():
    """a ResBlock whose later layers have dil * (k - 1) > 64 (k = 11, dilation 7) must not reach the grouped
    matrix-core launch, whose staged LDS window cannot hold that halo; an attention window > 7 needs more than 16
    relative-position logits.  Both against the oracle."""
    import dataclasses
    cfg = dataclasses.replace(sb.tiny_cfg("hifigan_fix"), res_k=(11, 3), res_d=((1, 7, 2), (1, 3, 9)), window=9)
    blob = sb.make_blob(cfg, 8)
    ids = sb.synthetic_ids(33, cfg.vocab, salt=1)
    o = pyref.PortModel(blob).infer_ids(ids, 0, 1.0, taps=True)
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    syn.run_batch([ids])
    assert (syn.durations(len(ids)) == o["durations"]).all()
    assert np.abs(syn.tap("x_enc") - o["x_enc"]).max() <= TAP_MAXABS_TOL
    assert_wave_close(syn.tap("wave")[0], o["wave"], "wide dilation")
    assert_pcm_close(syn.pcm_host(), o["pcm"], "wide dilation")
    syn.close()


@pytest.mark.parametrize("math", CONV_MATHS)
@pytest.mark.parametrize("up_init,res_k,res_d", [
    (256, (5, 9), ((1, 2, 4), (3, 1, 5))),          # stages of 128 / 64 channels: grouped split-bf16 convs + the fused layer kernel
    (128, (13, 3, 7), ((1, 5, 2), (2, 6, 1), (4, 1, 3))),   # stages of 64 / 32 channels, three chains, halos up to 60 positions
    (512, (3, 5), ((1, 3, 2), (5, 1, 4))),          # 256 / 128 channels: unfused grouped convs on both stages at this size
])
def test_unusual_resblock_geometries_against_the_oracle(up_init, res_k, res_d, math):
    """ResBlock kernel sizes / dilations / chain counts other than HiFi-GAN's (3, 7, 11) x (1, 3, 5), at the channel widths the
    fused and grouped trunk kernels are instantiated for, under every trunk arithmetic -- against the C restatement."""
    import dataclasses
    cfg = dataclasses.replace(sb.tiny_cfg("hifigan_fix"), up_init=up_init, res_k=res_k, res_d=res_d)
    blob = sb.make_blob(cfg, 31 + up_init)
    ids = sb.synthetic_ids(41, cfg.vocab, salt=2)
    o = pyref.PortModel(blob).infer_ids(ids, 0, 1.0, taps=True)
    syn = engine.Synthesizer(blob)
    syn.set_conv_math(math)
    syn.set_record_taps(True)
    syn.run_batch([ids])
    assert (syn.durations(len(ids)) == o["durations"]).all()
    assert_wave_close(syn.tap("wave")[0], o["wave"], f"resblock geometry {up_init} {res_k}")
    assert_pcm_close(syn.pcm_host(), o["pcm"], f"resblock geometry {up_init} {res_k}")
    syn.close()


@pytest.mark.parametrize("width", [256, 32])
def test_fused_column_layers_at_every_instantiated_width(width):
    """col_layer_kernel is instantiated for 32 / 64 / 192 / 256 channels (64 and 192 run in every other test): a model whose
    text encoder AND stochastic duration predictor are `width` wide drives the 16-wave (256) and 2-wave (32) variants, with
    ragged lengths that leave partial 16-step column blocks, against the oracle."""
    import dataclasses
    cfg = dataclasses.replace(sb.tiny_cfg("hifigan_sdp"), hidden=width, sdp_filter=width, n_layers=1, ffn=64)
    blob = sb.make_blob(cfg, 17)
    port = pyref.PortModel(blob)
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    lens = [37, 16, 5, 49]
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=i) for i, t in enumerate(lens)]
    syn.run_batch(ids)
    x_enc, logw, pcm = syn.tap("x_enc"), syn.tap("logw"), syn.pcm_host()
    dur = syn.durations(sum(lens))
    toff = np.concatenate([[0], np.cumsum(lens)])
    soff = 0
    for i, a in enumerate(ids):
        o = port.infer_ids(a, 0, 1.0, taps=True)
        assert (dur[toff[i]:toff[i + 1]] == o["durations"]).all(), i
        assert np.abs(x_enc[:, toff[i]:toff[i + 1]] - o["x_enc"]).max() <= TAP_MAXABS_TOL
        assert np.abs(logw[:, toff[i]:toff[i + 1]] - o["logw"]).max() <= 1e-3
        assert_pcm_close(pcm[soff:soff + o["pcm"].size], o["pcm"], f"width {width}, utterance {i}")
        soff += o["pcm"].size
    syn.close()


@pytest.mark.parametrize("kind", ["hifigan_sdp", "ms_hifigan_fix", "mbb_fix"])
def test_block_attention_kernel_matches_oracle_at_every_size(kind):
    """attention_mfma_kernel (16 queries per workgroup on the matrix cores) normally engages from ~100 workgroups on; here it is
    forced for tiny models and ragged lengths (partial 16-query blocks, a 1-phoneme utterance, keys past the last 64-key chunk)
    and the text-encoder output is compared with the oracle utterance by utterance."""
    cfg = sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 23)
    port = pyref.PortModel(blob)
    syn = engine.Synthesizer(blob)
    syn.debug_set("attn_block_min_wgs", 1)
    syn.set_record_taps(True)
    lens = [70, 16, 1, 33, 129]
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=i) for i, t in enumerate(lens)]
    sid = [i % syn.get_speaker_num() for i in range(len(lens))]
    syn.run_batch(ids, sid)
    x_enc, pcm, dur = syn.tap("x_enc"), syn.pcm_host(), syn.durations(sum(lens))
    toff = np.concatenate([[0], np.cumsum(lens)])
    soff = 0
    for i, a in enumerate(ids):
        o = port.infer_ids(a, sid[i], 1.0, taps=True)
        assert np.abs(x_enc[:, toff[i]:toff[i + 1]] - o["x_enc"]).max() <= TAP_MAXABS_TOL, (kind, i)
        assert (dur[toff[i]:toff[i + 1]] == o["durations"]).all()
        assert_pcm_close(pcm[soff:soff + o["pcm"].size], o["pcm"], f"{kind} utterance {i}")
        soff += o["pcm"].size
    syn.close()


@pytest.mark.parametrize("H,k,half", [(32, 5, 32), (96, 5, 32), (128, 5, 64), (160, 3, 48), (192, 3, 96), (256, 3, 64), (64, 1, 32)])
def test_flow_layer_kernel_at_every_width_it_admits(H, k, half):
    """flow_layer_kernel (wn_flow.hip: one launch per WaveNet layer) was tested at H = 64 and 192 only, while
    flow_layer_shape_ok admits H = 32 ... 256 (with the gate conv's taps limited by the registers a wave keeps its weights in).  Small models
    with every admitted width / tap count / half-channel count, one utterance and a ragged batch, the latent z and the PCM against the
    oracle -- and against the per-conv launches, which must not be bit-identical (different arithmetic: that is the evidence the fused
    kernel engaged)."""
    import dataclasses
    cfg = dataclasses.replace(sb.tiny_cfg("hifigan_sdp"), flow_hidden=H, flow_k=k, inter=2 * half, flow_layers=3)
    blob = sb.make_blob(cfg, 23)
    port = pyref.PortModel(blob)
    syn = engine.Synthesizer(blob)
    syn.set_record_taps(True)
    for T in (7, 40):
        ids = sb.synthetic_ids(T, cfg.vocab, salt=T)
        o = port.infer_ids(ids, 0, 1.0, taps=True)
        z = {}
        for fused in (1, 0):
            syn.debug_set("flow_fused", fused)
            syn.run_batch([ids])
            assert (syn.durations(T) == o["durations"]).all()
            z[fused] = syn.tap("z").copy()
            assert np.abs(z[fused] - o["z"]).max() <= TAP_MAXABS_TOL * max(1.0, float(np.abs(o["z"]).max())), (H, k, half, T, fused)
            assert_pcm_close(syn.pcm_host(), o["pcm"], f"H={H} k={k} half={half} T={T} fused={fused}")
        assert not np.array_equal(z[1], z[0]), "the one-launch-per-layer kernel did not engage at this width"
    syn.debug_set("flow_fused", 1)
    idsb = [sb.synthetic_ids(t, cfg.vocab, salt=t) for t in (11, 5, 19)]
    n = syn.run_batch(idsb)
    pcm = syn.pcm_host()
    off = 0
    for i, a in enumerate(idsb):
        o = port.infer_ids(a, 0, 1.0)
        assert_pcm_close(pcm[off:off + int(n[i])], o["pcm"], f"H={H} batch member {i}")
        off += int(n[i])
    syn.close()


@pytest.mark.parametrize("kind,size", [("ms_hifigan_sdp", "tiny"), ("mbb_fix", "tiny"), ("hifigan_sdp", "full")])
def test_launch_ahead_for_packed_batches(kind, size):
    """SURVEY 8 f3, finished in round 5: a packed batch whose members the engine has ALL served before (per-utterance memo: ids, speaker, length
    scale -> frame count) is launched without the host waiting for the frame counts -- the geometry tables come from the memo, the counts the
    durations kernel writes are checked after the run's one synchronisation.  Same geometry, same launches: the PCM is bit-identical to the
    waiting path's, which is pinned to the oracle.  A batch with one new member waits; a wrong memo entry (test mode: keyed by length alone) is
    detected and the two stages are repeated."""
    cfg = sb.full_cfg(kind) if size == "full" else sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 17)
    lens = [23, 9, 31, 9, 16] if size == "tiny" else [70, 64, 90]
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=3 + i) for i, t in enumerate(lens)]
    nspk = 5 if cfg.is_ms else 1
    sid = [i % nspk for i in range(len(ids))]
    ls = [1.0, 1.1, 0.9, 1.0, 1.2][:len(ids)]
    syn = engine.Synthesizer(blob)
    syn.set_profiling(True)
    n1 = syn.run_batch(ids, sid, ls).copy()
    first = syn.pcm_host().copy()
    assert syn.profile()["launch_ahead"] == 0
    if size == "tiny":
        port = pyref.PortModel(blob)
        off = 0
        for i in range(len(ids)):
            o = port.infer_ids(ids[i], sid[i], ls[i])
            assert_pcm_close(first[off:off + int(n1[i])], o["pcm"], f"{kind} member {i} (waiting path) vs the oracle")
            off += int(n1[i])
    for _ in range(2):
        n2 = syn.run_batch(ids, sid, ls)
        p = syn.profile()
        assert p["launch_ahead"] == 1 and p["launch_ahead_misses"] == 0 and p["ms_sync_wait_host"] < 0.05, p
        assert np.array_equal(n2, n1) and np.array_equal(syn.pcm_host(), first), "launch-ahead changed a sample of the batch"
    # a member the engine has not served: the whole batch waits; a sub-batch of known members runs ahead
    ids_new = ids[:-1] + [sb.synthetic_ids(lens[-1], cfg.vocab, salt=77)]
    fresh = engine.Synthesizer(blob)
    fresh.run_batch(ids_new, sid, ls)
    want_new = fresh.pcm_host().copy()
    syn.run_batch(ids_new, sid, ls)
    assert syn.profile()["launch_ahead"] == 0 and np.array_equal(syn.pcm_host(), want_new)
    fresh.run_batch(ids[1:], sid[1:], ls[1:])
    want_sub = fresh.pcm_host().copy()
    syn.run_batch(ids[1:], sid[1:], ls[1:])
    assert syn.profile()["launch_ahead"] == 1 and np.array_equal(syn.pcm_host(), want_sub)
    # a wrong memo entry (keyed by the phoneme count alone): detected after the run, flow + decoder repeated
    syn.debug_set("launch_ahead", 2)
    syn.run_batch(ids, sid, ls)
    assert syn.profile()["launch_ahead"] == 0
    ls2 = [v * 1.25 for v in ls]
    fresh.run_batch(ids, sid, ls2)
    want2 = fresh.pcm_host().copy()
    syn.run_batch(ids, sid, ls2)
    assert syn.profile()["launch_ahead_misses"] >= 1 and np.array_equal(syn.pcm_host(), want2), "the repeated batch differs from a waiting run"
    fresh.close()
    syn.close()


def test_multi_device_rccl_gather_with_a_one_rank_communicator():
    """The native RCCL path of sts_multi (ncclCommInitAll, counts by ncclAllGather, gather buffer on device 0, ONE download) on what a
    one-GPU box allows with the REAL librccl: a single-rank communicator.  Same PCM as the per-device download, as a plain engine and
    as the oracle; the automatic mode = downloads (the RCCL gather is opt-in).  Three ranks: the next test."""
    cfg = sb.tiny_cfg("hifigan_sdp")
    blob = sb.make_blob(cfg, 11)
    lens = [9, 33, 5, 21]
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=i) for i, t in enumerate(lens)]
    syn = engine.Synthesizer(blob)
    want = [syn.infer_ids(a, 0, 1.0) for a in ids]
    syn.close()
    port = pyref.PortModel(blob)
    for i, (a, w) in enumerate(zip(ids, want)):
        assert_pcm_close(w, port.infer_ids(a, 0, 1.0)["pcm"], f"plain engine vs oracle, utterance {i}")
    md = engine.MultiDevice(blob, [0], gather="rccl")
    assert md.gather_mode() == "rccl"
    for _ in range(2):                      # twice: buffers and communicator are reused
        got = md.infer_batch(ids)
        for i, (g, w) in enumerate(zip(got, want)):
            assert_pcm_close(g, w, f"RCCL gather, utterance {i}")      # (a packed batch may pick other tiles than a single call)
    md.close()
    md = engine.MultiDevice(blob, [0, 0])
    assert md.gather_mode() == "download"
    got2 = md.infer_batch(ids)
    md.close()
    md = engine.MultiDevice(blob, [0], gather="download")
    got1 = md.infer_batch(ids)
    md.close()
    for i, (g, w) in enumerate(zip(got2, want)):
        assert_pcm_close(g, w, f"download, utterance {i}")
    for g, w in zip(got, got1):
        assert np.array_equal(g, w)          # same shard, same engine path: the gather itself must not change a sample


def test_prepared_batch_and_host_view_return_the_copied_samples():
    """The bench step's host path (bench.py step()): run_batch on a PreparedBatch (argument arrays built once) and the zero-copy view of
    the engine's pinned download buffer (sts_pcm_host_view) must hand over exactly the samples of the copying calls, run after run, and
    the view must refuse to exist when the run did not download (sts_set_host_pcm(0))."""
    g, cfg, blob, utts, stride = load_golden_v2([p for p in golden_files_v2("amp_") if p.endswith("amp_hifigan_fix_sat10.npz")][0])
    syn = engine.Synthesizer(blob)
    utts = [utts[0], utts[0]]                # the same utterance twice: a batch
    ids = [u[1] for u in utts]
    sid = [u[2] for u in utts]
    ls = [u[3] for u in utts]
    want = syn.infer_batch(ids, sid, ls)
    prep = syn.prepare(ids, sid, ls)
    for _ in range(3):
        n_out = syn.run_batch(prep)
        view = syn.pcm_host(copy=False)
        assert not view.flags.writeable and view.size == int(n_out.sum())
        assert np.array_equal(view, np.concatenate(want)) and np.array_equal(syn.pcm_host(), view)
    p = syn.profile()
    assert p["us_host_setup"] > 0 and p["us_host_enqueue"] >= p["us_host_setup"] and p["us_host_tail"] >= 0
    # one utterance: the decoder's last kernel writes the PCM into the pinned host buffer itself (STS_DBG_PCM_DIRECT); the download
    # behind the run must hand over the same samples, and a device-side copy-out must still work from either
    one = syn.prepare(ids[:1], sid[:1], ls[:1])
    import torch
    base = None
    for direct in (0, 1, 0, 1):
        syn.debug_set("pcm_direct", direct)
        n1 = syn.run_batch(one)
        got = syn.pcm_host()
        if base is None:
            base = got
            assert_pcm_close(base, utts[0][5], "one utterance vs the reference")
        assert got.size == int(n1[0]) and np.array_equal(got, base) and np.array_equal(syn.pcm_host(copy=False), base), f"pcm_direct={direct}"
        dev = torch.empty(int(n1[0]), dtype=torch.int16, device="cuda")
        syn.pcm_to_device_ptr(dev.data_ptr(), dev.numel())
        assert np.array_equal(dev.cpu().numpy(), base)
    syn.set_host_pcm(False)
    syn.run_batch(prep)
    with pytest.raises(engine.StsError):
        syn.pcm_host(copy=False)
    assert np.array_equal(syn.pcm_host(), np.concatenate(want))          # (the copying call downloads by itself)
    syn.close()


def test_launch_ahead_returns_the_samples_of_the_waiting_path():
    """SURVEY 8 f3 / VERDICT r03 item 4, re-cut in round 5: a one-utterance call the engine has served before no longer waits
    for the data-dependent frame count between the duration predictor and the flow.  The reference's noise scale is 0, so the count is a
    pure function of (ids, speaker, length scale): the engine keeps a memo of its last 64 requests under a hash of those inputs, enqueues
    flow + decoder for the remembered count and lets the kernels read the real one from device memory.  Both paths size buffers and
    dispatch by the 64-frame bucket, so the samples are BIT-IDENTICAL to the waiting path's -- and pinned to the reference
    (tests/golden/full_hifigan_sdp_T128.npz).  Any other request (another length scale, other ids of the same length) takes the waiting
    path: nothing is over-provisioned and no result depends on the engine's history.  A count outside the predicted bucket (possible only
    under a hash collision; provoked here with the test mode that keys the memo by the phoneme count alone) is detected, counted and answered by
    repeating flow + decoder the waiting way -- again bit-identical."""
    g, cfg, blob, utts, stride = load_golden_v2([p for p in golden_files_v2("full_") if p.endswith("full_hifigan_sdp_T128.npz")][0])
    _, ids, sid_u, ls_u, dur_u, pcm_ref, wave_ref = utts[0]
    syn = engine.Synthesizer(blob)
    syn.set_profiling(True)
    first = syn.infer_ids(ids, sid_u, ls_u)
    p1 = syn.profile()
    assert p1["launch_ahead"] == 0 and (syn.durations(len(ids)) == dur_u).all()
    assert_pcm_close(first, pcm_ref, "waiting path vs the reference")
    for _ in range(3):
        again = syn.infer_ids(ids, sid_u, ls_u)
        p2 = syn.profile()
        assert p2["launch_ahead"] == 1 and p2["launch_ahead_misses"] == 0
        assert p2["ms_sync_wait_host"] < 0.02, p2["ms_sync_wait_host"]        # (the count is read after the run's one stream synchronisation)
        assert np.array_equal(again, first), "launch-ahead changed a sample"
        assert (syn.durations(len(ids)) == dur_u).all()
        assert abs(p2["flops_decoder_mfma"] - p1["flops_decoder_mfma"]) <= 1e-6 * p1["flops_decoder_mfma"]      # booked for the real count
        assert p2["frames"] == p1["frames"] and p2["samples"] == first.size
    syn.debug_set("launch_ahead", 0)
    assert np.array_equal(syn.infer_ids(ids, sid_u, ls_u), first) and syn.profile()["launch_ahead"] == 0
    syn.debug_set("launch_ahead", 1)
    # the same ids with another length scale, and other ids of the same length: requests the engine has not served -> the waiting path
    fresh = engine.Synthesizer(blob)
    want_long = fresh.infer_ids(ids, sid_u, 1.3)
    ids2 = sb.synthetic_ids(len(ids), cfg.vocab, salt=5)
    want_other = fresh.infer_ids(ids2, sid_u, ls_u)
    fresh.close()
    got_long = syn.infer_ids(ids, sid_u, 1.3)
    p3 = syn.profile()
    assert p3["launch_ahead"] == 0 and p3["launch_ahead_misses"] == 0 and want_long.size > first.size
    assert np.array_equal(got_long, want_long), "a new request differs from a fresh engine's answer"
    got_other = syn.infer_ids(ids2, sid_u, ls_u)
    assert syn.profile()["launch_ahead"] == 0 and np.array_equal(got_other, want_other)
    # ... and each of the three, once served, runs ahead with its own count: bit-identical whatever came before
    for a_ids, a_ls, want in ((ids, ls_u, first), (ids, 1.3, want_long), (ids2, ls_u, want_other), (ids, ls_u, first)):
        got = syn.infer_ids(a_ids, sid_u, a_ls)
        assert syn.profile()["launch_ahead"] == 1 and syn.profile()["launch_ahead_misses"] == 0
        assert np.array_equal(got, want), "a remembered request changed a sample"
    # a collision (test mode: the memo keyed by the phoneme count alone): the count falls outside the predicted bucket in either direction
    syn.debug_set("launch_ahead", 2)
    assert np.array_equal(syn.infer_ids(ids, sid_u, ls_u), first) and syn.profile()["launch_ahead"] == 0
    got_long = syn.infer_ids(ids, sid_u, 1.3)                      # predicted 668 frames, needs more
    assert syn.profile()["launch_ahead_misses"] == 1 and np.array_equal(got_long, want_long), "the repeated run differs from a waiting run"
    back = syn.infer_ids(ids, sid_u, ls_u)                        # predicted the longer count, needs fewer
    assert syn.profile()["launch_ahead_misses"] == 2 and np.array_equal(back, first), "an over-provisioned run was not repeated"
    again = syn.infer_ids(ids, sid_u, ls_u)
    assert syn.profile()["launch_ahead"] == 1 and syn.profile()["launch_ahead_misses"] == 2 and np.array_equal(again, first)
    syn.close()
    # every decoder family / duration predictor, small models, against the oracle
    for kind in ("mbb_fix", "ms_sdp", "istft_fix", "ms_hifigan_sdp", "odd"):
        cfg = sb.tiny_cfg(kind)
        blob = sb.make_blob(cfg, 31)
        port = pyref.PortModel(blob)
        syn = engine.Synthesizer(blob)
        syn.set_profiling(True)
        for T in (23, 9, 23, 23, 9):
            ids = sb.synthetic_ids(T, cfg.vocab, salt=T)
            sid = 1 if cfg.is_ms else 0
            o = port.infer_ids(ids, sid, 1.1)
            got = syn.infer_ids(ids, sid, 1.1)
            assert (syn.durations(T) == o["durations"]).all()
            assert_pcm_close(got, o["pcm"], f"{kind} T={T} (launch_ahead={syn.profile()['launch_ahead']})")
        assert syn.profile()["launch_ahead"] == 1
        syn.close()


def test_multi_device_rccl_gather_with_three_emulated_ranks():
    """The N > 1 protocol of sts_multi's RCCL gather on a one-GPU box: THREE communicator ranks (device 0 listed three times) against
    tests/fake_rccl/libfake_rccl.so -- a host-side stand-in for the nine nccl* entry points multi.hip resolves, selected with
    sts_multi_set_rccl_library (it has to be the process's first RCCL provider, hence the subprocess).  Pairing of sends and receives,
    zero-count ranks, a failing shard, buffer regrowth, an injected ncclRecv failure (all communicators aborted, bounded wait, the
    handle continues with downloads); every PCM against the plain engine AND the oracle.  Hardware N > 1 stays unmeasured."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.run(["make", "-C", os.path.dirname(fake)], check=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fake_rccl", "three_ranks.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    failed = [c for c in res["checks"] if not c["ok"]]
    assert res["ok"] and not failed, failed
    assert len(res["checks"]) >= 12


def test_bench_multi_native_drives_the_library_gather_with_three_emulated_ranks():
    """`bench.py --gpus 3 --multi native`: one process, sts_multi_create_ex(STS_MULTI_RCCL) -- here with device 0 listed three
    times against tests/fake_rccl (the only N > 1 a one-GPU box can offer).  The line must carry the communicator size as the RCCL library reports
    it (ncclCommCount), a gather time, and the samples of all three ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = os.path.join(root, "tests", "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(fake):
        subprocess.run(["make", "-C", os.path.dirname(fake)], check=True)
    env = dict(os.environ, STS_TEST_HOOKS="1", STS_BENCH_RCCL_LIB=fake, STS_BENCH_TINY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--multi", "native", "--share-gpu", "--steps", "3", "--warmup", "1",
                        "--batch", "2", "--phonemes", "21"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    mg = out["multi_gpu"]
    assert out["n_gpus"] == 3 and mg["gather_mode"] == "rccl" and mg["rccl_ranks"] == 3, mg
    assert mg["utterances_per_device"] == [2, 2, 2] and all(s > 0 for s in mg["samples_per_device"]) and mg["gather_ms_per_step_rank0"] > 0
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 3 - sum(mg["samples_per_device"])) < 1.0


def test_concurrent_engines_on_one_gpu_return_the_reference_result():
    """Three engines on ONE GPU driven from three host threads at the same time (what sts_pool does), twenty calls each, on the bench's
    own utterance: every call must return, bit for bit, what the engine returns when it runs alone -- and THAT result is pinned to the
    reference's output for this utterance (tests/golden/full_hifigan_sdp_T128.npz, made by the compiled reference), not to the HIP
    path itself."""
    import threading
    g, cfg, blob, utts, stride = load_golden_v2([p for p in golden_files_v2("full_") if p.endswith("full_hifigan_sdp_T128.npz")][0])
    _, ids, sid_u, ls_u, dur_u, pcm_ref, wave_ref = utts[0]
    engines = [engine.Synthesizer(blob) for _ in range(3)]
    want = engines[0].infer_ids(ids, sid_u, ls_u)
    assert (engines[0].durations(len(ids)) == dur_u).all()
    assert_pcm_close(want, pcm_ref, "single engine vs the reference's PCM")
    bad, err = [], []

    def work(k):
        try:
            for it in range(20):
                if not np.array_equal(engines[k].infer_ids(ids, sid_u, ls_u), want):
                    bad.append((k, it))
        except Exception as ex:      # noqa: BLE001
            err.append(ex)
    th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in th), "concurrent engines hang"
    assert not err, err
    for k in range(3):
        assert np.array_equal(engines[k].infer_ids(ids, sid_u, ls_u), want), "an engine is left in a bad state"
    for e in engines:
        e.close()
    assert not bad, f"concurrent engines: {len(bad)} of 60 calls differ from the single-engine result {bad[:4]}"
