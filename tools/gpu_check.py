"""Diagnostic: per-stage parity of the HIP engine against the oracle (run on a GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from summertts_amd import synth_blob as sb
from summertts_amd import engine as eng
from oracle import pyref


def cmp(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.shape != b.shape:
        return f"SHAPE {a.shape} vs {b.shape}"
    return "max %.2e rms %.2e (ref rms %.2e)" % (np.abs(a - b).max(), np.sqrt(((a - b) ** 2).mean()), np.sqrt((b ** 2).mean()))


def conv_checks():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(0)
    cases = [  # Cin, Cout, k, pad, dil, L, stride_t, depthwise
        (64, 64, 3, 1, 1, 300, 0, False), (32, 32, 11, 25, 5, 700, 0, False), (96, 192, 1, 0, 1, 77, 0, False),
        (64, 128, 5, 2, 1, 129, 0, False), (48, 40, 7, 3, 1, 200, 0, False), (20, 24, 3, 3, 3, 50, 0, False),
        (64, 32, 8, 2, 1, 100, 4, False), (128, 64, 16, 4, 1, 65, 8, False), (24, 12, 7, 2, 1, 33, 3, False),
        (16, 16, 3, 9, 9, 120, 0, True), (64, 72, 7, 3, 1, 500, 0, False),
    ]
    for (ci, co, k, pad, dil, L, st, dw) in cases:
        x = rng.standard_normal((ci, L)).astype(np.float32)
        w = (rng.standard_normal((co, k, 1 if dw else ci)) / np.sqrt(k * (1 if dw else ci))).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        xt = torch.from_numpy(x)[None].double()
        if st:
            ref = F.conv_transpose1d(xt, torch.from_numpy(w).permute(2, 0, 1).double(), torch.from_numpy(b).double(), stride=st, padding=pad)[0].numpy()
        elif dw:
            ref = F.conv1d(xt, torch.from_numpy(w).permute(0, 2, 1).double(), torch.from_numpy(b).double(), padding=pad, dilation=dil, groups=ci)[0].numpy()
        else:
            ref = F.conv1d(xt, torch.from_numpy(w).permute(0, 2, 1).double(), torch.from_numpy(b).double(), padding=pad, dilation=dil)[0].numpy()
        modes = [1] if (dw or ci < 32 or co < 32) else [1, 0, 2, 3, 4, 5, 6, 7]
        for mode in modes:
            try:
                y = eng.debug_conv1d(x, w, b, pad, dil, st, dw, mode=mode)
                print("conv", (ci, co, k, pad, dil, L, st, dw), "mode", mode, cmp(y, ref))
            except Exception as e:
                print("conv", (ci, co, k, pad, dil, L, st, dw), "mode", mode, "ERR", e)


def model_checks(kinds, T=23, mode=0):
    for kind in kinds:
        cfg = sb.tiny_cfg(kind)
        blob = sb.make_blob(cfg, 1234)
        ids = sb.synthetic_ids(T, cfg.vocab)
        p = pyref.PortModel(blob)
        po = p.infer_ids(ids, sid=1, length_scale=1.1, taps=True)
        s = eng.Synthesizer(blob)
        s.set_conv_mode(mode)
        s.set_record_taps(True)
        s.set_forced_durations(po["durations"])
        n = s.run_batch([ids], [1], [1.1])
        pcm = s.pcm_host()
        print(kind, "mode", mode, "samples", n, po["wave"].size)
        for k in ["x_enc", "m", "logw", "z_p", "z", "wave"]:
            try:
                t = s.tap(k)
                r = po[k] if k != "wave" else po["wave"][None]
                print("   ", k, cmp(t, r))
            except Exception as e:
                print("   ", k, "ERR", e)
        if pcm.size == po["pcm"].size:
            d = np.abs(pcm.astype(int) - po["pcm"].astype(int))
            print("    pcm maxdiff", d.max(), "nz", np.count_nonzero(d), "/", d.size)
        # free-running durations
        s.set_record_taps(False)
        s.run_batch([ids], [1], [1.1])
        dur = s.durations(T)
        print("    durations equal:", bool((dur == po["durations"]).all()), dur.tolist()[:8], po["durations"].tolist()[:8])
        s.close()


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("conv", "all"):
        conv_checks()
    if what in ("model", "all"):
        kinds = ["hifigan_fix", "hifigan_sdp", "mbb_fix", "ms_sdp", "istft_fix", "ms_hifigan_sdp", "ms_hifigan_fix", "odd"]
        model_checks(kinds, mode=0)
        model_checks(["hifigan_sdp", "ms_hifigan_sdp"], mode=1)
