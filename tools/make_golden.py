#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the REAL reference (oracle/_ref = /root/reference compiled in
place).  Run where /root/reference exists; the fixtures travel to machines where it does not.
Each fixture: the synthetic-blob recipe (kind, seed, sha256 of the blob bytes), the inputs
(ids, sid, length_scale) and the reference's outputs (durations, float waveform, int16 PCM, taps m/z)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref                      # noqa: E402
from summertts_amd import synth_blob as sb    # noqa: E402

CASES = [  # kind, T, sid, length_scale
    ("hifigan_sdp", 17, 0, 1.0), ("hifigan_fix", 21, 0, 1.15), ("mbb_fix", 13, 0, 1.0), ("ms_sdp", 11, 0, 0.9),
    ("istft_fix", 19, 0, 1.0), ("ms_hifigan_sdp", 15, 3, 1.1), ("ms_hifigan_fix", 16, 2, 1.0), ("odd", 9, 0, 1.3),
    ("hifigan_fix", 5, 0, 1.0),
]


def main():
    pyref.build(port=False, ref=True)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for kind, T, sid, ls in CASES:
        cfg = sb.tiny_cfg(kind)
        blob = sb.make_blob(cfg, 1234)
        ids = sb.synthetic_ids(T, cfg.vocab, salt=T)
        ref = pyref.RefModel(blob)
        assert ref.consumed == blob.size
        o = ref.infer_ids(ids, sid, ls, taps=True)
        name = f"{kind}_T{T}.npz"
        np.savez_compressed(os.path.join(out_dir, name), kind=kind, seed=1234, blob_sha256=hashlib.sha256(blob.tobytes()).hexdigest(),
                            ids=ids, sid=sid, length_scale=np.float32(ls), durations=o["durations"], wave=o["wave"],
                            pcm=o["pcm"], m=o["m"], z=o["z"], logw=o["logw"])
        print(name, "frames", int(o["durations"].sum()), "samples", o["wave"].size)


if __name__ == "__main__":
    main()
