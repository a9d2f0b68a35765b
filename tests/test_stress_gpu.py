"""Create / destroy stress of the engine: the one segmentation fault ever seen in this repository was inside
``sts_destroy`` with eight processes sharing the GPU (profiles/r05_ab_log.md).  Model construction and teardown touch the driver's
virtual-memory calls (model.hip Store), host-mapped pinned buffers and stream / event objects; these tests hammer exactly that:
many short-lived engines, from several processes at once and from several threads of one process, every result checked against the
oracle's PCM so that a torn-down engine's memory being reused under a live one would show.
Mirrors the lifetime contract of /root/reference/src/models/SynthesizerTrn.cpp:403-415 (the destructor frees everything the constructor made)."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from summertts_amd import engine, synth_blob as sb
n, seed, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
cfg = sb.tiny_cfg(kind)
blob = sb.make_blob(cfg, 1234)
ids = sb.synthetic_ids(9 + seed %% 5, cfg.vocab, salt=seed)
want = None
for it in range(n):
    syn = engine.Synthesizer(blob, device=0)
    pcm = syn.infer_ids(ids, 0, 1.0).copy()
    syn.close()
    if want is None:
        want = pcm
    assert pcm.shape == want.shape and (pcm == want).all(), f"iteration {it}: PCM changed between engine lifetimes"
print("ok", n, int(want.size))
"""


@pytest.mark.gpu
def test_create_infer_destroy_from_eight_processes_at_once(tmp_path):
    script = tmp_path / "stress_worker.py"
    script.write_text(WORKER % {"root": ROOT})
    kinds = ["hifigan_sdp", "mbb_fix", "ms_sdp", "istft_fix"]
    procs = [subprocess.Popen([sys.executable, str(script), "50", str(p), kinds[p % len(kinds)]], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for p in range(8)]
    fails = []
    for p, pr in enumerate(procs):
        try:
            out, err = pr.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            pr.kill()
            out, err = pr.communicate()
            fails.append((p, "timeout", err[-2000:]))
            continue
        if pr.returncode != 0 or not out.strip().startswith("ok 50"):
            fails.append((p, pr.returncode, (out + err)[-2000:]))
    assert not fails, f"{len(fails)} of 8 stress processes failed: {fails}"


@pytest.mark.gpu
def test_create_infer_destroy_from_four_threads_of_one_process():
    from summertts_amd import engine, synth_blob as sb
    from oracle import pyref
    cfg = sb.tiny_cfg("hifigan_sdp")
    blob = sb.make_blob(cfg, 1234)
    errors = []

    def worker(t):
        try:
            ids = sb.synthetic_ids(8 + t, cfg.vocab, salt=t)
            want = pyref.PortModel(blob).infer_ids(ids, 0, 1.0)["pcm"]
            for it in range(25):
                syn = engine.Synthesizer(blob, device=0)
                pcm = syn.infer_ids(ids, 0, 1.0)
                d = int(np.abs(pcm.astype(np.int32) - want.astype(np.int32)).max()) if pcm.size == want.size else -1
                syn.close()
                if d < 0 or d > 1:
                    errors.append((t, it, d))
                    return
        except Exception as e:   # noqa: BLE001
            errors.append((t, "exception", repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
