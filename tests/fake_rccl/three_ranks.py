#!/usr/bin/env python3
"""Driver of tests/test_parity_gpu.py::test_multi_device_rccl_gather_with_three_emulated_ranks (runs in its own process: the nccl*
provider of a process is chosen once).  sts_multi's RCCL gather with THREE ranks on one GPU against tests/fake_rccl/libfake_rccl.so:
pairing, a zero-count rank, a failing shard, buffer regrowth, an injected ncclRecv failure (communicators aborted, no hang, the handle
continues with downloads).  Every PCM is compared with the plain engine and with the oracle (C restatement).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref                                # noqa: E402
from summertts_amd import engine, synth_blob as sb      # noqa: E402

FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
out = {"checks": []}


def check(name, ok, detail=""):
    out["checks"].append({"name": name, "ok": bool(ok), "detail": str(detail)})


def lsb(a, b):
    return int(np.abs(a.astype(np.int32) - b.astype(np.int32)).max()) if a.shape == b.shape else 10 ** 6


cfg = sb.tiny_cfg("ms_hifigan_sdp")
blob = sb.make_blob(cfg, 21)
port = pyref.PortModel(blob)
syn = engine.Synthesizer(blob)
try:                                   # the hook is gated: without STS_TEST_HOOKS=1 in the environment the library refuses it
    os.environ.pop("STS_TEST_HOOKS", None)
    engine.MultiDevice.set_rccl_library(FAKE, allow_repeated_devices=True)
    check("the test hook is refused without STS_TEST_HOOKS=1", False)
except engine.StsError:
    check("the test hook is refused without STS_TEST_HOOKS=1", True)
os.environ["STS_TEST_HOOKS"] = "1"
engine.MultiDevice.set_rccl_library(FAKE, allow_repeated_devices=True)
md = engine.MultiDevice(blob, [0, 0, 0], gather="rccl")
check("gather mode is rccl with three emulated ranks", md.gather_mode() == "rccl" and md.device_count() == 3)


def batch(lens, salt0=0):
    ids = [sb.synthetic_ids(t, cfg.vocab, salt=salt0 + i) for i, t in enumerate(lens)]
    sid = [(salt0 + i) % cfg.spk_num for i in range(len(lens))]
    ls = [1.0 + 0.05 * (i % 4) for i in range(len(lens))]
    return ids, sid, ls


def run_and_compare(tag, lens, salt0=0):
    ids, sid, ls = batch(lens, salt0)
    t0 = time.time()
    got = md.infer_batch(ids, sid, ls)
    dt = time.time() - t0
    worst_eng, worst_orc = 0, 0
    for a, s, l, g in zip(ids, sid, ls, got):
        worst_eng = max(worst_eng, lsb(g, syn.infer_ids(a, s, l)))
        worst_orc = max(worst_orc, lsb(g, port.infer_ids(a, s, l)["pcm"]))
    check(f"{tag}: PCM within 1 LSB of the plain engine and of the oracle", worst_eng <= 1 and worst_orc <= 1, f"engine {worst_eng} oracle {worst_orc} LSB, {dt:.2f} s")


run_and_compare("seven utterances on three ranks", [9, 31, 5, 17, 2, 24, 11])
run_and_compare("the same batch again (buffers and communicators reused)", [9, 31, 5, 17, 2, 24, 11])
run_and_compare("two utterances on three ranks (a zero-count rank takes part in every collective)", [13, 6], salt0=40)
run_and_compare("one utterance (two zero-count ranks)", [21], salt0=50)
run_and_compare("a larger batch (gather buffers regrow)", [40, 55, 33, 61, 47, 38, 52, 44, 36, 58, 41, 49], salt0=60)
# a failing shard: the bad utterance's rank reports count 0 and still joins; the call fails, nothing hangs, the handle stays usable
ids, sid, ls = batch([9, 31, 5, 17], salt0=0)
ids[1] = np.asarray([0, cfg.vocab + 3, 1], np.int32)
t0 = time.time()
try:
    md.infer_batch(ids, sid, ls)
    check("a bad phoneme id fails the call", False)
except engine.StsError as ex:
    check("a bad phoneme id fails the call (no hang)", time.time() - t0 < 30, str(ex)[:120])
check("the communicators survive a failing shard", md.gather_mode() == "rccl")
run_and_compare("after the failing shard", [9, 31, 5, 17, 2, 24, 11])
# an RCCL-level failure on rank 0 (the next ncclRecv returns an error): every communicator is aborted, the senders return, the call
# fails within the bounded wait, and the handle continues with per-device downloads
os.environ["FAKE_RCCL_FAIL_RECV"] = "1"        # (the library counts ncclRecv calls from the first time the variable is seen)
ids, sid, ls = batch([9, 31, 5, 17, 2, 24, 11])
t0 = time.time()
try:
    md.infer_batch(ids, sid, ls)
    check("an ncclRecv failure fails the call", False)
except engine.StsError as ex:
    check("an ncclRecv failure fails the call within the bounded wait (no hang)", time.time() - t0 < 30, f"{time.time() - t0:.2f} s: {str(ex)[:120]}")
del os.environ["FAKE_RCCL_FAIL_RECV"]
check("after the abort the handle reports the download path", md.gather_mode() == "download")
run_and_compare("after the abort (per-device downloads)", [9, 31, 5, 17, 2, 24, 11])
md.close()
syn.close()
out["ok"] = all(c["ok"] for c in out["checks"])
print(json.dumps(out))
