"""CPU suite: ``bench.py --gpus N`` launches N ranks itself (no torchrun), shards the utterances, gathers the PCM
(gloo here, RCCL on the GPU node) and prints ONE JSON line with n_gpus = N.  The engine is a stub (tests/stub_engine.py):
this covers the launcher and the distributed plumbing, not the kernels."""
import json
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT


def _run(extra, env_extra=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env["STS_BENCH_ENGINE"] = "stub_engine"
    env["PYTHONPATH"] = os.path.join(ROOT, "tests") + os.pathsep + env.get("PYTHONPATH", "")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    if "WORLD_SIZE" not in (env_extra or {}):
        assert len(lines) == 1, p.stdout      # self-launched: the parent relays exactly one stdout line, the JSON
    return json.loads(lines[-1])              # under an external launcher backends may print banners first: JSON is LAST


def test_gpus_2_spawns_two_ranks_and_gathers():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--ragged"])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak"
    assert out["config"]["launched_by"].startswith("bench.py")
    mg = out["multi_gpu"]
    lens = np.random.default_rng(1234).integers(64, 257, size=4).tolist()
    assert sum(mg["utterances_per_rank"]) == 4 and len(mg["samples_per_rank"]) == 2
    assert sum(mg["samples_per_rank"]) == 3 * 100 * sum(lens)               # every rank's samples, all steps
    assert mg["gathered_samples_rank0"] == 3 * 100 * sum(lens)              # ... and all of them arrived on rank 0
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 3 - sum(mg["samples_per_rank"])) < 1.0
    # round 6: the same invocation also measures the library's native gather (rank 0, in-process, after the distributed run)
    nat = out["multi_native"]
    assert "error" not in nat, nat
    assert nat["n_gpus"] == 2 and nat["rccl_ranks"] == 2 and nat["gather_mode"] == "rccl" and nat["value"] > 0 and nat["gather_ms_per_step"] > 0
    assert sum(nat["samples_per_device"]) == 3 * 100 * sum(lens)


def test_torchrun_style_environment_is_respected():
    # started "by a launcher" as the only rank of a world of 1 with the gather path forced: no self-spawn
    out = _run(["--gpus", "1", "--steps", "2", "--warmup", "0"],
               {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "STS_BENCH_FORCE_DIST": "1", "MASTER_PORT": "29641"})
    assert out["n_gpus"] == 1 and out["multi_gpu"]["gathered_samples_rank0"] == 2 * 100 * 128


def test_uneven_shards_with_an_empty_rank():
    # 3 ranks, batch such that one rank may hold fewer utterances: the gather must not hang
    out = _run(["--gpus", "3", "--steps", "2", "--warmup", "0", "--batch", "1", "--phonemes", "10"])
    assert out["n_gpus"] == 3 and sum(out["multi_gpu"]["samples_per_rank"]) == 2 * 3 * 1000


def test_config_presets_name_the_baseline_config_they_form():
    """`--config N` sets workload / batch / ragged for BASELINE.json's configs[N] and the line says which config it is;
    ad-hoc flags that form no config are labelled custom (config.workload was hard-coded to configs[1])."""
    out = _run(["--gpus", "2", "--config", "3", "--steps", "1", "--warmup", "0"])
    assert out["n_gpus"] == 2 and out["config"]["baseline_config"] == 3 and out["config"]["workload"].startswith("configs[3]")
    assert out["config"]["global_batch"] == 64 and sum(out["multi_gpu"]["utterances_per_rank"]) == 64
    out = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29643"})
    assert out["config"]["baseline_config"] == 1 and out["config"]["workload"].startswith("configs[1]")
    out = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--batch", "5", "--phonemes", "33"],
               {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_PORT": "29644"})
    assert out["config"]["baseline_config"] is None and out["config"]["workload"].startswith("custom")


def test_multi_native_runs_in_one_process_and_reports_the_communicator():
    """`bench.py --gpus N --multi native`: ONE process drives all devices through sts_multi's own RCCL gather; the line
    carries the communicator size as RCCL reports it and rank 0's gather time.  Here: the plumbing with the stub engine (the real library
    runs it against tests/fake_rccl on the GPU box: tests/test_parity_gpu.py)."""
    out = _run(["--gpus", "3", "--multi", "native", "--steps", "2", "--warmup", "1", "--batch", "2", "--phonemes", "10"])
    assert out["n_gpus"] == 3 and out["scaling"] == "weak" and out["config"]["launched_by"].startswith("bench.py --multi native")
    mg = out["multi_gpu"]
    assert mg["rccl_ranks"] == 3 and mg["gather_mode"] == "rccl" and sum(mg["utterances_per_device"]) == 6
    assert sum(mg["samples_per_device"]) == 2 * 6 * 1000 and abs(mg["gather_ms_per_step_rank0"] - 0.25) < 1e-9
    assert abs(out["value"] * out["ms_per_step"] * 1e-3 * 2 - 2 * 6 * 1000) < 1.0
