#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X SummerTTS acoustic+vocoder engine.

Metric (BASELINE.json): audio samples/sec/GPU (x real-time at 16 kHz) + p50 infer() latency on
``single_speaker_fast`` for one 128-phoneme utterance (configs[1]).  The reference's model blobs are
not available (``/root/reference/.MISSING_LARGE_BLOBS``), so the workload is a seeded random-weight
blob written in the reference's grammar at the upstream VITS dimensions (``summertts_amd/synth_blob.py``;
SURVEY.md section 8d) and ids[i] = (i*37+11) mod vocab.

A *step* is one pass of the hot path (phoneme ids -> int16 PCM on the host) over one batch.  At N GPUs
every rank runs its own shard of the utterance batch (per-GPU work fixed: weak scaling, no data-path
collective) and the int16 PCM is gathered to rank 0 over RCCL inside the timed step.

``--gpus N`` with N > 1: when the script was not started by torchrun (no WORLD_SIZE in the environment) it
launches the N ranks itself -- one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
127.0.0.1 -- and relays rank 0's JSON line.  Started under ``python -m torch.distributed.run`` it uses the
environment it is given.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: v_mfma_f32_32x32x16_bf16 dense peak (~2.5 PF; 2495 measured)
BF16_PRODUCTS_PER_F32 = 6       # conv_bf3.hip: six bf16 products per fp32 product (operands split into three bf16 terms)
SUSTAINED_F16_TWO_TERM_TFLOPS = 1490.0   # the same for the 12-MFMA sequence of the two-term fp16 form (profiles/r03_f16x2_kloop_decomposition.log / tools/ubench/run_peaks.py)
SUSTAINED_BF16_SPLIT_TFLOPS = 1712.0   # tools/ubench/mfma_bf16_peak.hip on MI355X (profiles/r02_mfma_bf16_peak_ubench.log): what a bare loop of
                                       # the kernel's MFMA sequence sustains on the hi/mid/lo planes of random fp32 data (data-dependent DVFS)
PEAK_HBM_GBS = 8000.0


# ------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` == N ranks, one per GPU
# ------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv, timeout_s: float = 3000.0) -> int:
    """Starts n copies of this script (ranks 0..n-1) and relays rank 0's stdout.  Returns the exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "STS_BENCH_SELF_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    rc = 0
    out0 = ""
    try:
        out0, _ = procs[0].communicate(timeout=timeout_s)
        rc = procs[0].returncode
        for p in procs[1:]:
            p.wait(timeout=120)
            rc = rc or p.returncode
    except subprocess.TimeoutExpired:
        rc = 124
    finally:
        for p in procs:                       # exact PIDs we started, nothing else
            if p.poll() is None:
                p.kill()
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)          # the JSON line stays the last thing on stdout
    return rc


# ------------------------------------------------------------------------------------------------
# CPU baseline (SURVEY.md 8d protocol)
# ------------------------------------------------------------------------------------------------
def _set_omp_threads(n: int) -> bool:
    """Eigen's GEMM parallelizer asks omp_get_max_threads() on every call, so the thread count can be changed in-process."""
    import ctypes
    for name in ("libgomp.so.1", "libgomp.so"):
        try:
            ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL).omp_set_num_threads(int(n))
            return True
        except OSError:
            continue
    return False


def cpu_baseline(blob, vocab, ids, reps: int, sweep):
    """The reference itself (oracle/_ref: the real Eigen path compiled with the reference's flags) or, failing that,
    the C restatement, on this box's host cores: the SAME blob and the SAME phoneme ids as the GPU step, one warm-up
    (pays the OpenMP start-up) then `reps` timed repetitions -> median; the thread count is the best of `sweep`,
    chosen once on a short utterance."""
    from oracle import pyref
    from summertts_amd import synth_blob as sb
    if pyref.have_ref():
        model, kind = pyref.RefModel(blob), "reference"
    else:
        try:
            model, kind = pyref.PortModel(blob), "port"
        except Exception:
            return None, None
    nproc = os.cpu_count() or 1
    probe = sb.synthetic_ids(12, vocab, salt=1)
    model.infer_ids(sb.synthetic_ids(6, vocab), 0, 1.0)            # warm-up: OpenMP thread pool, page-in
    sweep_res = {}
    best_t = None
    if os.environ.get("OMP_NUM_THREADS"):
        best_t = int(os.environ["OMP_NUM_THREADS"])
    else:
        cands = sorted({min(t, nproc) for t in sweep})
        for t in cands:
            if not _set_omp_threads(t):
                break
            model.infer_ids(probe, 0, 1.0)
            t0 = time.perf_counter()
            o = model.infer_ids(probe, 0, 1.0)
            sweep_res[t] = int(o["wave"].size) / (time.perf_counter() - t0)
        if sweep_res:
            best_t = max(sweep_res, key=sweep_res.get)
            _set_omp_threads(best_t)
        else:
            best_t = nproc
    model.infer_ids(probe, 0, 1.0)                                  # warm-up at the chosen thread count
    times, stages, n = [], [], 0
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        out = model.infer_ids(ids, 0, 1.0)
        times.append(time.perf_counter() - t0)
        stages.append(out["times"])
        n = int(out["wave"].size)
    med = float(np.median(times))
    st = stages[int(np.argsort(times)[len(times) // 2])]
    ref_out = {"pcm": out["pcm"], "durations": out["durations"], "wave": out["wave"]} if kind == "reference" else None
    return ref_out, {"value": n / med, "unit": "samples/s", "x_realtime": n / med / 16000.0, "cores": int(best_t), "nproc": nproc,
            "kind": kind, "phonemes": int(len(ids)), "reps": len(times), "seconds_per_rep_median": med,
            "seconds_per_rep_all": [round(t, 3) for t in times],
            "thread_sweep_samples_per_s": {str(k): round(v, 1) for k, v in sweep_res.items()},
            "stage_seconds": {k: round(float(v), 4) for k, v in st.items()},
            "sample": f"{len(times)} x 1 utterance of {len(ids)} phonemes (the GPU step's blob and ids) -> {n} samples, "
                      f"median {med:.2f} s, {best_t} OpenMP threads of {nproc} host CPUs "
                      f"({'reference Eigen path, -O3 -fopenmp -std=c++11' if kind == 'reference' else 'C restatement, OpenMP'})"}


# BASELINE.json configs[1..4] as bench presets (per-GPU share; weights are seeded synthetic stand-ins at the survey's assumed dims)
CONFIG_PRESETS = {
    1: dict(workload="hifigan_sdp", batch=1, phonemes=128, ragged=False,
            label="configs[1]: single_speaker_fast.bin, batch=1, 128-phoneme synthetic input, 1xMI355X"),
    2: dict(workload="hifigan_sdp", batch=32, phonemes=128, ragged=True,
            label="configs[2]: single_speaker_mid.bin, batch=32 utterances of 64-256 phonemes, 1xMI355X"),
    3: dict(workload="ms_hifigan_sdp", batch=32, phonemes=128, ragged=True,
            label="configs[3]: multi_speakers.bin (aishell3-like, 174 speakers), mixed-speaker utterances of 64-256 phonemes, 32 per GPU "
                  "(= the 256-utterance batch at --gpus 8), utterance-sharded, RCCL gather of the int16 PCM"),
    4: dict(workload="mbb_fix", batch=64, phonemes=128, ragged=True,
            label="configs[4]: single_speaker_english_fast.bin (MB-iSTFT decoder: iSTFT + PQMF), batch=64 utterances of 64-256 phonemes, 1xMI355X"),
    # not a BASELINE.json entry: configs[1]'s call shape on the OTHER decoder family (README.md:25,54 of the reference make it likely that
    # single_speaker_fast.bin is an MB-iSTFT model; the headline uses the heavier HiFi-GAN reading) -- measure_config only, key "configs[1]-mbb"
    5: dict(workload="mbb_fix", batch=1, phonemes=128, ragged=False,
            label="configs[1]-mbb: configs[1]'s call shape (batch=1, 128 phonemes) on the MB-iSTFT + PQMF decoder family"),
}


def config_label(args) -> str:
    """Names the BASELINE.json config the flags actually form (or says that they form none)."""
    for n, pz in CONFIG_PRESETS.items():
        if (args.workload == pz["workload"] and args.batch == pz["batch"] and bool(args.ragged) == pz["ragged"]
                and (pz["ragged"] or args.phonemes == pz["phonemes"])):
            if n == 2 and args.gpus > 1:
                continue
            return pz["label"]
    return (f"custom (no BASELINE.json config): '{args.workload}' blob, batch={args.batch}/GPU, "
            + ("64..256 phonemes/utterance (ragged)" if args.ragged else f"{args.phonemes} phonemes/utterance"))


def parity_report(ref_out, gpu_out):
    """GPU output of one timed leg against the REAL reference's output for the same blob and ids (the cpu_baseline leg computed
    it anyway): durations equal?  int16 PCM: max LSB difference and how many samples differ; float waveform RMSE / max-abs."""
    if ref_out is None or gpu_out is None:
        return None
    rp, gp = np.asarray(ref_out["pcm"]).astype(np.int64).ravel(), np.asarray(gpu_out["pcm"]).astype(np.int64).ravel()
    rep = {"durations_equal": bool(np.array_equal(np.asarray(ref_out["durations"]).ravel(), np.asarray(gpu_out["durations"]).ravel())),
           "samples_ref": int(rp.size), "samples_gpu": int(gp.size)}
    if rp.size == gp.size and rp.size:
        d = np.abs(rp - gp)
        rep.update(pcm_max_lsb=int(d.max()), pcm_n_off=int((d > 0).sum()))
        if gpu_out.get("wave") is not None:
            e = np.asarray(gpu_out["wave"], np.float64).ravel() - np.asarray(ref_out["wave"], np.float64).ravel()
            rep.update(wave_rmse=float(np.sqrt((e ** 2).mean())), wave_maxabs=float(np.abs(e).max()))
        rep["ok"] = bool(rep["durations_equal"] and rep["pcm_max_lsb"] <= 1)
    else:
        rep["ok"] = False
    return rep


def measured_mfma_ceiling(products: int = BF16_PRODUCTS_PER_F32):
    """Runs tools/ubench/libsts_ubench.so (a bare loop of conv_bf3.hip's MFMA sequence) in THIS process on operands with the
    statistics of the kernel's own (split fp32 values: three bf16 planes / six products, or the two-term fp16 planes / three
    products) and on constant operands: what the matrix pipe sustains on this box, now (data-dependent DVFS)."""
    import ctypes
    path = os.path.join(ROOT, "tools", "ubench", "libsts_ubench.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        h2 = products == 3 and hasattr(lib, "sts_ubench_mfma_f16")
        fn = lib.sts_ubench_mfma_f16 if h2 else lib.sts_ubench_mfma_bf16
        fn.restype = ctypes.c_double
        fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        n = 40000 if h2 else 20000                                  # ~20 ms
        split = float(fn(2, 512, n))
        const = float(fn(0, 512, n))
        if split <= 0:
            return None
        kind = "fp16" if h2 else "bf16"
        return {f"{kind}_tflops_split_fp32_operands": split, f"{kind}_tflops_constant_operands": const,
                "tflops_fp32_equivalent": split / products,
                "source": f"tools/ubench/mfma_bf16_peak.hip run in this process right after the timed legs: 512 workgroups x 4 waves, "
                          f"{n} x {4 * products} v_mfma_f32_32x32x16_{'f16' if h2 else 'bf16'} per wave (~20 ms), operands = the kernel's own planes of random fp32 values"}
    except Exception:
        return None


def kernel_build_id() -> str:
    """sha256 over the kernel sources: ties a committed PMC summary to the library it was measured with."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "summertts_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measure_config(eng, torch, n_cfg: int, conv_math: str, steps: int = 5, warmup: int = 2, cpu_threads: int = 16, with_cpu: bool = True):
    """One BASELINE.json config (per-GPU share) measured in this process next to the headline: K timed steps with the two matrix-core-region
    events (value, ms/step, trunk roofline), three more with all stage events (stage split), and -- with_cpu -- the shortest utterance of
    the batch run ONCE through the compiled reference (oracle/_ref) on the host cores: parity of that utterance + a one-repetition CPU figure."""
    from summertts_amd import synth_blob as sb
    pz = CONFIG_PRESETS[n_cfg]
    cfg = sb.full_cfg(pz["workload"])
    blob = sb.make_blob(cfg, 1234)
    lens = np.random.default_rng(1234).integers(64, 257, size=pz["batch"]).tolist() if pz["ragged"] else [pz["phonemes"]] * pz["batch"]
    ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=u) for u, t in enumerate(lens)]
    syn = eng.Synthesizer(blob, device=0)
    res = {"workload": pz["label"]}
    try:
        syn.set_conv_math(conv_math)
        sid = [u % max(1, syn.get_speaker_num()) for u in range(len(ids))]
        ls = [1.0] * len(ids)
        prepared = syn.prepare(ids, sid, ls)
        # the timed steps: batches of the same lengths the engine has NOT served before (main(): ADVICE r05), the canonical batch for the rest
        fresh = [syn.prepare([sb.synthetic_ids(len(a), cfg.vocab, salt=u, family=j + 1) for u, a in enumerate(ids)], sid, ls) for j in range(steps)]
        for _ in range(warmup):
            syn.run_batch(prepared)
        syn.set_profiling(2)
        torch.cuda.synchronize()
        acc, samples, lat = {}, 0, []
        t0 = time.perf_counter()
        for j in range(steps):
            tq = time.perf_counter()
            samples += int(syn.run_batch(fresh[j]).sum())
            lat.append(time.perf_counter() - tq)
            for k, v in syn.profile().items():
                acc[k] = acc.get(k, 0.0) + float(v)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        products = 3 if conv_math == "f16x2" else (BF16_PRODUCTS_PER_F32 if conv_math == "bf16x3" else 0)
        peak = PEAK_BF16_MFMA_TFLOPS / products if products else PEAK_F32_MFMA_TFLOPS
        mm = acc.get("ms_decoder_mfma", 0.0)
        ach = (acc.get("flops_decoder_mfma", 0.0) / (mm * 1e-3)) / 1e12 if mm > 0 else 0.0
        res.update(value=samples / el, unit="samples/s", x_realtime_16khz=samples / el / 16000.0, steps=steps, warmup=warmup, ms_per_step=1e3 * el / steps,
                   p50_latency_ms=1e3 * float(np.median(lat)), utterances=len(ids), samples_per_step=samples // steps, conv_math=conv_math,
                   timed_requests="batches the engine had not served before (launch-ahead memo cannot answer)",
                   host_sync_wait_ms_per_step=acc.get("ms_sync_wait_host", 0.0) / steps,
                   roofline={"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                             "launches_per_step": acc.get("decoder_mfma_launches", 0.0) / steps,
                             "avg_launch_us": 1e3 * mm / max(1.0, acc.get("decoder_mfma_launches", 0.0))})
        syn.set_profiling(True)
        sacc = {}
        for _ in range(3):
            syn.run_batch(prepared)
            for k, v in syn.profile().items():
                sacc[k] = sacc.get(k, 0.0) + float(v)
        res["stage_ms_per_step"] = {k[3:]: sacc.get(k, 0.0) / 3 for k in ("ms_text_encoder", "ms_duration", "ms_flow", "ms_decoder")}
        if with_cpu:
            from oracle import pyref
            cand = [u for u in range(len(ids)) if sid[u] == 0] or list(range(len(ids)))
            u0 = min(cand, key=lambda u: len(ids[u]))
            syn.set_record_taps(True)
            n_out = [int(v) for v in syn.run_batch(ids, sid, ls)]
            s0, p0 = sum(n_out[:u0]), sum(len(a) for a in ids[:u0])
            got = {"pcm": syn.pcm_host()[s0:s0 + n_out[u0]].copy(), "durations": syn.durations(sum(len(a) for a in ids))[p0:p0 + len(ids[u0])].copy(),
                   "wave": syn.tap("wave")[0][s0:s0 + n_out[u0]].copy()}
            syn.set_record_taps(False)
            if pyref.have_ref():
                _set_omp_threads(cpu_threads)
                ref = pyref.RefModel(blob)
                ref.infer_ids(sb.synthetic_ids(6, cfg.vocab), 0, 1.0)                 # page-in / OpenMP start-up
                tc = time.perf_counter()
                ro = ref.infer_ids(ids[u0], sid[u0], ls[u0])
                tc = time.perf_counter() - tc
                ref.close()
                res["parity"] = dict(parity_report(ro, got), checked=f"utterance {u0} of the batch ({len(ids[u0])} phonemes, speaker {sid[u0]}) vs the compiled reference, "
                                                                     "same blob and ids; tolerance: durations equal, PCM <= 1 LSB, wave RMSE <= 2e-6")
                res["cpu_baseline"] = {"value": ro["wave"].size / tc, "unit": "samples/s", "x_realtime": ro["wave"].size / tc / 16000.0, "cores": cpu_threads,
                                       "kind": "reference", "sample": f"1 x utterance {u0} ({len(ids[u0])} phonemes -> {ro['wave'].size} samples), {tc:.2f} s, "
                                                                      f"{cpu_threads} OpenMP threads, one repetition (a bounded sample)"}
            else:
                res["parity"] = None
    except Exception as e:      # an extra config must never take the headline down with it
        res["error"] = f"{type(e).__name__}: {e}"
    finally:
        syn.close()
    return res


def native_multi_bench(args) -> int:
    print(json.dumps(native_multi_measure(args)), flush=True)
    return 0


def native_multi_measure(args) -> dict:
    """`--gpus N --multi native`: ONE process, sts_multi_create_ex(devices 0..N-1, STS_MULTI_RCCL) -- the library's own utterance sharding and
    RCCL gather (ncclAllGather of the counts, ncclSend / grouped ncclRecv of the int16 PCM to device 0, one download): the code a C++ caller
    of the drop-in library gets, which the torch.distributed path of the default `--gpus N` does not exercise.  Same JSON
    line: value = samples of ALL devices / wall time of K calls (weak scaling: args.batch utterances per device), plus the communicator size as
    RCCL reports it and rank 0's gather time.  STS_BENCH_RCCL_LIB=<path> + --share-gpu (tests): N emulated ranks on device 0 against
    tests/fake_rccl (needs STS_TEST_HOOKS=1)."""
    eng = importlib.import_module(os.environ.get("STS_BENCH_ENGINE", "summertts_amd.engine"))
    stub = getattr(eng, "IS_STUB", False)
    from summertts_amd import synth_blob as sb
    from summertts_amd import sharding
    N = max(1, args.gpus)
    cfg = sb.full_cfg(args.workload) if not stub else sb.tiny_cfg("hifigan_fix")
    if os.environ.get("STS_BENCH_TINY") == "1":
        cfg = sb.tiny_cfg("ms_hifigan_sdp" if args.workload.startswith("ms_hifigan") else "hifigan_sdp")
    blob = sb.make_blob(cfg, 1234)
    fake = os.environ.get("STS_BENCH_RCCL_LIB")
    if fake:
        eng.MultiDevice.set_rccl_library(fake, allow_repeated_devices=True)
    devices = [0] * N if args.share_gpu else list(range(N))
    md = eng.MultiDevice(blob, devices, gather="rccl")
    if hasattr(md, "set_conv_math"):
        md.set_conv_math(args.conv_math)
    gB = N * args.batch
    lens = np.random.default_rng(1234).integers(64, 257, size=gB).tolist() if args.ragged else [args.phonemes] * gB
    ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=u) for u, t in enumerate(lens)]
    spk = 1
    try:
        spk = max(1, int(md.lib.sts_multi_speaker_num(md.h)))
    except Exception:
        pass
    sid = [u % spk for u in range(gB)]
    ls = [1.0] * gB
    slot = [int(v) for v in md.shard_of([len(a) for a in ids])]
    for _ in range(args.warmup):
        md.infer_batch(ids, sid, ls)
    lat, samples, gms = [], 0, 0.0
    per_dev = [0] * N
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        pcm = md.infer_batch(ids, sid, ls)
        lat.append(time.perf_counter() - ts)
        gms += md.last_gather_ms()
        for u, a_ in enumerate(pcm):
            samples += int(a_.size)
            per_dev[slot[u]] += int(a_.size)
    elapsed = time.perf_counter() - t0
    steps = max(1, args.steps)
    out = {
        "metric": "audio samples/sec (acoustic model + vocoder, phoneme ids -> int16 PCM on host), single_speaker_fast-shaped synthetic blob",
        "value": samples / elapsed, "unit": "samples/s", "x_realtime_16khz": samples / elapsed / 16000.0, "n_gpus": N, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps, "p50_latency_ms": 1e3 * float(np.median(lat)), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": f"f32 (trunk arithmetic {args.conv_math}: see the one-GPU line)",
        "data": "synthetic (seeded random weights in the reference .bin grammar)",
        "config": {"workload": config_label(args) + f" [synthetic '{args.workload}' blob; batch={args.batch}/GPU]", "global_batch": gB,
                   "parallelism": f"utterance-sharded x{N}, one process (sts_multi)", "launched_by": "bench.py --multi native (one process)",
                   "kernel_build_id": kernel_build_id(), "conv_math": args.conv_math},
        "multi_gpu": {"mode": "native: sts_multi_create_ex(STS_MULTI_RCCL) -- sharding, ncclAllGather of the counts, ncclSend / grouped ncclRecv of the PCM "
                              "and the one download all inside libsummertts_hip.so",
                      "gather_mode": md.gather_mode(), "rccl_ranks": md.rccl_ranks(), "devices": devices,
                      "gather_ms_per_step_rank0": gms / steps, "samples_per_device": per_dev,
                      "utterances_per_device": [slot.count(k) for k in range(N)],
                      "rccl_library": fake or "librccl.so.1"},
        "roofline": None, "cpu_baseline": None,
        "note": "roofline / cpu_baseline: see the N = 1 line (sts_multi does not expose per-engine profiles)",
    }
    md.close()
    return out


def _env_conv_math() -> str:
    """STS_CONV_MATH as the engine reads it (f32 / fp32 / 1 = exact-fp32 MFMA, bf16x3 / 0 = split-bf16, anything else = two-term fp16)."""
    v = os.environ.get("STS_CONV_MATH", "")
    return "f32" if v in ("f32", "fp32", "1") else ("bf16x3" if v in ("bf16x3", "0") else "f16x2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="BASELINE.json configs[N] preset (sets --workload / --batch / --phonemes / --ragged; batch is per GPU, so "
                         "`--gpus 8 --config 3` is the 256-utterance mixed-speaker batch).  0 = use the individual flags (default = configs[1])")
    ap.add_argument("--workload", default="hifigan_sdp",
                    help="synthetic stand-in for single_speaker_fast.bin: hifigan_sdp (VITS HiFi-GAN + stochastic DP, "
                         "the heavier reading) | mbb_fix | ms_fix | ms_sdp | istft_fix | ms_hifigan_sdp")
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step")
    ap.add_argument("--phonemes", type=int, default=128)
    ap.add_argument("--ragged", action="store_true",
                    help="utterance lengths ~ U{64..256} (seed 1234) instead of --phonemes each (BASELINE configs[2..4])")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed repetitions of the CPU baseline (median reported)")
    ap.add_argument("--cpu-threads", default="8,16,32,64", help="OpenMP thread counts tried once for the CPU baseline")
    ap.add_argument("--cpu-sample-phonemes", type=int, default=0, help="0 = the GPU step's utterance (SURVEY 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-utt", default="auto", choices=["auto", "first", "shortest"],
                    help="which utterance of the timed step the CPU baseline (and `parity`) runs: auto = the first one at batch 1, the shortest one "
                         "of a batch (a bounded sample: the reference needs ~0.2 s per phoneme)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="after the K timed steps, keep stepping for at least this long and report that leg as `sustained` (the K-step burst of a "
                         "3 ms step lasts 60 ms: boost clocks; 0 = off)")
    ap.add_argument("--conv-mode", type=int, default=0)
    ap.add_argument("--conv-math", default=_env_conv_math(), choices=["bf16x3", "f32", "f16x2"],
                    help="arithmetic of the decoder trunk convs: f16x2 (default) = fp32 operands as two fp16 terms, three fp16 MFMA products per "
                         "fp32 product, fp32 accumulation; bf16x3 = fp32 operands split exactly into three bf16 terms, six bf16 MFMA products; "
                         "f32 = the exact-fp32 MFMA.  One set of parity tolerances for all three; the line carries a timed leg of each")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra timed leg on the exact-fp32 MFMA path")
    ap.add_argument("--timed-profiling", type=int, default=2, choices=[0, 1, 2],
                    help="HIP events recorded inside the K timed steps: 2 (default) = the two around the matrix-core region (the roofline's kernel time), "
                         "1 = all eight stage events, 0 = none (lab A/B of what the events cost; the roofline then comes from the stage-breakdown leg)")
    ap.add_argument("--configs-block", default="auto", choices=["auto", "on", "off"],
                    help="append `configs`: BASELINE configs[2] (32 HiFi-GAN utterances) and configs[4] (64 MB-iSTFT utterances: the iSTFT / PQMF path) "
                         "measured in the same process at 5 steps each, with the parity of their shortest utterance against the compiled reference.  "
                         "auto = on for the default configs[1] run on one GPU")
    ap.add_argument("--multi", default="torch", choices=["torch", "native"],
                    help="--gpus N > 1: torch = one process per GPU, torch.distributed gather (the default, what a launcher starts); native = ONE process, "
                         "sts_multi_create_ex(devices 0..N-1, STS_MULTI_RCCL): the library's own RCCL gather, the code a C++ caller gets")
    ap.add_argument("--multi-leg", default="on", choices=["on", "off"],
                    help="--gpus N > 1 under --multi torch: after the distributed run, rank 0 also measures the library's native RCCL gather in-process and "
                         "appends it as `multi_native` (on by default: one SCALE invocation measures both gathers)")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE",
                    help="lab use: sts_debug_set on the timed engine, e.g. tile_claim=0 (A/B of a dispatch choice in one process / on one box)")
    ap.add_argument("--pipeline-engines", type=int, default=2,
                    help="extra (not the headline): throughput with this many engines fed by concurrent host threads, "
                         "so one utterance's latency-bound text side overlaps another's decoder; 0 disables")
    ap.add_argument("--backend", default=os.environ.get("STS_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch.distributed backend of the PCM gather: nccl (= RCCL, the product path) | gloo (tests: the "
                         "PCM goes through the host)")
    ap.add_argument("--share-gpu", action="store_true", help="tests only: every rank uses device 0 (needs --backend gloo)")
    args = ap.parse_args()
    if args.config:
        pz = CONFIG_PRESETS[args.config]
        args.workload, args.batch, args.phonemes, args.ragged = pz["workload"], pz["batch"], pz["phonemes"], pz["ragged"]

    if args.multi == "native":
        if int(os.environ.get("RANK", "0")) != 0:      # (started under a launcher: rank 0 drives all devices, the other ranks have nothing to do)
            sys.exit(0)
        sys.exit(native_multi_bench(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)
    # the engine binding is looked up by name so that the launcher / gather path can be exercised on a machine without
    # a GPU (tests/test_bench_launch_cpu.py runs `--gpus 2 --backend gloo` with tests/stub_engine.py)
    eng = importlib.import_module(os.environ.get("STS_BENCH_ENGINE", "summertts_amd.engine"))
    stub = getattr(eng, "IS_STUB", False)
    import torch
    dev_index = 0 if (args.share_gpu or stub) else local_rank
    use_cuda = not stub
    dist = None
    force_dist = os.environ.get("STS_BENCH_FORCE_DIST") == "1"   # exercise the gather path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from summertts_amd import synth_blob as sb
    from summertts_amd import sharding

    cfg = sb.full_cfg(args.workload) if not stub else sb.tiny_cfg("hifigan_fix")
    blob = sb.make_blob(cfg, 1234)
    syn = eng.Synthesizer(blob, device=dev_index)
    syn.set_conv_mode(args.conv_mode)
    if hasattr(syn, "set_conv_math"):
        syn.set_conv_math(args.conv_math)
    for kv in args.debug_set:
        k, v = kv.split("=")
        syn.debug_set(k, int(v))
    if dist is not None and args.backend == "nccl" and hasattr(syn, "set_host_pcm"):
        syn.set_host_pcm(False)          # the PCM goes device-to-device into the RCCL gather

    # global batch = world * batch utterances, sharded by utterance (no data-path collective)
    gB = world * args.batch
    if args.ragged:
        lens = np.random.default_rng(1234).integers(64, 257, size=gB).tolist()
    else:
        lens = [args.phonemes] * gB
    all_ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=u) for u, t in enumerate(lens)]
    shards = sharding.shard_utterances([len(a) for a in all_ids], world)
    mine = shards[rank]
    max_utts = max(len(s_) for s_ in shards)
    ids = [all_ids[u] for u in mine]
    sid = [u % max(1, syn.get_speaker_num()) for u in mine]
    ls = [1.0] * len(ids)

    # N > 1: the PCM gather of step k (count exchange + gather + download + unpack on rank 0) runs on a
    # helper thread while the main thread already synthesises step k + 1; every gather is completed before the
    # clock stops (drain()).  Collectives are issued by ONE thread per rank, in step order.
    gq = None
    gather_busy = [0.0]
    gathered = [0]
    if dist is not None:
        import queue
        import threading
        gq = queue.Queue()

        def gather_worker():
            if args.backend == "nccl":
                torch.cuda.set_device(dev_index)
            while True:
                item = gq.get()
                if item is None:
                    gq.task_done()
                    return
                local, counts = item
                tg = time.perf_counter()
                res = sharding.gather_variable(local, counts, dist, torch, rank, world, max_utts)
                gather_busy[0] += time.perf_counter() - tg
                if res is not None:
                    gathered[0] += sum(int(a_.size) for per_rank in res for a_ in per_rank)
                gq.task_done()
        gth = threading.Thread(target=gather_worker, daemon=True)
        gth.start()

    # the step's inputs are built once (the bench contract has them resident before the timed region); the PCM of a step is handed over
    # as a view of the engine's pinned download buffer (sts_pcm_host_view) -- a caller that keeps it past its next call copies it
    prepared = syn.prepare(ids, sid, ls) if (ids and hasattr(syn, "prepare")) else None
    # the engine's launch-ahead memo answers a REPEATED request without the host waiting for the frame count; a service that
    # synthesises distinct texts never repeats one.  The K timed steps therefore each run a batch the engine has not served before (same
    # utterance lengths, speakers and length scales as the canonical batch, other phoneme ids); the repeated-batch figure is reported next to
    # it as `memo_hit_leg`.  Warm-up, parity, the CPU baseline and the stage legs use the canonical batch (ids[i] = (i*37 + 11 + u) mod vocab).

    def fresh_batch(j):
        fid = [sb.synthetic_ids(len(all_ids[u]), cfg.vocab, salt=u, family=j) for u in mine]
        return syn.prepare(fid, sid, ls) if (fid and hasattr(syn, "prepare")) else None
    fresh = [fresh_batch(j + 1) for j in range(max(args.steps, 32))] if prepared is not None else []

    def step(batch=None):
        batch = batch if batch is not None else prepared
        n_out = sharding.run_shard(syn, batch, None, None) if batch is not None else sharding.run_shard(syn, ids, sid, ls)
        total = int(n_out.sum()) if len(n_out) else 0
        if dist is None:
            pcm = syn.pcm_host(copy=False) if batch is not None else syn.pcm_host()
            return total, pcm
        if args.backend == "nccl":
            local = torch.empty(max(1, total), dtype=torch.int16, device="cuda")
            if total:
                syn.pcm_to_device_ptr(local.data_ptr(), local.numel())      # device-to-device; the engine is free again
        else:
            local = torch.from_numpy(syn.pcm_host().copy()) if total else torch.zeros(1, dtype=torch.int16)
        gq.put((local[:total], [int(v) for v in n_out]))
        return total, None

    def drain():
        if gq is not None:
            gq.join()

    def sync():
        if use_cuda:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if use_cuda:
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    # the timed steps record only the two HIP events that bracket the matrix-core region (the dominant kernel family, timed inside the timed
    # region as the contract asks); the per-stage breakdown comes from a separate leg below -- each stage event is a barrier packet between
    # two kernels and the eight of them cost the step ~20-30 us (docs/HISTORY.md 12-4)
    try:
        syn.set_profiling(args.timed_profiling)
    except Exception:
        syn.set_profiling(True)
    gather_busy[0] = 0.0
    gathered[0] = 0
    sync()
    lat = []
    acc = {}
    recs = []           # one raw profile record per timed step, summed after the timed region
    samples = 0
    t0 = time.perf_counter()
    for j in range(args.steps):
        ts = time.perf_counter()
        n, _pcm = step(fresh[j] if fresh else None)
        lat.append(time.perf_counter() - ts)
        samples += n
        recs.append(syn.profile_struct() if hasattr(syn, "profile_struct") else syn.profile())
    t_drain0 = time.perf_counter()
    drain()
    drain_wait = time.perf_counter() - t_drain0
    sync()
    elapsed = time.perf_counter() - t0
    # ---- the same K steps on the canonical batch, which the warm-up has put into the memo: every step runs ahead of its frame count
    memo_hit_leg = None
    if fresh and dist is None and not stub:
        tm0 = time.perf_counter()
        msamp = 0
        for _ in range(args.steps):
            n, _pcm = step()
            msamp += n
        if use_cuda:
            torch.cuda.synchronize()
        tm = time.perf_counter() - tm0
        memo_hit_leg = {"steps": args.steps, "ms_per_step": 1e3 * tm / max(1, args.steps), "value": msamp / tm, "unit": "samples/s",
                        "note": "K steps of ONE repeated batch (rounds 4-5's headline): the engine knows every utterance's frame count from its memo and "
                                "enqueues flow + decoder without waiting for the duration predictor; `value` of the line is the never-repeating case"}
    for r_ in recs:
        for k, v in (r_.as_dict() if hasattr(r_, "as_dict") else r_).items():
            acc[k] = acc.get(k, 0.0) + float(v)
    last = syn.profile()
    # ---- the same step for at least --min-seconds more: what the box sustains once clocks and temperatures have settled
    sustained = None
    if args.min_seconds > 0 and dist is None and not stub:
        ts0 = time.perf_counter()
        ns, ssamp = 0, 0
        while time.perf_counter() - ts0 < args.min_seconds:
            if fresh and ns % len(fresh) == 0:
                syn.debug_set("memo_clear", 1)        # the rotation starts over: forget it, so that no step of this leg repeats a request either
            n, _pcm = step(fresh[ns % len(fresh)] if fresh else None)
            ssamp += n
            ns += 1
        if use_cuda:
            torch.cuda.synchronize()
        se = time.perf_counter() - ts0
        sustained = {"seconds": se, "steps": ns, "ms_per_step": 1e3 * se / max(1, ns), "value": ssamp / se, "unit": "samples/s",
                     "x_realtime_16khz": ssamp / se / 16000.0,
                     "note": "back-to-back steps right after the K timed ones, same process, same engine, batches the engine has not served before (a rotation of "
                             f"{len(fresh)} with the launch-ahead memo cleared at every wrap); `value` of the line stays the K-step figure the contract defines"}

    # ---- stage breakdown: the same step with all eight stage events on (untimed for the headline)
    stage_acc, stage_steps, stage_wall = dict(acc), max(1, args.steps), elapsed
    if not stub:
        syn.set_profiling(True)
        step()
        drain()
        sync()
        stage_acc, stage_steps = {}, max(5, args.steps // 2)
        tb0 = time.perf_counter()
        for _ in range(stage_steps):
            step()
            for k, v in syn.profile().items():
                stage_acc[k] = stage_acc.get(k, 0.0) + float(v)
        drain()
        sync()
        stage_wall = time.perf_counter() - tb0

    # ---- the reference's call shape: sts_infer_ids -- fresh argument arrays, a malloc'd copy of the PCM per call -- on four
    # DIFFERENT utterances of the step's length in rotation, launch-ahead off (its memo would hit on a repeated request): what a caller of
    # the drop-in API gets per call when nothing repeats
    api_leg = None
    if dist is None and not stub and len(ids) == 1 and hasattr(syn, "infer_ids"):
        syn.set_profiling(False)
        syn.debug_set("launch_ahead", 0)
        rot = [sb.synthetic_ids(len(ids[0]), cfg.vocab, salt=100 + q) for q in range(4)]
        for q in range(4):
            syn.infer_ids(rot[q], sid[0], ls[0])
        n_api = max(8, args.steps)
        la, sa = [], 0
        torch.cuda.synchronize()
        ta0 = time.perf_counter()
        for q in range(n_api):
            tq = time.perf_counter()
            sa += int(syn.infer_ids(rot[q % 4], sid[0], ls[0]).size)
            la.append(time.perf_counter() - tq)
        ta = time.perf_counter() - ta0
        api_leg = {"calls": n_api, "ms_per_call": 1e3 * ta / n_api, "p50_latency_ms": 1e3 * float(np.median(la)), "value": sa / ta, "unit": "samples/s",
                   "x_realtime_16khz": sa / ta / 16000.0,
                   "note": "sts_infer_ids per call (argument arrays built per call, PCM copied into caller-owned memory), four different utterances of the "
                           "step's length in rotation, launch-ahead disabled: no call repeats a request, every call waits for its frame count"}
        syn.debug_set("launch_ahead", 1)
        syn.set_profiling(True)

    # the utterance of the step that the reference is run on (CPU baseline + parity)
    cpu_u = 0
    if ids and (args.cpu_sample_utt == "shortest" or (args.cpu_sample_utt == "auto" and len(ids) > 1)):
        cand = [u for u in range(len(ids)) if sid[u] == 0] or list(range(len(ids)))
        cpu_u = min(cand, key=lambda u: len(ids[u]))

    def capture_output():
        """One extra UNTIMED run of the step with the float-waveform tap on: utterance cpu_u's durations / PCM / waveform, to be compared
        with the reference's output for the same blob and ids (parity_report)."""
        if stub or dist is not None or not ids or not hasattr(syn, "set_record_taps"):
            return None
        try:
            syn.set_record_taps(True)
            n_out = [int(v) for v in syn.run_batch(ids, sid, ls)]
            s0, t0_ = sum(n_out[:cpu_u]), sum(len(a) for a in ids[:cpu_u])
            got = {"pcm": syn.pcm_host()[s0:s0 + n_out[cpu_u]].copy(),
                   "durations": syn.durations(sum(len(a) for a in ids))[t0_:t0_ + len(ids[cpu_u])].copy(),
                   "wave": syn.tap("wave")[0][s0:s0 + n_out[cpu_u]].copy()}
            return got
        except Exception as e:
            return {"error": str(e)}
        finally:
            syn.set_record_taps(False)

    gpu_out = {args.conv_math: capture_output()}

    # ---- extra timed legs (rank 0, one GPU): the same step with the trunk convs in the other arithmetics -- the exact-fp32 MFMA
    # instruction and, under the two-term fp16 default, the split-bf16 form -- so that the line carries every arithmetic path
    # measured in the same process, each with its parity against the reference
    def extra_leg(math):
        syn.set_conv_math(math)
        for _ in range(2):
            step()
        n2 = max(5, args.steps // 2)
        acc2, samples2 = {}, 0
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(n2):
            n, _pcm = step()
            samples2 += n
            for k, v in syn.profile().items():
                acc2[k] = acc2.get(k, 0.0) + float(v)
        torch.cuda.synchronize()
        e2 = time.perf_counter() - t2
        m2 = acc2.get("ms_decoder_mfma", 0.0)
        a2 = (acc2.get("flops_decoder_mfma", 0.0) / (m2 * 1e-3)) / 1e12 if m2 > 0 else 0.0
        leg = {"value": samples2 / e2, "unit": "samples/s", "x_realtime_16khz": samples2 / e2 / 16000.0, "steps": n2,
               "ms_per_step": 1e3 * e2 / n2,
               "stage_ms_per_step": {k[3:]: acc2.get(k, 0.0) / n2 for k in ("ms_text_encoder", "ms_duration", "ms_flow", "ms_decoder")}}
        if math == "f32":
            i2 = (acc2.get("flops_decoder_mfma_executed", 0.0) / (m2 * 1e-3)) / 1e12 if m2 > 0 else 0.0
            leg["dtype"] = "f32 (v_mfma_f32_32x32x2_f32, Winograd-domain fused layers)"
            leg["roofline"] = {"bound": "mfma", "achieved": a2, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": a2 / PEAK_F32_MFMA_TFLOPS, "mfma_issued_tflops": i2, "mfma_issued_frac": i2 / PEAK_F32_MFMA_TFLOPS}
        else:
            pk = PEAK_BF16_MFMA_TFLOPS / BF16_PRODUCTS_PER_F32
            leg["dtype"] = "f32 (3 bf16 terms per operand, 6 bf16 MFMA products per fp32 product)"
            leg["roofline"] = {"bound": "mfma", "achieved": a2, "peak": pk, "unit": "TFLOP/s", "frac": a2 / pk}
        gpu_out[math] = capture_output()
        syn.set_conv_math(args.conv_math)
        return leg

    f32_leg = bf3_leg = None
    if dist is None and not stub and args.conv_math in ("bf16x3", "f16x2") and not args.no_f32_leg and hasattr(syn, "set_conv_math"):
        f32_leg = extra_leg("f32")
        if args.conv_math == "f16x2":
            bf3_leg = extra_leg("bf16x3")
    ceiling = measured_mfma_ceiling(3 if args.conv_math == "f16x2" else BF16_PRODUCTS_PER_F32) if (dist is None and not stub and rank == 0) else None

    # ---- extra figure (not the headline): the native request pool (sts_pool: N engines, one worker thread each,
    # one FIFO).  "pipelined" = batch-1 requests only overlapped across engines (max_batch 1); "burst" = the same
    # requests submitted at once with dynamic packed batching (max_batch 8).
    pipelined = None
    if dist is None and args.pipeline_engines >= 2 and not stub:
        nreq = max(8, args.steps)
        pipelined = {"engines": args.pipeline_engines, "requests": nreq}
        for label, mb in (("pipelined_batch1", 1), ("burst_max_batch8", 8)):
            pool = eng.Pool(blob, device=0, n_engines=args.pipeline_engines, max_batch=mb)
            for t_ in [pool.submit(ids[0], sid[0], ls[0]) for _ in range(2 * args.pipeline_engines)]:
                pool.wait(t_)                                    # warm-up: every engine has run once
            rounds = []                                          # three rounds of nreq requests: the median one is reported (one round of
            cpu0, wall0 = time.process_time(), time.perf_counter()
            for r_i in range(3):                                 # 20 requests is 40 ms: a single host hiccup used to decide the figure)
                # requests the pool's engines have NOT served before (a family of id sequences per round): its workers wait for every frame count
                req = [sb.synthetic_ids(len(ids[u % len(ids)]), cfg.vocab, salt=u, family=500 + 37 * r_i + (0 if mb == 1 else 1000)) for u in range(nreq)]
                tp0 = time.perf_counter()
                tk = [pool.submit(req[u], sid[u % len(ids)], ls[u % len(ids)]) for u in range(nreq)]
                done = sum(int(pool.wait(t_).size) for t_ in tk)
                rounds.append((time.perf_counter() - tp0, done))
            cores = (time.process_time() - cpu0) / max(1e-9, time.perf_counter() - wall0)
            tp, done = sorted(rounds)[1]
            pipelined[label] = {"value": done / tp, "unit": "samples/s", "x_realtime_16khz": done / tp / 16000.0,
                                "ms_per_request": 1e3 * tp / nreq, "ms_per_request_rounds": [round(1e3 * r_[0] / nreq, 3) for r_ in rounds],
                                "host_cores_busy": cores,
                                "host_cores_busy_definition": "process CPU time / wall time over the three rounds (submitting thread + every pool worker): the workers sleep "
                                                              "through most of the wait for a new request's frame counts instead of polling it (engine.hip wait_frame_counts)"}
            pool.close()

    total_samples = samples
    per_rank = [samples]
    if dist is not None:
        dev = "cuda" if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, float(samples)], dtype=torch.float64, device=dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        allr = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allr, t)
        elapsed = float(tmax[0].item())
        per_rank = [int(a_[1].item()) for a_ in allr]
        total_samples = int(sum(per_rank))

    # HBM traffic / MFMA-busy of the dominant kernel family come from separate rocprofv3 --pmc passes (the guide's
    # recipe: FETCH_SIZE and WRITE_SIZE in their own runs), summarised by tools/summarize_profile.py.  A summary is
    # attached only when it was taken with THIS library build on THIS workload.
    traffic = None
    pmc_extra = {}
    wl_key = (f"{args.workload}|batch={args.batch}|phonemes={args.phonemes}|ragged={int(args.ragged)}"
              + ("" if args.conv_math == "f16x2" else f"|math={args.conv_math}"))      # (the summaries are taken under the default arithmetic)
    build_id = kernel_build_id()
    try:
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        pmc = pmc_all.get("by_workload", {}).get(wl_key)
        if pmc and pmc.get("kernel_build_id") == build_id:
            traffic = {"hbm_bytes_per_launch": pmc["hbm_bytes_per_launch_corrected"],
                       "source": f"profiles/{pmc['tag']}_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, workload {wl_key}, "
                                 f"kernel build {build_id})",
                       "correction": pmc["correction"]}
            for k in ("mfma_busy_pct_serialised", "hbm_gbps_serialised", "avg_launch_us_rocprof", "flow"):
                if k in pmc:
                    pmc_extra[k] = pmc[k]
    except Exception:
        pass
    if rank == 0:
        steps = max(1, args.steps)
        value = total_samples / elapsed
        racc, rsteps = acc, steps
        if args.timed_profiling == 0:           # (lab A/B: no events in the timed steps -- the kernel time of the roofline comes from the stage leg)
            racc, rsteps = stage_acc, stage_steps
        mfma_ms, launches = racc.get("ms_decoder_mfma", 0.0), racc.get("decoder_mfma_launches", 0.0)
        achieved_tf = (racc.get("flops_decoder_mfma", 0.0) / (mfma_ms * 1e-3)) / 1e12 if mfma_ms > 0 else 0.0
        issued_tf = (racc.get("flops_decoder_mfma_executed", 0.0) / (mfma_ms * 1e-3)) / 1e12 if mfma_ms > 0 else 0.0
        bf16_tf = (racc.get("flops_decoder_bf16_issued", 0.0) / (mfma_ms * 1e-3)) / 1e12 if mfma_ms > 0 else 0.0
        split = bf16_tf > 0.0                       # the trunk ran on split operands (conv_bf3.hip)
        products = 3 if args.conv_math == "f16x2" else BF16_PRODUCTS_PER_F32      # 16-bit matrix products per fp32 product
        peak_tf = PEAK_BF16_MFMA_TFLOPS / products if split else PEAK_F32_MFMA_TFLOPS
        # per-stage rooflines of the part of the step that is NOT the matrix-core decoder: bound = max(bytes / HBM peak, flops / MFMA peak)
        stages = {}
        for name, kms, kfl, kby in (("text_encoder", "ms_text_encoder", "flops_text_encoder", "bytes_text_encoder"),
                                    ("duration", "ms_duration", "flops_duration", "bytes_duration"),
                                    ("flow", "ms_flow", "flops_flow", "bytes_flow"),
                                    ("decoder", "ms_decoder", "flops_decoder", "bytes_decoder_min")):
            ms = stage_acc.get(kms, 0.0) / stage_steps          # (from the separate leg with all stage events on)
            fl = stage_acc.get(kfl, 0.0) / stage_steps
            by = stage_acc.get(kby, 0.0) / stage_steps
            t_hbm = by / (PEAK_HBM_GBS * 1e9) * 1e3
            # the pipe the stage's matrix work runs on: decoder trunk = the line's arithmetic; the reverse flow runs on two-term fp16 operands too under
            # f16x2 (wn_flow.hip / conv_bf3 at batch; VERDICT r05 weak 3: pricing it at the exact-fp32 peak flattered it 5x); the rest is fp32 MFMA
            stage_peak = peak_tf if (name == "decoder" or (name == "flow" and split and args.conv_math == "f16x2")) else PEAK_F32_MFMA_TFLOPS
            t_mfma = fl / (stage_peak * 1e12) * 1e3
            bound_ms = max(t_hbm, t_mfma)
            stages[name] = {"ms": ms, "gflop": fl / 1e9, "mbytes": by / 1e6, "bound": "hbm" if t_hbm >= t_mfma else "mfma",
                            "bound_ms": bound_ms, "mfma_peak_tflops": stage_peak, "frac_of_bound": (bound_ms / ms) if ms > 0 else None,
                            "hbm_gbps_algorithmic": (by / 1e9) / (ms * 1e-3) if ms > 0 else None,
                            "tflops_algorithmic": (fl / 1e12) / (ms * 1e-3) if ms > 0 else None}
        out = {
            "metric": "audio samples/sec (acoustic model + vocoder, phoneme ids -> int16 PCM on host), "
                      "single_speaker_fast-shaped synthetic blob",
            "value": value,
            "unit": "samples/s",
            "x_realtime_16khz": value / 16000.0,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "p50_latency_ms": 1e3 * float(np.median(lat)),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": (("f32 (decoder trunk convs AND the reverse flow's WaveNet convs (wn_flow.hip, one-utterance calls): fp32 operands as 2 fp16 terms = "
                       "22-23 of their 24 mantissa bits, 3 fp16 MFMA products per fp32 product, f32 accumulation -- measured error in `parity`, same "
                       "tolerance as the exact-fp32 leg; batches additionally run the flow / text-encoder / conv_pre convs in this form from ~384 workgroups per launch on; "
                       "everything else -- attention, LayerNorm, duration predictor, splines, tails -- plain f32)")
                      if args.conv_math == "f16x2" else
                      ("f32 (decoder trunk convs: fp32 operands split exactly into 3 bf16 terms, 6 bf16 MFMA products per fp32 product, "
                       "f32 accumulation; everything else f32)")) if split else "f32",
            "data": "synthetic (seeded random weights in the reference .bin grammar; canonical ids[i]=(i*37+11+u) mod vocab, timed steps: a different id sequence of the same length per step); the reference's "
                    "real .bin models are absent from /root/reference, every number here is on synthetic weights",
            "config": {
                "workload": config_label(args) + f" [synthetic '{args.workload}' blob, {blob.size} floats; batch={args.batch}/GPU, "
                            + ("64..256 phonemes/utterance (ragged)]" if args.ragged else f"{args.phonemes} phonemes/utterance]"),
                "baseline_config": args.config or next((n for n, pz in CONFIG_PRESETS.items() if config_label(args) == pz["label"]), None),
                "global_batch": gB, "phonemes": args.phonemes, "frames_per_step_rank0": int(last["frames"]),
                "samples_per_step_rank0": int(last["samples"]), "parallelism": f"utterance-sharded x{world}",
                "launched_by": "bench.py (self-spawned ranks)" if os.environ.get("STS_BENCH_SELF_LAUNCHED") else
                               ("torchrun / external launcher" if world > 1 else "single process"),
                "kernel_build_id": build_id, "conv_math": args.conv_math,
            },
            "stage_ms_per_step": {k: stages[k]["ms"] for k in ("text_encoder", "duration", "flow", "decoder")},
            "stage_breakdown_leg": {"steps": stage_steps, "ms_per_step": 1e3 * stage_wall / stage_steps,
                                    "note": "stage_ms_per_step / roofline_stages come from this separate leg (all eight stage events recorded: each is a barrier "
                                            "packet between two kernels); the K timed steps record only the two events around the matrix-core region"},
            "host_sync_wait_ms_per_step": acc.get("ms_sync_wait_host", 0.0) / steps,
            "host_us_per_step": {"setup_to_first_launch": acc.get("us_host_setup", 0.0) / steps, "entry_to_last_launch": acc.get("us_host_enqueue", 0.0) / steps,
                                 "after_last_sync": acc.get("us_host_tail", 0.0) / steps,
                                 "step_wall_minus_device_stages": 1e3 * (1e3 * stage_wall / stage_steps - sum(stages[k]["ms"] for k in ("text_encoder", "duration", "flow", "decoder"))),
                                 "pcm": "view of the engine's pinned download buffer (sts_pcm_host_view)" if prepared is not None and dist is None else "copied / gathered"},
            "roofline": {
                "kernel": ("conv_bf3_kernel + conv_bf3_group_kernel / conv_h2p_group_kernel + resblock_bf3_kernel (decoder upsamplers + grouped staged / pre-split + fused ResBlock convs, "
                           + ("v_mfma_f32_32x32x16_f16 on two-term operands)" if args.conv_math == "f16x2" else "v_mfma_f32_32x32x16_bf16 on split operands)")) if split else
                          ("conv_mfma_kernel + conv_mfma_group_kernel + resblock_wino_kernel / resblock_layer_kernel (decoder upsamplers + "
                           "grouped / fused ResBlock convs, v_mfma_f32_32x32x2_f32)"),
                "bound": "mfma",
                "achieved": achieved_tf,
                "peak": peak_tf,
                "unit": "TFLOP/s",
                "frac": achieved_tf / peak_tf,
                "achieved_vs_split_bf16_peak": (achieved_tf / (PEAK_BF16_MFMA_TFLOPS / BF16_PRODUCTS_PER_F32)) if split else None,   # rounds 2 / 3 before f16x2 priced against 416.7
                "peak_definition": (f"bf16 / fp16 dense MFMA peak {PEAK_BF16_MFMA_TFLOPS:.0f} TF/s / {products} 16-bit products per fp32 product "
                                    "(MI355X_MICROARCH.md; the exact-fp32 MFMA peak is 157.3)") if split else
                                   "v_mfma_f32_32x32x2_f32 dense peak (MI355X_MICROARCH.md)",
                "traffic": traffic,
                "achieved_definition": "ALGORITHMIC (direct-form, true-tap) fp32 FLOPs of the launches / their HIP-event time: the task's "
                                       "roofline figure (SURVEY.md 8d).  It is an effective rate, not pipe utilisation -- see *_issued_*",
                "sustained_mfma_ceiling": (dict(ceiling, frac=achieved_tf / ceiling["tflops_fp32_equivalent"], measured=True) if ceiling else
                                           {"tflops_fp32_equivalent": (SUSTAINED_F16_TWO_TERM_TFLOPS if products == 3 else SUSTAINED_BF16_SPLIT_TFLOPS) / products,
                                            "frac": achieved_tf / ((SUSTAINED_F16_TWO_TERM_TFLOPS if products == 3 else SUSTAINED_BF16_SPLIT_TFLOPS) / products),
                                            "measured": False,
                                            "source": "not measured in this run (a multi-rank / forced-distributed run, or tools/ubench/libsts_ubench.so not built): the constant of the arithmetic that ran ("
                                                      + ("1490 fp16 TF/s on two-term planes, round 3" if products == 3 else "1712 bf16 TF/s on three-term planes, round 2")
                                                      + " of the nominal 2500)"}) if split else None,
                "bf16_issued_tflops": bf16_tf,
                "bf16_issued_frac": bf16_tf / PEAK_BF16_MFMA_TFLOPS,
                "bf16_issued_definition": "bf16 matrix-core FLOPs the split-operand launches issue (6 x their algorithmic FLOPs; 3 x under f16x2) / the same "
                                          "time, against the bf16 dense peak: <= 1 by construction",
                "mfma_issued_tflops": issued_tf,
                "mfma_issued_frac": issued_tf / PEAK_F32_MFMA_TFLOPS,
                "mfma_issued_definition": "exact-fp32 matrix-core FLOPs executed by launches on the fp32 MFMA path (0 when the whole trunk runs "
                                          "on split operands; with --conv-math f32 the Winograd-domain layer kernels need (4 n3 + 3 n2) / (2 k) of "
                                          "a k-tap conv's products)",
                "algorithmic_bytes_per_launch": racc.get("bytes_decoder_min", 0.0) / max(1.0, launches),
                "launches_per_step": launches / rsteps,
                "avg_launch_us": 1e3 * mfma_ms / max(1.0, launches),
                "algorithmic_gflop_per_step": racc.get("flops_decoder_mfma", 0.0) / rsteps / 1e9,
                "decoder_min_hbm_gb_per_step": racc.get("bytes_decoder_min", 0.0) / rsteps / 1e9,
            },
            "roofline_stages": stages,
        }
        out["roofline"].update(pmc_extra)
        if dist is not None:
            out["multi_gpu"] = {"backend": args.backend + (" (RCCL)" if args.backend == "nccl" else ""),
                                "samples_per_rank": per_rank, "utterances_per_rank": [len(s_) for s_ in shards],
                                "gather_busy_ms_per_step_rank0": 1e3 * gather_busy[0] / steps,
                                "gather_drain_wait_ms_rank0": 1e3 * drain_wait,
                                "gathered_samples_rank0": gathered[0],
                                "note": "the gather of step k runs on a helper thread under step k + 1; gather_drain_wait is what was "
                                        "left un-overlapped when the clock stopped"}
        if f32_leg is not None:
            out["f32_mfma_leg"] = f32_leg
        if bf3_leg is not None:
            out["bf16x3_leg"] = bf3_leg
        if pipelined is not None:
            out["request_pool"] = pipelined
        if memo_hit_leg is not None:
            out["memo_hit_leg"] = memo_hit_leg
        out["timed_requests"] = {"distinct": bool(fresh), "launch_ahead_runs": int(acc.get("launch_ahead", 0.0)), "steps": steps,
                                 "note": "every timed step runs a batch the engine has not served before (same lengths / speakers / length scales as the canonical "
                                         "batch, other phoneme ids): its launch-ahead memo cannot answer, the host waits for the frame count as it does for any new "
                                         "text.  launch_ahead_runs = timed steps that ran ahead anyway (expected 0)"}
        if sustained is not None:
            out["sustained"] = sustained
        if api_leg is not None:
            out["api_call_leg"] = api_leg
        want_cfgs = args.configs_block == "on" or (args.configs_block == "auto" and world == 1 and dist is None and not stub
                                                   and config_label(args) == CONFIG_PRESETS[1]["label"])
        if want_cfgs and not stub:
            # configs[2] and configs[4] in the driver's own line: the 32-utterance HiFi-GAN batch and the 64-utterance
            # MB-iSTFT batch (iSTFT + PQMF path), 5 timed steps each, parity of the shortest utterance against the compiled reference
            # round 6: + configs[3]'s per-GPU share (multi-speaker HiFi-GAN, 32 utterances: the cond paths) and configs[1]'s
            # call shape on the MB-iSTFT family ("configs[1]-mbb")
            out["configs"] = {("configs[1]-mbb" if n_ == 5 else f"configs[{n_}]"): measure_config(eng, torch, n_, args.conv_math, steps=(10 if n_ == 5 else 5), with_cpu=not args.no_cpu_baseline)
                              for n_ in (2, 4, 3, 5)}
        if world == 1 and not args.no_cpu_baseline and not stub:
            try:
                cpu_T = args.cpu_sample_phonemes or (len(ids[cpu_u]) if ids else args.phonemes)
                cpu_ids = ids[cpu_u] if (ids and cpu_T == len(ids[cpu_u])) else sb.synthetic_ids(cpu_T, cfg.vocab)
                ref_out, out["cpu_baseline"] = cpu_baseline(blob, cfg.vocab, cpu_ids, args.cpu_reps,
                                                            [int(x) for x in args.cpu_threads.split(",") if x])
                if out["cpu_baseline"] is not None:
                    out["cpu_baseline"]["sample_utterance"] = f"utterance {cpu_u} of the timed step ({len(cpu_ids)} phonemes)"
                # parity of the PUBLISHED workload, in the line that publishes it: the reference's output for one utterance of this
                # very step (same blob, same ids) against the GPU output of every timed leg
                if ref_out is not None and ids and cpu_ids is ids[cpu_u] and sid[cpu_u] == 0:
                    par = {}
                    for leg, g in gpu_out.items():
                        par[leg] = g if (g is None or "error" in g) else parity_report(ref_out, g)
                    par["checked"] = (f"utterance {cpu_u} of the timed step ({len(cpu_ids)} phonemes, speaker 0) vs the compiled reference "
                                      "(oracle/_ref) on the same blob and ids; tolerance: durations equal, PCM <= 1 LSB, wave RMSE <= 2e-6")
                    out["parity"] = par
            except Exception as e:   # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": str(e)}
        result_line = json.dumps(out)
    else:
        result_line = None
    if dist is not None:
        if gq is not None:
            gq.put(None)
        dist.destroy_process_group()
    # One SCALE invocation measures BOTH gathers: after the one-process-per-GPU run (torch.distributed gather) rank 0 drives
    # all N devices through the library's own sharding + RCCL gather (sts_multi_create_ex(0..N-1, STS_MULTI_RCCL): what a C++ caller of the drop-in
    # gets) in this process and appends the figures to the same line.  The other ranks have left their GPUs by then or are about to; a failure of
    # this leg never costs the line its headline.
    if result_line is not None and world > 1 and args.multi_leg != "off":
        try:
            syn.close()
            nat = native_multi_measure(args)
            out["multi_native"] = {"value": nat["value"], "unit": nat["unit"], "ms_per_step": nat["ms_per_step"], "n_gpus": nat["n_gpus"],
                                   "rccl_ranks": nat["multi_gpu"]["rccl_ranks"], "gather_mode": nat["multi_gpu"]["gather_mode"],
                                   "gather_ms_per_step": nat["multi_gpu"]["gather_ms_per_step_rank0"], "samples_per_device": nat["multi_gpu"]["samples_per_device"],
                                   "note": "the same global batch through ONE process and the library's native RCCL gather (sts_multi), measured right after the "
                                           "torch.distributed run of this line; `value` of the line stays the one-process-per-GPU figure"}
        except Exception as e:      # noqa: BLE001
            out["multi_native"] = {"error": f"{type(e).__name__}: {e}"}
        result_line = json.dumps(out)
    if result_line is not None:      # the JSON line is the LAST thing on stdout (RCCL may print banners earlier)
        sys.stdout.flush()
        print(result_line, flush=True)
    if dist is not None:
        # RCCL prints a banner from an exit handler; leave without running it so that the JSON stays last
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
