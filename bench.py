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

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def cpu_baseline(blob, cfg, vocab, sample_T):
    """Times the reference (oracle/_ref, the real Eigen path) or, failing that, the C restatement on
    a bounded sample of the same workload on this box's host cores."""
    from oracle import pyref
    from summertts_amd import synth_blob as sb
    kind = None
    model = None
    if pyref.have_ref():
        model, kind = pyref.RefModel(blob), "reference"
    else:
        try:
            model, kind = pyref.PortModel(blob), "port"
        except Exception:
            return None
    ids_w = sb.synthetic_ids(6, vocab)
    model.infer_ids(ids_w, 0, 1.0)            # warm-up: pays the OpenMP start-up
    ids = sb.synthetic_ids(sample_T, vocab)
    t0 = time.perf_counter()
    out = model.infer_ids(ids, 0, 1.0)
    dt = time.perf_counter() - t0
    n = int(out["wave"].size)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    return {"value": n / dt, "unit": "samples/s", "cores": cores, "kind": kind,
            "sample": f"1 utterance, {sample_T} phonemes -> {n} samples in {dt:.2f} s "
                      f"(same blob; {'reference Eigen path, -O3 -fopenmp' if kind == 'reference' else 'C restatement, OpenMP'})",
            "x_realtime": n / dt / 16000.0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="hifigan_sdp",
                    help="synthetic stand-in for single_speaker_fast.bin: hifigan_sdp (VITS HiFi-GAN + stochastic DP, "
                         "the heavier reading) | mbb_fix | ms_fix | istft_fix | ms_hifigan_sdp")
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step")
    ap.add_argument("--phonemes", type=int, default=128)
    ap.add_argument("--ragged", action="store_true",
                    help="utterance lengths ~ U{64..256} (seed 1234) instead of --phonemes each (BASELINE configs[2..4])")
    ap.add_argument("--cpu-sample-phonemes", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-mode", type=int, default=0)
    ap.add_argument("--pipeline-engines", type=int, default=2,
                    help="extra (not the headline): throughput with this many engines fed by concurrent host threads, "
                         "so one utterance's latency-bound text side overlaps another's decoder; 0 disables")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    force_dist = os.environ.get("STS_BENCH_FORCE_DIST") == "1"   # exercise the RCCL gather path on one GPU
    if world > 1 or force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from summertts_amd import engine as eng
    from summertts_amd import synth_blob as sb
    from summertts_amd import sharding

    cfg = sb.full_cfg(args.workload)
    blob = sb.make_blob(cfg, 1234)
    syn = eng.Synthesizer(blob, device=local_rank if world > 1 else 0)
    syn.set_conv_mode(args.conv_mode)

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

    # N > 1: the PCM gather of step k (count exchange + RCCL gather + download + unpack on rank 0) runs on a
    # helper thread while the main thread already synthesises step k + 1; every gather is completed before the
    # clock stops (drain()).  Collectives are issued by ONE thread per rank, in step order.
    gq = None
    if dist is not None:
        import queue
        import threading
        gq = queue.Queue()
        gathered = [0]

        def gather_worker():
            torch.cuda.set_device(local_rank)
            while True:
                item = gq.get()
                if item is None:
                    gq.task_done()
                    return
                local, counts = item
                res = sharding.gather_variable(local, counts, dist, torch, rank, world, max_utts)
                if res is not None:
                    gathered[0] += sum(int(a_.size) for per_rank in res for a_ in per_rank)
                gq.task_done()
        gth = threading.Thread(target=gather_worker, daemon=True)
        gth.start()

    def step():
        n_out = syn.run_batch(ids, sid, ls)
        if dist is None:
            pcm = syn.pcm_host()
            return int(n_out.sum()), pcm
        total = int(n_out.sum())
        local = torch.empty(max(1, total), dtype=torch.int16, device="cuda")
        syn.pcm_to_device_ptr(local.data_ptr(), local.numel())      # device-to-device; the engine is free again
        gq.put((local[:total], [int(v) for v in n_out]))
        return total, None

    def drain():
        if gq is not None:
            gq.join()

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    syn.set_profiling(True)
    sync()
    lat = []
    mfma_ms = mfma_flops = dec_ms = dec_flops = dec_bytes = 0.0
    launches = 0
    stage_ms = np.zeros(4)
    samples = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        n, _pcm = step()
        lat.append(time.perf_counter() - ts)
        samples += n
        p = syn.profile()
        mfma_ms += p["ms_decoder_mfma"]; mfma_flops += p["flops_decoder_mfma"]; launches += p["decoder_mfma_launches"]
        dec_ms += p["ms_decoder"]; dec_flops += p["flops_decoder"]; dec_bytes += p["bytes_decoder_min"]
        stage_ms += np.array([p["ms_text_encoder"], p["ms_duration"], p["ms_flow"], p["ms_decoder"]])
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    last = syn.profile()

    # ---- extra figure (not the headline): the native request pool (sts_pool: N engines, one worker thread each,
    # one FIFO).  "pipelined" = batch-1 requests only overlapped across engines (max_batch 1); "burst" = the same
    # requests submitted at once with dynamic packed batching (max_batch 8).
    pipelined = None
    if dist is None and args.pipeline_engines >= 2:
        nreq = max(8, args.steps)
        pipelined = {"engines": args.pipeline_engines, "requests": nreq}
        for label, mb in (("pipelined_batch1", 1), ("burst_max_batch8", 8)):
            pool = eng.Pool(blob, device=0, n_engines=args.pipeline_engines, max_batch=mb)
            for t_ in [pool.submit(ids[0], sid[0], ls[0]) for _ in range(2 * args.pipeline_engines)]:
                pool.wait(t_)                                    # warm-up: every engine has run once
            tp0 = time.perf_counter()
            tk = [pool.submit(ids[u % len(ids)], sid[u % len(ids)], ls[u % len(ids)]) for u in range(nreq)]
            done = sum(int(pool.wait(t_).size) for t_ in tk)
            tp = time.perf_counter() - tp0
            pipelined[label] = {"value": done / tp, "unit": "samples/s", "x_realtime_16khz": done / tp / 16000.0,
                                "ms_per_request": 1e3 * tp / nreq}
            pool.close()

    total_samples = samples
    if dist is not None:
        t = torch.tensor([elapsed, float(samples)], dtype=torch.float64, device="cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        total_samples = int(tsum[1].item())

    # HBM traffic of the dominant kernel family comes from a separate rocprofv3 --pmc pass (the guide's
    # recipe: FETCH_SIZE and WRITE_SIZE in their own runs), summarised by tools/summarize_profile.py
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "latest_pmc.json")))
        if args.workload == "hifigan_sdp" and args.batch == 1 and args.phonemes == 128 and not args.ragged:
            traffic = {"hbm_bytes_per_launch": pmc["hbm_bytes_per_launch_corrected"], "source": "profiles/latest_pmc.json (" + pmc["tag"] + ")",
                       "correction": pmc["correction"]}
    except Exception:
        pass
    if rank == 0:
        value = total_samples / elapsed
        achieved_tf = (mfma_flops / (mfma_ms * 1e-3)) / 1e12 if mfma_ms > 0 else 0.0
        out = {
            "metric": "audio samples/sec (acoustic model + vocoder, phoneme ids -> int16 PCM on host), "
                      "single_speaker_fast-shaped synthetic blob",
            "value": value,
            "unit": "samples/s",
            "x_realtime_16khz": value / 16000.0,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "p50_latency_ms": 1e3 * float(np.median(lat)),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (seeded random weights in the reference .bin grammar; ids[i]=(i*37+11) mod vocab)",
            "config": {
                "workload": f"configs[1]: single_speaker_fast (synthetic '{args.workload}' blob, "
                            f"{blob.size} floats), batch={args.batch}/GPU, "
                            + ("64..256 phonemes/utterance (ragged)" if args.ragged else f"{args.phonemes} phonemes/utterance"),
                "global_batch": gB, "phonemes": args.phonemes, "frames_per_step_rank0": int(last["frames"]),
                "samples_per_step_rank0": int(last["samples"]), "parallelism": f"utterance-sharded x{world}",
            },
            "stage_ms_per_step": {k: float(v / args.steps) for k, v in zip(("text_encoder", "duration", "flow", "decoder"), stage_ms)},
            "roofline": {
                "kernel": "conv_mfma_kernel + conv_mfma_group_kernel + resblock_wino_kernel / resblock_layer_kernel (decoder upsamplers + grouped / fused ResBlock convs, v_mfma_f32_32x32x2_f32)",
                "bound": "mfma",
                "achieved": achieved_tf,
                "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved_tf / PEAK_F32_MFMA_TFLOPS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": dec_bytes / max(1, launches),
                "launches_per_step": launches / max(1, args.steps),
                "avg_launch_us": 1e3 * mfma_ms / max(1, launches),
                "flops_note": "achieved = ALGORITHMIC (direct-form, true-tap) FLOPs / time; the Winograd-domain layer kernels "
                              "(resblock_wino_kernel) execute about 0.73x as many MFMA FLOPs for the same fp32 result",
                "algorithmic_gflop_per_step": mfma_flops / max(1, args.steps) / 1e9,
                "decoder_min_hbm_gb_per_step": dec_bytes / max(1, args.steps) / 1e9,
            },
        }
        if pipelined is not None:
            out["request_pool"] = pipelined
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(blob, cfg, cfg.vocab, args.cpu_sample_phonemes)
            except Exception as e:   # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"error": str(e)}
        result_line = json.dumps(out)
    else:
        result_line = None
    if dist is not None:
        dist.destroy_process_group()
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
