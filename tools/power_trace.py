#!/usr/bin/env python3
"""Power + shader-clock telemetry under the decoder trunk and under the bare-MFMA micro-benchmark.

Samples the GPU's hwmon sensors (sysfs: power1_average / power1_input in uW, freq1_input = sclk in Hz; falls back to
`rocm-smi --showpower --showclocks --json`) every 50 ms from a helper thread while the main thread keeps the GPU busy for
~5 s per phase:
  idle | the synthesis step at batch 1 | at batch 8 (ragged) | tools/ubench/libsts_ubench.so on constant operands, random
  bf16 operands, and the hi/mid/lo planes of random fp32 data.
Prints per phase: samples, mean / max power (W), mean / min sclk (MHz), and the phase's own throughput figure.
  python tools/power_trace.py [seconds_per_phase]"""
import ctypes
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def all_sensors():
    out = []
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        p = next((os.path.join(hw, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, n))), None)
        f = os.path.join(hw, "freq1_input") if os.path.exists(os.path.join(hw, "freq1_input")) else None
        if p:
            out.append((p, f))
    return out


def find_sensors():
    c = all_sensors()
    return c[0] if c else (None, None)


def pick_loaded_sensor(load_fn):
    """A box may expose the hwmon nodes of several cards while HIP sees one GPU: the sensor that belongs to OUR device is the one whose
    power rises while `load_fn` keeps the device busy (round 4: a box whose card0 was somebody else's idle GPU reported 250 W / 95 MHz
    through a whole trace)."""
    cands = all_sensors()
    if len(cands) <= 1:
        return cands[0] if cands else (None, None)

    def rd(pf):
        try:
            return int(open(pf).read()) / 1e6
        except Exception:
            return float("nan")
    idle = [rd(pf) for pf, _ in cands]
    peak = list(idle)
    stop = [False]

    def poll():
        while not stop[0]:
            for i, (pf, _) in enumerate(cands):
                v = rd(pf)
                if v == v and (peak[i] != peak[i] or v > peak[i]):
                    peak[i] = v
            time.sleep(0.02)
    th = threading.Thread(target=poll, daemon=True)
    th.start()
    load_fn(1.5)
    stop[0] = True
    th.join()
    rise = [(peak[i] - idle[i]) if (peak[i] == peak[i] and idle[i] == idle[i]) else -1.0 for i in range(len(cands))]
    best = int(np.argmax(rise))
    print("sensor candidates (idle W -> peak W under load): " + ", ".join(f"{os.path.basename(os.path.dirname(os.path.dirname(pf)))}/{os.path.basename(os.path.dirname(pf))} {idle[i]:.0f}->{peak[i]:.0f}" for i, (pf, _) in enumerate(cands)) + f"; using #{best}", flush=True)
    return cands[best]


class Sampler:
    def __init__(self, period=0.05):
        self.period, self.rows, self.stop = period, [], False
        self.pfile, self.ffile = find_sensors()
        self.mode = "sysfs" if self.pfile else "rocm-smi"

    def read(self):
        if self.pfile:
            try:
                pw = int(open(self.pfile).read()) / 1e6
                fq = int(open(self.ffile).read()) / 1e6 if self.ffile else float("nan")
                return pw, fq
            except Exception:
                return float("nan"), float("nan")
        try:
            j = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = next(iter(j.values()))
            pw = next((float(v) for k, v in c.items() if "Power" in k and "W" in k), float("nan"))
            fq = next((float(str(v).strip("()Mhz ")) for k, v in c.items() if k.lower().startswith("sclk")), float("nan"))
            return pw, fq
        except Exception:
            return float("nan"), float("nan")

    def run(self):
        while not self.stop:
            t = time.perf_counter()
            self.rows.append((t,) + self.read())
            time.sleep(max(0.0, self.period - (time.perf_counter() - t)))

    def phase(self, name, fn, seconds):
        self.rows, self.stop = [], False
        th = threading.Thread(target=self.run, daemon=True)
        th.start()
        t0 = time.perf_counter()
        note = fn(seconds)
        el = time.perf_counter() - t0
        self.stop = True
        th.join()
        a = np.array([r[1:] for r in self.rows if r[0] - t0 > 0.5] or [(float("nan"), float("nan"))])   # skip the ramp
        print(f"{name:58s} {el:5.1f} s  {len(a):3d} samples @ {1e3 * self.period:.0f} ms ({self.mode})  power mean {np.nanmean(a[:, 0]):6.1f} W max {np.nanmax(a[:, 0]):6.1f} W"
              f"   sclk mean {np.nanmean(a[:, 1]):6.0f} MHz min {np.nanmin(a[:, 1]):6.0f} MHz   {note}", flush=True)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    from summertts_amd import engine, synth_blob as sb
    cfg = sb.full_cfg("hifigan_sdp")
    blob = sb.make_blob(cfg, 1234)
    syn = engine.Synthesizer(blob)
    s = Sampler()
    cap = ""
    if s.pfile:
        for n in ("power1_cap", "power1_cap_default", "power1_cap_max"):
            f = os.path.join(os.path.dirname(s.pfile), n)
            if os.path.exists(f):
                try:
                    cap += f" {n}={int(open(f).read()) / 1e6:.0f} W"
                except Exception:
                    pass
    print(f"sensors: power={s.pfile} sclk={s.ffile}{cap}", flush=True)

    def idle(sec):
        time.sleep(sec)
        return ""

    def calib_load(sec):                 # keeps OUR device busy: which hwmon node reacts?
        ids0 = [sb.synthetic_ids(128, cfg.vocab, salt=0)] * 8
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < sec:
            syn.run_batch(ids0)
    if s.mode == "sysfs":
        s.pfile, s.ffile = pick_loaded_sensor(calib_load)
        print(f"sensors in use: power={s.pfile} sclk={s.ffile}", flush=True)

    def synth(batch):
        lens = [128] if batch == 1 else np.random.default_rng(1234).integers(64, 257, size=batch).tolist()
        ids = [sb.synthetic_ids(int(t), cfg.vocab, salt=u) for u, t in enumerate(lens)]

        def f(sec):
            syn.set_profiling(True)
            t0, n, dec, fl = time.perf_counter(), 0, 0.0, 0.0
            while time.perf_counter() - t0 < sec:
                n += int(syn.run_batch(ids).sum())
                p = syn.profile()
                dec += p["ms_decoder_mfma"]; fl += p["flops_decoder_mfma"]
            return f"{n / (time.perf_counter() - t0) / 16000:.0f}x real-time, trunk {fl / dec / 1e9:.0f} TF/s fp32-equivalent"
        return f

    lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libsts_ubench.so"))
    lib.sts_ubench_mfma_bf16.restype = ctypes.c_double
    lib.sts_ubench_mfma_bf16.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def ub(mode):
        def f(sec):
            t0, r = time.perf_counter(), []
            while time.perf_counter() - t0 < sec:
                r.append(lib.sts_ubench_mfma_bf16(mode, 512, 200000))       # ~0.2 s per call
            return f"{np.mean(r):.0f} bf16 TF/s = {np.mean(r) / 6:.0f} fp32-equivalent"
        return f

    def ubh(mode):
        lib.sts_ubench_mfma_f16.restype = ctypes.c_double
        lib.sts_ubench_mfma_f16.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]

        def f(sec):
            t0, r = time.perf_counter(), []
            while time.perf_counter() - t0 < sec:
                r.append(lib.sts_ubench_mfma_f16(mode, 512, 400000))        # ~0.2 s per call
            return f"{np.mean(r):.0f} fp16 TF/s = {np.mean(r) / 3:.0f} fp32-equivalent"
        return f

    if os.environ.get("POWER_TRACE_SHORT"):      # the phases that changed with the two-term fp16 default (round 3)
        s.phase("idle", idle, 1.0)
        s.phase("synthesis step, batch 1 (128 phonemes), default arithmetic", synth(1), secs)
        s.phase("synthesis step, batch 8 (64..256 phonemes), default arithmetic", synth(8), secs)
        s.phase("synthesis step, batch 32 (config 2), default arithmetic", synth(32), secs)
        s.phase("bare fp16 MFMA loop, two-term planes of random fp32", ubh(2), secs)
        return
    s.phase("idle", idle, 2.0)
    s.phase("synthesis step, batch 1 (128 phonemes)", synth(1), secs)
    s.phase("synthesis step, batch 8 (64..256 phonemes)", synth(8), secs)
    s.phase("synthesis step, batch 32 (config 2)", synth(32), secs)
    s.phase("bare MFMA loop, constant operands", ub(0), secs)
    s.phase("bare MFMA loop, random bf16 operands", ub(1), secs)
    s.phase("bare MFMA loop, hi/mid/lo planes of random fp32", ub(2), secs)
    s.phase("bare fp16 MFMA loop, two-term planes of random fp32", ubh(2), secs)
    s.phase("idle", idle, 2.0)


if __name__ == "__main__":
    main()
