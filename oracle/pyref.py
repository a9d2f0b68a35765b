"""ctypes bindings for the parity checkers.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``summertts_amd``) never does.

Two checkers share one Python interface (``RefModel`` / ``PortModel``):

* ``oracle/_ref/libsummertts_ref.so`` -- the real SummerTTS Eigen path, compiled in place from
  /root/reference by ``oracle/Makefile`` and driven by ``oracle/ref_harness.cpp``.
* ``oracle/libvits_oracle.so`` -- the plain-C restatement ``oracle/vits_oracle.c``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "libsummertts_ref.so")
PORT_SO = os.path.join(HERE, "libvits_oracle.so")
REFERENCE_ROOT = "/root/reference"


# The checkers parallelise with OpenMP.  On a many-core host (the GPU box: 256 CPUs) libgomp's default -- one thread per CPU, each spinning
# between parallel regions -- turns a tiny model's thousands of short loops into minutes of wall time and ~16 cores of pure spinning next
# to the HIP runtime's own threads (round 6: tests/fake_rccl/three_ranks.py took 8 m 45 s of a 12-minute GPU suite, 137 CPU-minutes).  The
# team is capped at 16 threads when the first model is created (omp_set_num_threads: works whether or not libgomp is already loaded), unless
# the user chose OMP_NUM_THREADS; bench.py's cpu_baseline leg sets the team size it measures with explicitly afterwards.
_omp_capped = False


def _cap_omp_threads() -> None:
    global _omp_capped
    if _omp_capped or os.environ.get("OMP_NUM_THREADS"):
        return
    _omp_capped = True
    for name in ("libgomp.so.1", "libgomp.so"):
        try:
            C.CDLL(name, mode=C.RTLD_GLOBAL).omp_set_num_threads(int(min(16, os.cpu_count() or 1)))
            return
        except OSError:
            continue


def build(port: bool = True, ref: bool = True, quiet: bool = True) -> None:
    """Compile the checkers (ref only when /root/reference exists: it does not on the GPU box)."""
    targets = []
    if port:
        targets.append("port")
    if ref and os.path.isdir(REFERENCE_ROOT):
        targets.append("ref")
    if targets:
        subprocess.run(["make", "-C", HERE, "-j8"] + targets, check=True,
                       stdout=subprocess.DEVNULL if quiet else None)


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def have_port() -> bool:
    return os.path.exists(PORT_SO)


class _Model:
    """Common wrapper: both libraries export the same C symbols with a prefix."""

    TAPS = ("x_enc", "m", "logs", "logw", "z_p", "z")

    def __init__(self, so: str, prefix: str, blob: np.ndarray):
        self.lib = C.CDLL(so)
        _cap_omp_threads()
        self.p = prefix
        f = self._f
        f("create").restype = C.c_void_p
        f("create").argtypes = [C.c_void_p, C.c_int64]
        f("consumed").restype = C.c_int64
        f("consumed").argtypes = [C.c_void_p]
        f("speaker_num").restype = C.c_int
        f("speaker_num").argtypes = [C.c_void_p]
        f("destroy").argtypes = [C.c_void_p]
        f("infer_ids").restype = C.c_int64
        f("infer_ids").argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int]
        f("wave").restype = C.POINTER(C.c_float)
        f("wave").argtypes = [C.c_void_p]
        f("durations").restype = C.POINTER(C.c_int32)
        f("durations").argtypes = [C.c_void_p]
        f("pcm").argtypes = [C.c_void_p, C.c_void_p]
        f("tap").restype = C.c_int
        f("tap").argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_float)),
                             C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        f("times").argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self.blob = np.ascontiguousarray(blob, dtype=np.float32)
        self.h = f("create")(self.blob.ctypes.data, self.blob.size)
        if not self.h:
            raise RuntimeError("oracle: model construction failed")

    def _f(self, name):
        return getattr(self.lib, self.p + name)

    @property
    def consumed(self) -> int:
        return int(self._f("consumed")(self.h))

    @property
    def speaker_num(self) -> int:
        return int(self._f("speaker_num")(self.h))

    def infer_ids(self, ids: Sequence[int], sid: int = 0, length_scale: float = 1.0,
                  forced_dur: Optional[Sequence[int]] = None, taps: bool = False) -> Dict[str, np.ndarray]:
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        fd = None
        if forced_dur is not None:
            fd = np.ascontiguousarray(forced_dur, dtype=np.int32)
            assert fd.size == ids.size
        n = int(self._f("infer_ids")(self.h, ids.ctypes.data, ids.size, sid, length_scale,
                                       fd.ctypes.data if fd is not None else None, 1 if taps else 0))
        wave = np.ctypeslib.as_array(self._f("wave")(self.h), shape=(n,)).copy()
        dur = np.ctypeslib.as_array(self._f("durations")(self.h), shape=(ids.size,)).copy()
        pcm = np.empty(n, np.int16)
        self._f("pcm")(self.h, pcm.ctypes.data)
        tm = (C.c_double * 5)()
        self._f("times")(self.h, tm)
        out = {"wave": wave, "durations": dur, "pcm": pcm,
               "times": dict(zip(("te", "dp", "flow", "dec", "total"), list(tm)))}
        if taps:
            for name in self.TAPS:
                ptr = C.POINTER(C.c_float)()
                r, c = C.c_int32(), C.c_int32()
                if self._f("tap")(self.h, name.encode(), C.byref(ptr), C.byref(r), C.byref(c)) == 0:
                    # column-major [rows=time, cols=channels]  ->  numpy [channels, time]
                    out[name] = np.ctypeslib.as_array(ptr, shape=(c.value, r.value)).copy()
        return out

    def close(self):
        if self.h:
            self._f("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RefModel(_Model):
    """The real reference (Eigen CPU) acoustic path."""

    def __init__(self, blob: np.ndarray):
        if not have_ref():
            raise FileNotFoundError(REF_SO + " (run `make -C oracle ref` where /root/reference exists)")
        super().__init__(REF_SO, "ref_", blob)


class PortModel(_Model):
    """The plain-C restatement."""

    def __init__(self, blob: np.ndarray):
        if not have_port():
            build(port=True, ref=False)
        super().__init__(PORT_SO, "port_", blob)
