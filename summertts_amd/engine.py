"""ctypes binding of ``libsummertts_hip.so`` (C ABI: ``include/summertts_hip.h``).

This is host-side plumbing only: every call goes straight to the HIP engine; there is NO CPU
fallback -- if the shared library or a GPU is missing the constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SUMMERTTS_HIP_LIB") or os.path.join(_HERE, "lib", "libsummertts_hip.so")   # override: kernel timing experiments
CSRC = os.path.join(_HERE, "csrc")


def build_library(verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into ``summertts_amd/lib/libsummertts_hip.so`` (in-tree)."""
    subprocess.run(["make", "-C", CSRC, "-j8"], check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


class ModelInfo(C.Structure):
    _fields_ = [("is_multi_speaker", C.c_int32), ("lang_type", C.c_int32), ("dur_pred_type", C.c_int32),
                ("dec_type", C.c_int32), ("vocab", C.c_int32), ("hidden", C.c_int32), ("inter_channels", C.c_int32),
                ("speaker_num", C.c_int32), ("gin_channels", C.c_int32), ("samples_per_frame", C.c_int32),
                ("sample_rate", C.c_int32), ("blob_floats_consumed", C.c_int64)]


class Profile(C.Structure):
    _fields_ = [("ms_text_encoder", C.c_float), ("ms_duration", C.c_float), ("ms_flow", C.c_float),
                ("ms_decoder", C.c_float), ("ms_total_device", C.c_float), ("ms_decoder_mfma", C.c_float),
                ("decoder_mfma_launches", C.c_int32), ("flops_text_encoder", C.c_double),
                ("flops_duration", C.c_double), ("flops_flow", C.c_double), ("flops_decoder", C.c_double),
                ("flops_decoder_mfma", C.c_double), ("bytes_decoder_min", C.c_double), ("frames", C.c_int64),
                ("samples", C.c_int64), ("phonemes", C.c_int64), ("flops_decoder_mfma_executed", C.c_double),
                ("bytes_text_encoder", C.c_double), ("bytes_duration", C.c_double), ("bytes_flow", C.c_double),
                ("ms_sync_wait_host", C.c_float), ("flops_decoder_bf16_issued", C.c_double),
                ("conv_math_fallbacks", C.c_int64), ("conv_math_pinned", C.c_int32), ("launch_ahead", C.c_int32), ("launch_ahead_misses", C.c_int64),
                ("us_host_setup", C.c_float), ("us_host_enqueue", C.c_float), ("us_host_tail", C.c_float)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class PreparedBatch:
    """run_batch's argument arrays, built once (Synthesizer.prepare)."""

    def __init__(self, ids, sid=None, length_scale=None):
        B = self.B = len(ids)
        self.arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in ids]
        self.ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in self.arrs])
        self.n = np.asarray([a.size for a in self.arrs], dtype=np.int32)
        self.sid = np.zeros(B, np.int32) if sid is None else np.ascontiguousarray(sid, dtype=np.int32)
        self.ls = np.ones(B, np.float32) if length_scale is None else np.ascontiguousarray(length_scale, dtype=np.float32)
        self.n_out = np.zeros(B, np.int32)
        self.total = C.c_int64()
        self.n_p, self.sid_p, self.ls_p, self.n_out_p = self.n.ctypes.data, self.sid.ctypes.data, self.ls.ctypes.data, self.n_out.ctypes.data
        self.total_ref = C.byref(self.total)

    def __len__(self):
        return self.B


_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.sts_last_error.restype = C.c_char_p
    lib.sts_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
    lib.sts_destroy.argtypes = [C.c_void_p]
    lib.sts_speaker_num.argtypes = [C.c_void_p]
    lib.sts_get_info.argtypes = [C.c_void_p, C.POINTER(ModelInfo)]
    lib.sts_infer_ids.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                  C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.c_int32)]
    lib.sts_run_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(C.c_int64)]
    lib.sts_copy_pcm_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sts_copy_pcm_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sts_pcm_host_view.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.c_int64)]
    lib.sts_set_forced_durations.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sts_set_record_taps.argtypes = [C.c_void_p, C.c_int]
    lib.sts_set_conv_mode.argtypes = [C.c_void_p, C.c_int]
    lib.sts_set_conv_math.argtypes = [C.c_void_p, C.c_int]
    lib.sts_debug_set.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.sts_set_profiling.argtypes = [C.c_void_p, C.c_int]
    lib.sts_set_host_pcm.argtypes = [C.c_void_p, C.c_int]
    lib.sts_get_profile.argtypes = [C.c_void_p, C.POINTER(Profile)]
    lib.sts_get_profile_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sts_get_tap.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int64)]
    lib.sts_get_durations.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.sts_free.argtypes = [C.c_void_p]
    lib.sts_debug_conv1d.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                     C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int32)]
    lib.sts_debug_conv1d_bench.argtypes = lib.sts_debug_conv1d.argtypes + [C.c_int32, C.POINTER(C.c_float)]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "sts_create", "sts_destroy", "sts_speaker_num", "sts_get_info", "sts_infer_ids", "sts_infer_ids_batch",
    "sts_run_batch", "sts_copy_pcm_device", "sts_copy_pcm_host", "sts_pcm_host_view", "sts_set_forced_durations",
    "sts_set_record_taps", "sts_get_tap", "sts_get_durations", "sts_set_conv_mode", "sts_set_conv_math", "sts_set_profiling",
    "sts_get_profile", "sts_set_host_pcm", "sts_debug_conv1d", "sts_debug_conv1d_bench", "sts_debug_conv_h2p", "sts_debug_conv_h2w", "sts_free", "sts_last_error",
    "sts_infer_ids_stream", "sts_stream_halo_frames", "sts_debug_wino_pack", "sts_debug_set", "sts_multi_create_ex", "sts_multi_gather_mode", "sts_multi_gather_layout",
    "sts_pool_create", "sts_pool_destroy", "sts_pool_submit", "sts_pool_wait", "sts_pool_stats", "sts_pool_last_error",
    "sts_multi_create", "sts_multi_destroy", "sts_multi_device_count", "sts_multi_speaker_num", "sts_multi_infer_ids_batch",
    "sts_multi_shard_of", "sts_multi_last_error", "sts_multi_set_rccl_library", "sts_multi_rccl_ranks", "sts_multi_last_gather_ms", "sts_multi_set_conv_math", "sts_get_profile_ex", "sts_abi_version", "sts_build_flags",
]


def lab_build() -> bool:
    """True when the loaded library was built with -DSTS_EXPERIMENTS (`make -C summertts_amd/csrc exp`)."""
    return bool(load_library().sts_build_flags() & 1)


class StsError(RuntimeError):
    pass


def _check(lib, rc: int):
    if rc != 0:
        raise StsError(f"sts error {rc}: {lib.sts_last_error().decode(errors='replace')}")


class Synthesizer:
    """Python mirror of the reference's ``SynthesizerTrn`` (include/SynthesizerTrn.h) at the phoneme-id
    boundary: construct from a model blob, ``infer_ids`` -> int16 PCM.  Adds the batched entry."""

    def __init__(self, blob: np.ndarray, device: int = 0):
        self.lib = load_library()
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self.h = C.c_void_p()
        _check(self.lib, self.lib.sts_create(blob.ctypes.data, blob.nbytes, device, C.byref(self.h)))
        self.info = ModelInfo()
        _check(self.lib, self.lib.sts_get_info(self.h, C.byref(self.info)))
        self.set_host_pcm(True)     # run_batch + pcm_host is the common pairing; a device-side gather switches it off

    def set_host_pcm(self, on: bool):
        """PCM download as part of the run (one stream sync per call); off = the PCM only stays on the device."""
        _check(self.lib, self.lib.sts_set_host_pcm(self.h, 1 if on else 0))

    # -- reference surface -------------------------------------------------------------------
    def get_speaker_num(self) -> int:
        return int(self.lib.sts_speaker_num(self.h))

    def infer_ids(self, ids: Sequence[int], sid: int = 0, length_scale: float = 1.0) -> np.ndarray:
        return self.infer_batch([ids], [sid], [length_scale])[0]

    # -- streaming (sts_infer_ids_stream) -----------------------------------------------------
    def infer_ids_stream(self, ids: Sequence[int], chunk_frames: int, sid: int = 0, length_scale: float = 1.0,
                         on_chunk=None):
        """Decodes chunk by chunk; ``on_chunk(pcm: np.int16[], sample_offset, t_seconds)`` is called per chunk
        (return True to stop).  Returns (list of chunks, list of arrival times since the call started)."""
        import time
        a = np.ascontiguousarray(ids, dtype=np.int32)
        chunks, times = [], []
        t0 = time.perf_counter()

        def _cb(user, pcm, n, off):
            arr = np.ctypeslib.as_array(pcm, shape=(n,)).copy()
            t = time.perf_counter() - t0
            chunks.append(arr); times.append(t)
            return 1 if (on_chunk and on_chunk(arr, off, t)) else 0
        CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int16), C.c_int32, C.c_int32)
        cb = CB(_cb)
        total = C.c_int32()
        self.lib.sts_infer_ids_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, CB,
                                                  C.c_void_p, C.POINTER(C.c_int32)]
        _check(self.lib, self.lib.sts_infer_ids_stream(self.h, a.ctypes.data, a.size, sid, length_scale, chunk_frames, cb, None,
                                                       C.byref(total)))
        return chunks, times

    def stream_halo_frames(self) -> int:
        return int(self.lib.sts_stream_halo_frames(self.h))

    # -- batched -----------------------------------------------------------------------------
    def prepare(self, ids: Sequence[Sequence[int]], sid: Optional[Sequence[int]] = None,
                length_scale: Optional[Sequence[float]] = None) -> "PreparedBatch":
        """The argument arrays of run_batch built once, for a caller that submits the same batch object repeatedly (a benchmark
        loop, a retry): run_batch(prepared) then costs one foreign call."""
        return PreparedBatch(ids, sid, length_scale)

    def run_batch(self, ids, sid: Optional[Sequence[int]] = None, length_scale: Optional[Sequence[float]] = None) -> np.ndarray:
        """Runs the batch (ids: a sequence of id sequences, or a PreparedBatch) and leaves the PCM on the device (and, with
        set_host_pcm, in the engine's pinned host buffer); returns per-utterance sample counts (a fresh array per call).  Views handed out by
        pcm_host(copy=False) alias the engine's pinned buffer: the next run on this engine rewrites them in place."""
        p = ids if isinstance(ids, PreparedBatch) else PreparedBatch(ids, sid, length_scale)
        _check(self.lib, self.lib.sts_run_batch(self.h, p.B, p.ptrs, p.n_p, p.sid_p, p.ls_p, p.n_out_p, p.total_ref))
        self._total = int(p.total.value)
        self._n_out = p.n_out.copy()       # (the prepared object's own array is overwritten by its next run: ADVICE r04)
        return self._n_out

    def pcm_host(self, copy: bool = True) -> np.ndarray:
        """PCM of the last run on the host.  copy=False: a read-only view of the engine's pinned download buffer (valid until the
        next run on this engine; needs set_host_pcm(True), the default of this class)."""
        if not copy:
            ptr, cnt = C.POINTER(C.c_int16)(), C.c_int64()
            _check(self.lib, self.lib.sts_pcm_host_view(self.h, C.byref(ptr), C.byref(cnt)))
            if cnt.value == 0:
                return np.empty(0, np.int16)
            v = np.ctypeslib.as_array(ptr, shape=(cnt.value,))
            v.flags.writeable = False
            return v
        out = np.empty(self._total, np.int16)
        _check(self.lib, self.lib.sts_copy_pcm_host(self.h, out.ctypes.data, out.size))
        return out

    def profile_struct(self) -> "Profile":
        """The raw profile record of the last run (no dict built: for a timed loop that converts afterwards)."""
        p = Profile()
        _check(self.lib, self.lib.sts_get_profile_ex(self.h, C.byref(p), C.sizeof(p)))
        return p

    def pcm_to_device_ptr(self, ptr: int, capacity: int) -> None:
        _check(self.lib, self.lib.sts_copy_pcm_device(self.h, C.c_void_p(ptr), capacity))

    def infer_batch(self, ids, sid=None, length_scale=None) -> List[np.ndarray]:
        n_out = self.run_batch(ids, sid, length_scale)
        flat = self.pcm_host()
        offs = np.concatenate([[0], np.cumsum(n_out)])
        return [flat[offs[b]:offs[b + 1]].copy() for b in range(len(n_out))]

    # -- parity / diagnostics ---------------------------------------------------------------
    def set_forced_durations(self, dur: Optional[Sequence[int]]):
        if dur is None:
            _check(self.lib, self.lib.sts_set_forced_durations(self.h, None, 0))
            return
        d = np.ascontiguousarray(dur, dtype=np.int32)
        _check(self.lib, self.lib.sts_set_forced_durations(self.h, d.ctypes.data, d.size))

    def set_record_taps(self, on: bool):
        _check(self.lib, self.lib.sts_set_record_taps(self.h, 1 if on else 0))

    def set_conv_math(self, mode):
        """Arithmetic of the decoder trunk convs: 0 / 'bf16x3' = fp32 operands as three bf16 terms on the bf16 matrix cores,
        1 / 'f32' = the exact-fp32 MFMA instruction, 3 / 'f16x2' = two fp16 terms, three products (default; a call whose
        activations leave fp16's range is repeated as 'bf16x3'; Profile.conv_math_fallbacks counts them)."""
        m = {"bf16x3": 0, "f32": 1, "bf16x3_all": 2, "f16x2": 3}.get(mode, mode)
        _check(self.lib, self.lib.sts_set_conv_math(self.h, int(m)))

    def set_conv_mode(self, mode: int):
        _check(self.lib, self.lib.sts_set_conv_mode(self.h, mode))

    def debug_set(self, key: str, value: int):
        """Test hooks (include/summertts_hip.h sts_debug_set): 'attn_block_min_wgs' | 'flow_fused' | 'launch_ahead' | ..."""
        _check(self.lib, self.lib.sts_debug_set(self.h, {"attn_block_min_wgs": 1, "flow_fused": 5, "launch_ahead": 6, "attn_reg": 7, "dds_tail": 8, "pcm_direct": 9, "memo_clear": 10, "h2p": 11, "h2p_tile": 12, "chain_streams": 13, "tail_fused": 14, "ups_rowph": 15}[key], int(value)))

    def set_profiling(self, on):
        """False / True: no / all eight stage events per run; 2: only the two events around the decoder's matrix-core region (the
        per-stage times of Profile then read 0; every event is a barrier packet between two kernels, so the timed headline step uses 2)."""
        _check(self.lib, self.lib.sts_set_profiling(self.h, 2 if on == 2 else (1 if on else 0)))

    def profile(self) -> dict:
        p = Profile()
        _check(self.lib, self.lib.sts_get_profile_ex(self.h, C.byref(p), C.sizeof(p)))
        return p.as_dict()

    def tap(self, name: str) -> np.ndarray:
        ptr = C.POINTER(C.c_float)()
        ch, ln = C.c_int32(), C.c_int64()
        _check(self.lib, self.lib.sts_get_tap(self.h, name.encode(), C.byref(ptr), C.byref(ch), C.byref(ln)))
        a = np.ctypeslib.as_array(ptr, shape=(ch.value, ln.value)).copy()
        self.lib.sts_free(ptr)
        return a

    def durations(self, total_phonemes: int) -> np.ndarray:
        d = np.zeros(total_phonemes, np.int32)
        _check(self.lib, self.lib.sts_get_durations(self.h, d.ctypes.data, d.size))
        return d

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.sts_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def debug_conv1d(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], pad: int, dil: int = 1,
                 stride_transposed: int = 0, depthwise: bool = False, in_slope: float = 0.0, in_act: int = 0,
                 mode: int = 0, device: int = 0, iters: int = 0):
    """One conv through the engine's kernels.  x: [Cin, L]; w: [Cout, k, Cin] (reference layout).
    With iters > 0 returns (y, mean milliseconds per launch)."""
    lib = load_library()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    cout, k = w.shape[0], w.shape[1]
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    y = C.POINTER(C.c_float)()
    lout = C.c_int32()
    ms = C.c_float(0.0)
    _check(lib, lib.sts_debug_conv1d_bench(device, x.ctypes.data, x.shape[0], x.shape[1], w.ctypes.data,
                                           None if b is None else b.ctypes.data, cout, k, pad, dil, stride_transposed,
                                           1 if depthwise else 0, in_slope, in_act, mode, C.byref(y), C.byref(lout),
                                           iters, C.byref(ms)))
    out = np.ctypeslib.as_array(y, shape=(cout, lout.value)).copy()
    lib.sts_free(y)
    return (out, float(ms.value)) if iters > 0 else out


def debug_conv_h2p(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], dil: int = 1, res: Optional[np.ndarray] = None,
                   in_slope: float = 0.1, out_slope: float = 0.1, tile: int = -1, members: int = 1, device: int = 0, iters: int = 0):
    """One 'same'-padded conv through the pre-split path (conv_h2p.hip).  x: [Cin, L]; w: [Cout, k, Cin].  Returns the three output
    forms decoded to fp32 [Cout, L]: (y, y16, yp) -- yp = lrelu(out, out_slope) to the two-term precision -- and ms per launch."""
    lib = load_library()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    cout, k = w.shape[0], w.shape[1]
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if res is None else np.ascontiguousarray(res, np.float32)
    L = x.shape[1]
    y = np.zeros((cout, L), np.float32); y16 = np.zeros((cout, L), np.float32); yp = np.zeros((cout, L), np.float32)
    ms = C.c_float(0.0)
    lib.sts_debug_conv_h2p.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.POINTER(C.c_float)]
    _check(lib, lib.sts_debug_conv_h2p(device, x.ctypes.data, x.shape[0], L, w.ctypes.data, None if b is None else b.ctypes.data, cout, k, dil,
                                       None if r is None else r.ctypes.data, in_slope, out_slope, tile, members, y.ctypes.data,
                                       y16.ctypes.data, yp.ctypes.data, iters, C.byref(ms)))
    return y, y16, yp, float(ms.value)     # (iters < 0: |iters| timed launches WITHOUT the fp32 [C][L] output: a layer's second conv as the engine runs it)


def debug_conv_h2w(x: np.ndarray, w: np.ndarray, bias: Optional[np.ndarray], dil: int = 1, res: Optional[np.ndarray] = None,
                   in_slope: float = 0.1, out_slope: float = 1.0, members: int = 1, device: int = 0, iters: int = 0):
    """One 'same'-padded conv through the Winograd-domain lab kernel (conv_h2w.hip).  x: [C, L]; w: [C, k, C].  Returns (y, y16, ms)."""
    lib = load_library()
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    c, k = w.shape[0], w.shape[1]
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    r = None if res is None else np.ascontiguousarray(res, np.float32)
    L = x.shape[1]
    y = np.zeros((c, L), np.float32); y16 = np.zeros((c, L), np.float32)
    ms = C.c_float(0.0)
    lib.sts_debug_conv_h2w.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_float)]
    _check(lib, lib.sts_debug_conv_h2w(device, x.ctypes.data, c, L, w.ctypes.data, None if b is None else b.ctypes.data, k, dil,
                                       None if r is None else r.ctypes.data, in_slope, out_slope, members, y.ctypes.data, y16.ctypes.data, iters, C.byref(ms)))
    return y, y16, float(ms.value)


def debug_wino_pack(w: np.ndarray) -> np.ndarray:
    """Winograd-domain transform of conv weights w[Cout][k][Cin] -> U[seg][4][Cin_pad][Cout_pad] (host only)."""
    lib = load_library()
    w = np.ascontiguousarray(w, dtype=np.float32)
    co, k, ci = w.shape
    cip, cop = (ci + 15) // 16 * 16, (co + 31) // 32 * 32
    nseg = {0: k // 3, 2: (k - 2) // 3 + 1, 1: (k - 4) // 3 + 2}[k % 3]
    out = np.zeros((nseg, 4, cip, cop), np.float32)
    rc = lib.sts_debug_wino_pack(w.ctypes.data_as(C.c_void_p), co, k, ci, out.ctypes.data_as(C.c_void_p), C.c_int64(out.size))
    if rc != nseg:
        raise StsError(f"sts_debug_wino_pack: {rc}")
    return out


class Pool:
    """``sts_pool``: N engines on one GPU behind one request queue (dynamic packed batching)."""

    def __init__(self, blob: np.ndarray, device: int = 0, n_engines: int = 2, max_batch: int = 8):
        self.lib = load_library()
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self.h = C.c_void_p()
        self.lib.sts_pool_last_error.restype = C.c_char_p
        self.lib.sts_pool_submit.restype = C.c_int64
        self.lib.sts_pool_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float]
        self.lib.sts_pool_wait.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.POINTER(C.c_int16)), C.POINTER(C.c_int32)]
        self.lib.sts_pool_create.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        self.lib.sts_pool_destroy.argtypes = [C.c_void_p]
        self.lib.sts_pool_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        rc = self.lib.sts_pool_create(blob.ctypes.data, blob.nbytes, device, n_engines, max_batch, C.byref(self.h))
        if rc != 0:
            raise StsError(f"sts_pool_create: {rc}: {self.lib.sts_pool_last_error().decode()}")

    def submit(self, ids: Sequence[int], sid: int = 0, length_scale: float = 1.0) -> int:
        a = np.ascontiguousarray(ids, dtype=np.int32)
        t = int(self.lib.sts_pool_submit(self.h, a.ctypes.data, a.size, sid, length_scale))
        if t <= 0:
            raise StsError(f"sts_pool_submit: {t}: {self.lib.sts_pool_last_error().decode()}")
        return t

    def wait(self, ticket: int) -> np.ndarray:
        p = C.POINTER(C.c_int16)()
        n = C.c_int32()
        rc = self.lib.sts_pool_wait(self.h, ticket, C.byref(p), C.byref(n))
        if rc != 0:
            raise StsError(f"sts_pool_wait: {rc}: {self.lib.sts_pool_last_error().decode()}")
        out = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int16)
        self.lib.sts_free(p)
        return out

    def stats(self):
        b, r = C.c_int64(), C.c_int64()
        self.lib.sts_pool_stats(self.h, C.byref(b), C.byref(r))
        return int(b.value), int(r.value)

    def close(self):
        if self.h:
            self.lib.sts_pool_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def multi_gather_layout(counts: Sequence[int]):
    """Host arithmetic of the RCCL gather buffer (sts_multi_gather_layout): -> (offsets per rank, extent), in samples."""
    lib = load_library()
    lib.sts_multi_gather_layout.restype = C.c_int64
    lib.sts_multi_gather_layout.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    c = np.ascontiguousarray(counts, dtype=np.int64)
    off = np.zeros(c.size, np.int64)
    total = int(lib.sts_multi_gather_layout(c.ctypes.data, c.size, off.ctypes.data))
    return off, total


class MultiDevice:
    """``sts_multi``: one process, one engine per listed HIP device; batches are sharded by utterance."""

    MODES = {"auto": 0, "rccl": 1, "download": 2}

    def __init__(self, blob: np.ndarray, devices: Sequence[int], gather: str = "auto"):
        """gather: 'auto' / 'download' (every device downloads its own shard) | 'rccl' (opt-in: RCCL gather to devices[0];
        distinct devices, one is allowed)."""
        self.lib = load_library()
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        self.h = C.c_void_p()
        self.lib.sts_multi_last_error.restype = C.c_char_p
        self.lib.sts_multi_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
        self.lib.sts_multi_destroy.argtypes = [C.c_void_p]
        self.lib.sts_multi_device_count.argtypes = [C.c_void_p]
        self.lib.sts_multi_speaker_num.argtypes = [C.c_void_p]
        self.lib.sts_multi_infer_ids_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                       C.c_void_p, C.c_void_p]
        self.lib.sts_multi_shard_of.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        self.lib.sts_multi_create_ex.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
        self.lib.sts_multi_gather_mode.argtypes = [C.c_void_p]
        rc = self.lib.sts_multi_create_ex(blob.ctypes.data, blob.nbytes, dev.ctypes.data, dev.size, self.MODES[gather], C.byref(self.h))
        if rc != 0:
            raise StsError(f"sts_multi_create_ex: {rc}: {self.lib.sts_multi_last_error().decode()}")

    def gather_mode(self) -> str:
        return "rccl" if int(self.lib.sts_multi_gather_mode(self.h)) == 1 else "download"

    def rccl_ranks(self) -> int:
        """Size of the handle's communicator as RCCL reports it (ncclCommCount); 0 without an RCCL gather."""
        self.lib.sts_multi_rccl_ranks.argtypes = [C.c_void_p]
        return int(self.lib.sts_multi_rccl_ranks(self.h))

    def last_gather_ms(self) -> float:
        self.lib.sts_multi_last_gather_ms.argtypes = [C.c_void_p]
        self.lib.sts_multi_last_gather_ms.restype = C.c_double
        return float(self.lib.sts_multi_last_gather_ms(self.h))

    def set_conv_math(self, mode):
        m = {"bf16x3": 0, "f32": 1, "bf16x3_all": 2, "f16x2": 3}.get(mode, mode)
        self.lib.sts_multi_set_conv_math.argtypes = [C.c_void_p, C.c_int]
        if self.lib.sts_multi_set_conv_math(self.h, int(m)) != 0:
            raise StsError(f"sts_multi_set_conv_math: {self.lib.sts_multi_last_error().decode()}")

    @staticmethod
    def set_rccl_library(path: Optional[str], allow_repeated_devices: bool = False):
        """Test hook (sts_multi_set_rccl_library): which shared library provides the nccl* entry points; only before the first
        'rccl' handle of the process."""
        lib = load_library()
        lib.sts_multi_set_rccl_library.argtypes = [C.c_char_p, C.c_int]
        lib.sts_multi_last_error.restype = C.c_char_p
        rc = lib.sts_multi_set_rccl_library(path.encode() if path else None, 1 if allow_repeated_devices else 0)
        if rc != 0:
            raise StsError(f"sts_multi_set_rccl_library: {rc}: {lib.sts_multi_last_error().decode()}")

    def device_count(self) -> int:
        return int(self.lib.sts_multi_device_count(self.h))

    def shard_of(self, lengths: Sequence[int]) -> np.ndarray:
        n = np.ascontiguousarray(lengths, dtype=np.int32)
        out = np.zeros(n.size, np.int32)
        rc = self.lib.sts_multi_shard_of(self.h, n.size, n.ctypes.data, out.ctypes.data)
        if rc != 0:
            raise StsError(f"sts_multi_shard_of: {rc}: {self.lib.sts_multi_last_error().decode()}")
        return out

    def infer_batch(self, ids, sid=None, length_scale=None) -> List[np.ndarray]:
        B = len(ids)
        arrs = [np.ascontiguousarray(a, dtype=np.int32) for a in ids]
        ptrs = (C.c_void_p * B)(*[a.ctypes.data for a in arrs])
        n = np.asarray([a.size for a in arrs], dtype=np.int32)
        sidv = np.zeros(B, np.int32) if sid is None else np.ascontiguousarray(sid, dtype=np.int32)
        lsv = np.ones(B, np.float32) if length_scale is None else np.ascontiguousarray(length_scale, dtype=np.float32)
        outp = (C.POINTER(C.c_int16) * B)()
        n_out = np.zeros(B, np.int32)
        rc = self.lib.sts_multi_infer_ids_batch(self.h, B, ptrs, n.ctypes.data, sidv.ctypes.data, lsv.ctypes.data,
                                                C.cast(outp, C.c_void_p), n_out.ctypes.data)
        if rc != 0:
            raise StsError(f"sts_multi_infer_ids_batch: {rc}: {self.lib.sts_multi_last_error().decode()}")
        res = []
        for b in range(B):
            res.append(np.ctypeslib.as_array(outp[b], shape=(int(n_out[b]),)).copy() if n_out[b] else np.zeros(0, np.int16))
            self.lib.sts_free(C.cast(outp[b], C.c_void_p))
        return res

    def close(self):
        if self.h:
            self.lib.sts_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
