"""CPU suite: the C-ABI library loads, exports every declared symbol and refuses to run without a GPU
(no CPU fallback); the C++ drop-in class keeps the reference's mangled symbols."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from summertts_amd import engine


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    return engine.load_library()


def test_exports_every_symbol_declared_in_the_header(lib):
    hdr = open(os.path.join(ROOT, "include", "summertts_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(sts_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in summertts_hip.h but not exported"
    assert set(engine.EXPORTED_SYMBOLS) == set(declared)


def test_reference_cxx_surface_is_exported():
    syms = subprocess.run(["nm", "-D", "--defined-only", engine.LIB_PATH], capture_output=True, text=True, check=True).stdout
    # mangled names of include/SynthesizerTrn.h:9-19 and include/utils.h:4-5 as g++ emits them for the reference
    for s in ("_ZN14SynthesizerTrnC1EPfi", "_ZN14SynthesizerTrn5inferERKNSt7__cxx1112basic_stringIcSt11char_traitsIcESaIcEEEifRi",
              "_ZN14SynthesizerTrn13getSpeakerNumEv", "_ZN14SynthesizerTrnD1Ev", "_Z12ttsLoadModelPcPPf", "_Z13tts_free_dataPv",
              "_Z7tts_log13TTS_LOG_CAT_tPKc"):
        assert s in syms, s


def test_shipped_library_reads_only_the_documented_environment_variables():
    """The product is not the lab bench: every experiment knob lives behind -DSTS_EXPERIMENTS (summertts_amd/csrc/knobs.hpp,
    `make exp`); the default build must not even contain another variable name, so a stray STS_* in the environment
    cannot change tiles or arithmetic of the drop-in library."""
    out = subprocess.run(["strings", "-n", "6", engine.LIB_PATH], capture_output=True, text=True, check=True).stdout
    names = sorted({ln.strip() for ln in out.splitlines() if re.fullmatch(r"(STS|SUMMERTTS)_[A-Z0-9_]+", ln.strip())})
    names = [n for n in names if not re.fullmatch(r"STS_(OK|E[A-Z]+|DBG_[A-Z_]+)", n)]
    assert names == ["STS_CONV_MATH", "STS_TEST_HOOKS", "SUMMERTTS_FRONTEND_LIB", "SUMMERTTS_HIP_DEVICE"], names
    src = os.path.join(ROOT, "summertts_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".hip", ".hpp")) and f != "knobs.hpp":
            for m in re.finditer(r'getenv\("([A-Z_0-9]+)"\)', open(os.path.join(src, f)).read()):
                assert m.group(1) in ("STS_CONV_MATH", "STS_TEST_HOOKS", "SUMMERTTS_FRONTEND_LIB", "SUMMERTTS_HIP_DEVICE"), (f, m.group(1))


def _gpu_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_visible(), reason="this check is about machines without a GPU")
def test_fails_loudly_without_gpu(lib):
    with pytest.raises(engine.StsError) as ei:
        engine.Synthesizer(np.zeros(64, np.float32))
    assert "no CPU fallback" in str(ei.value)
    with pytest.raises(engine.StsError):
        engine.debug_conv1d(np.zeros((4, 8), np.float32), np.zeros((4, 1, 4), np.float32), None, 0)


def test_prepared_batch_builds_the_argument_arrays_of_run_batch():
    """Synthesizer.prepare (engine.PreparedBatch): the arrays sts_run_batch takes, built once -- ragged ids, defaults for speaker ids and
    length scales, and pointers that stay valid for the life of the object (the bench step re-submits one object every step)."""
    import ctypes as C
    ids = [[1, 2, 3], np.arange(7, dtype=np.int64), (4,)]
    p = engine.PreparedBatch(ids)
    assert len(p) == 3 and p.n.tolist() == [3, 7, 1] and p.n.dtype == np.int32
    assert p.sid.tolist() == [0, 0, 0] and p.ls.tolist() == [1.0, 1.0, 1.0] and p.ls.dtype == np.float32
    for b, want in enumerate(ids):
        got = np.ctypeslib.as_array(C.cast(p.ptrs[b], C.POINTER(C.c_int32)), shape=(len(want),))
        assert got.tolist() == [int(v) for v in want]
    assert (p.n_p, p.sid_p, p.ls_p, p.n_out_p) == (p.n.ctypes.data, p.sid.ctypes.data, p.ls.ctypes.data, p.n_out.ctypes.data)
    q = engine.PreparedBatch(ids, sid=[2, 0, 1], length_scale=[0.5, 1.0, 2.0])
    assert q.sid.tolist() == [2, 0, 1] and q.ls.tolist() == [0.5, 1.0, 2.0]


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under summertts_amd/ may import, link or dlopen it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "summertts_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "vits_oracle" not in txt and "libsummertts_ref" not in txt and "pyref" not in txt, os.path.join(dirpath, f)
    ldd = subprocess.run(["ldd", engine.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


@pytest.mark.parametrize("k,dil", [(3, 1), (5, 2), (7, 3), (11, 5), (13, 1), (2, 1), (4, 1)])
def test_winograd_weight_transform_reproduces_the_direct_conv(k, dil):
    """Host logic of the Winograd-domain layer kernels: the packed U = G g (segmented F(2,3), 3-tap segments first,
    then 2-tap ones) pushed through the kernel's arithmetic -- per segment M0 += U0 (X0 - X2), M1 += U1 (X1 + X2),
    M2 += U2 (X2 - X1), M3 += U3 (X1 - X3); y[n] = M0 + M1 + M2, y[n + d] = M1 - M2 - M3 -- equals the direct conv."""
    from summertts_amd import engine
    rng = np.random.default_rng(k * 10 + dil)
    co, ci, npairs = 5, 3, 40
    w = rng.standard_normal((co, k, ci)).astype(np.float32)
    U = engine.debug_wino_pack(w).astype(np.float64)          # [seg][4][16][32]
    n3 = {0: k // 3, 2: (k - 2) // 3, 1: (k - 4) // 3}[k % 3]
    nseg = U.shape[0]
    assert nseg == n3 + {0: 0, 2: 1, 1: 2}[k % 3] and U.shape[1:] == (4, 16, 32)
    assert not U[:, :, ci:, :].any() and not U[:, :, :, co:].any()                      # padding stays zero
    L = 2 * dil * npairs
    x = rng.standard_normal((ci, L + (k + 1) * dil))
    direct = np.zeros((co, L))
    for j in range(k):
        direct += w[:, j, :].astype(np.float64) @ x[:, j * dil:j * dil + L]
    n = np.arange(L).reshape(-1, 2, dil)[:, 0, :].reshape(-1)     # first position of every output pair
    M = [np.zeros((co, n.size)) for _ in range(4)]
    for sg in range(nseg):
        j0 = 3 * sg if sg < n3 else 3 * n3 + 2 * (sg - n3)
        X = [x[:, n + (j0 + m) * dil] for m in range(4)]
        V = [X[0] - X[2], X[1] + X[2], X[2] - X[1], X[1] - X[3]]
        for xi in range(4 if sg < n3 else 3):
            M[xi] += U[sg, xi, :ci, :co].T @ V[xi]
        if sg >= n3:
            assert not U[sg, 3].any()                                                       # 2-tap segment: no fourth product
    got = np.zeros((co, L))
    got[:, n] = M[0] + M[1] + M[2]
    got[:, n + dil] = M[1] - M[2] - M[3]
    assert np.abs(got - direct).max() < 1e-5


def test_multi_device_gather_layout_and_flag_validation(lib):
    """Host logic of the native RCCL gather (multi.hip): every rank's block of the gather buffer starts 256-byte aligned, empty
    ranks take no room, blocks do not overlap; and the argument checks that run before any device is touched."""
    off, total = engine.multi_gather_layout([171008, 0, 5, 128, 129])
    assert off.tolist() == [0, 171008, 171008, 171136, 171264] and total == 171264 + 256
    assert all(int(o) % 128 == 0 for o in off)
    off, total = engine.multi_gather_layout([0, 0])
    assert off.tolist() == [0, 0] and total == 0
    blob = np.zeros(64, np.float32)
    with pytest.raises(engine.StsError, match="distinct devices"):
        engine.MultiDevice(blob, [0, 0], gather="rccl")
    lib.sts_multi_create_ex.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    h = C.c_void_p()
    dev = np.zeros(1, np.int32)
    assert lib.sts_multi_create_ex(blob.ctypes.data, blob.nbytes, dev.ctypes.data, 1, 7, C.byref(h)) < 0      # unknown flags


def _header_struct_fields(name):
    """(type, field) pairs of `typedef struct <name> { ... } <name>;` in include/summertts_hip.h, in declaration order."""
    src = open(os.path.join(ROOT, "include", "summertts_hip.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        ctype, names = decl.split(None, 1)
        out += [(ctype, n.strip()) for n in names.split(",")]
    return out


@pytest.mark.parametrize("cname, mirror", [("sts_profile", engine.Profile), ("sts_model_info", engine.ModelInfo)])
def test_python_struct_mirrors_match_the_header(cname, mirror, tmp_path):
    """The ctypes mirrors of the ABI structs (summertts_amd/engine.py) name the header's fields in the header's order with the
    header's types, and have the size the C compiler gives the struct -- a field appended on one side only (round 3 added
    conv_math_fallbacks) would silently shift every later read."""
    ctypes_of = {"float": C.c_float, "double": C.c_double, "int32_t": C.c_int32, "int64_t": C.c_int64}
    want = [(n, ctypes_of[t]) for t, n in _header_struct_fields(cname)]
    assert [(n, t) for n, t in mirror._fields_] == want
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include "summertts_hip.h"\nint main(void) { printf("%%zu", sizeof(%s)); return 0; }\n' % cname)
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    assert int(subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout) == C.sizeof(mirror)
