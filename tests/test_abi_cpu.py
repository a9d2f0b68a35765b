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


def test_product_never_touches_the_oracle():
    # the oracle is test infrastructure: nothing under summertts_amd/ may import, link or dlopen it
    for dirpath, _, files in os.walk(os.path.join(ROOT, "summertts_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "vits_oracle" not in txt and "libsummertts_ref" not in txt and "pyref" not in txt, os.path.join(dirpath, f)
    ldd = subprocess.run(["ldd", engine.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd
