"""CPU suite: the optional host text frontend.  frontend/_ref/libsummertts_frontend.so is the reference's
own frontend compiled in place (frontend/Makefile) behind frontend/frontend_shim.cpp; the real model blobs -- the only
carriers of the WeTextProcessing FSTs and jieba dictionaries -- are absent, so what is checked here is what a synthetic blob
can carry: the English frontend end to end (its GRU matrices are synthesizable, its 125 k-word table is compiled in), the
section walk of the Chinese blob layout incl. the `off += off % 4` rule, and that the reference's UNMODIFIED demo
(test/main.cpp) links against this repo's libraries."""
import ctypes as C
import dataclasses
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from summertts_amd import engine, synth_blob as sb

FE = os.path.join(ROOT, "frontend", "_ref", "libsummertts_frontend.so")
REF = "/root/reference"


@pytest.fixture(scope="module")
def fe():
    if not os.path.exists(FE):
        if not os.path.isdir(REF):
            pytest.skip("libsummertts_frontend.so is built from /root/reference, which is absent here, and no prebuilt copy travelled")
        subprocess.run(["make", "-C", os.path.join(ROOT, "frontend"), "-j8"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(FE)
    lib.stsfe_create.restype = C.c_void_p
    lib.stsfe_create.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
    lib.stsfe_text_to_ids.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]
    lib.stsfe_sections_end.restype = C.c_int64
    lib.stsfe_sections_end.argtypes = [C.c_void_p]
    lib.stsfe_scan_sections.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]
    lib.stsfe_destroy.argtypes = [C.c_void_p]
    lib.stsfe_free.argtypes = [C.c_void_p]
    return lib


def text_to_ids(lib, h, text):
    p, n = C.POINTER(C.c_int32)(), C.c_int32()
    assert lib.stsfe_text_to_ids(h, text.encode(), C.byref(p), C.byref(n)) == 0
    ids = np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
    lib.stsfe_free(C.cast(p, C.c_void_p))
    return ids


def eng_blob(seed=5):
    cfg = dataclasses.replace(sb.tiny_cfg("mbb_fix"), lang=1, vocab=sb.ENG_IPA_SYMBOLS)
    ac = sb.make_blob(cfg, seed)
    return cfg, ac, np.concatenate([ac, sb.eng_frontend_section()])


def test_english_frontend_consumes_its_section_and_emits_ids_in_range(fe):
    cfg, ac, blob = eng_blob()
    h = fe.stsfe_create(blob.ctypes.data, blob.nbytes, ac.size, 1)
    assert h
    assert fe.stsfe_sections_end(h) == blob.size                 # EnglishText2Id.cpp:74-131 read exactly the section
    a = text_to_ids(fe, h, "hello world, this is a test.")
    assert a.size > 20 and a.min() >= 0 and a.max() < sb.ENG_IPA_SYMBOLS
    assert np.array_equal(a, text_to_ids(fe, h, "hello world, this is a test."))
    # EnglishText2Id.cpp:590-604: every symbol is preceded by a blank (id 0) and every word ends with (0, 16 = ' ')
    assert a[0] == 0 and a[-2] == 0 and a[-1] == 16
    # digits are spelled out before the lookup (replaceNum, EnglishText2Id.cpp:308-379); an OOV word runs the GRU fallback
    assert np.array_equal(text_to_ids(fe, h, "7"), text_to_ids(fe, h, " seven "))
    oov = text_to_ids(fe, h, "flibbertigibbetish")
    assert oov.size > 0 and oov.max() < sb.ENG_IPA_SYMBOLS
    fe.stsfe_destroy(h)
    # a weights-only blob has no frontend
    assert not fe.stsfe_create(ac.ctypes.data, ac.nbytes, ac.size, 1)


@pytest.mark.parametrize("sizes", [(5, 3, (7, 2, 0, 1, 9), 6, 5), (8, 8, (4, 4, 4, 4, 4), 8, 4), (1, 0, (0, 0, 0, 0, 3), 2, 1),
                                   (13, 2, (1, 1, 1, 1, 2), 3, 3)])
def test_chinese_section_walk_follows_the_reference_alignment_rule(fe, sizes):
    """SynthesizerTrn.cpp:181-297: [2 sizes + bytes] [5 sizes + bytes] [2 sizes + bytes]; after each section
    off_char += off_char % 4 (remainder 1 -> +1, 2 -> +2, 3 -> +3), off = off_char / 4."""
    tg, vb, jb, pw, pp = sizes
    ac = sb.make_blob(sb.tiny_cfg("mbb_fix"), 5)
    rng = np.random.default_rng(sum(jb) + tg)
    mk = lambda n: bytes(rng.integers(1, 255, size=n, dtype=np.uint8))
    chunks = (mk(tg), mk(vb), [mk(n) for n in jb], mk(pw), mk(pp))
    try:
        sec, info = sb.chs_frontend_sections(chunks[0], chunks[1], chunks[2], chunks[3], chunks[4], ac.size)
    except AssertionError:
        pytest.skip("these sizes make the reference's walk land off the float grid of a writable stream")
    blob = np.concatenate([ac, sec])
    out = (C.c_int64 * 13)()
    assert fe.stsfe_scan_sections(blob.ctypes.data, blob.nbytes, ac.size, 0, out) == 0
    o = list(out)
    assert o[0] == info["tn"] and o[1] == tg and o[2] == vb
    assert o[3] == info["jieba"] and tuple(o[4:9]) == jb
    assert o[9] == info["poly"] and o[10] == pw and o[11] == pp and o[12] == info["end"]
    # python restatement of the rule, independent of the writer
    off = ac.size + 2
    oc = off * 4 + tg + vb
    oc += oc % 4
    assert o[3] == oc // 4 + 5
    raw = blob.view(np.uint8)
    assert bytes(raw[o[0] * 4: o[0] * 4 + tg]) == chunks[0]
    assert bytes(raw[o[9] * 4 + pw: o[9] * 4 + pw + pp]) == chunks[4]
    # truncated blob: the walk reports it instead of reading past the end
    assert fe.stsfe_scan_sections(blob.ctypes.data, blob.nbytes - 8, ac.size, 0, out) != 0 or pw + pp <= 8


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference's test/main.cpp")
def test_reference_demo_links_unmodified(fe, tmp_path):
    """The reference's own caller (test/main.cpp:75-148: Hanz2Piny helpers + ttsLoadModel + SynthesizerTrn + tts_free_data)
    compiles against the reference's headers and LINKS against libsummertts_hip.so (+ the frontend library for the
    Hanz2Piny symbols it uses to read the text file) without a single change."""
    if not os.path.exists(engine.LIB_PATH):
        engine.build_library()
    exe = tmp_path / "tts_test"
    r = subprocess.run(["g++", "-O1", "-std=c++11", "-w", f"-I{REF}/include", f"-I{REF}/src/header", f"{REF}/test/main.cpp",
                        "-L", os.path.dirname(engine.LIB_PATH), "-lsummertts_hip", "-L", os.path.dirname(FE), "-lsummertts_frontend",
                        "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH), "-Wl,-rpath," + os.path.dirname(FE),
                        "-Wl,--allow-shlib-undefined", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-u", str(exe)], capture_output=True, text=True).stdout
    assert "SynthesizerTrn" in syms and "ttsLoadModel" in syms       # resolved from the shared libraries at load time
