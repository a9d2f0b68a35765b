"""GPU suite: the reference-shaped C++ surface end to end (SynthesizerTrn class + ttsLoadModel + WAV writer)
through the demo CLI, against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import ROOT, assert_pcm_close
from oracle import pyref
from summertts_amd import engine, synth_blob as sb

pytestmark = pytest.mark.gpu


def build_cli(tmp_path):
    exe = tmp_path / "tts_ids"
    subprocess.run(["g++", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cli", "tts_ids.cpp"),
                    "-L", os.path.dirname(engine.LIB_PATH), "-lsummertts_hip", "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH),
                    "-o", str(exe)], check=True)
    return exe


def read_wav(path):
    raw = path.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<I", raw[24:28])[0] == 16000
    return np.frombuffer(raw[44:], dtype=np.int16)


def test_cli_text_input_through_the_reference_frontend(tmp_path):
    """SynthesizerTrn::infer(TEXT) on a blob that carries frontend sections -- here an English model: the
    reference's EnglishText2Id (frontend/_ref/libsummertts_frontend.so, compiled in place from the reference) turns the
    text into IPA ids on the host, lengthScale *= 0.83 (SynthesizerTrn.cpp:354), the HIP engine does the rest.  Checked
    against the oracle fed with the ids the same frontend emits.  (The Chinese frontend needs the FSTs / dictionaries of a
    real model blob, which are absent: not testable end to end.)"""
    import ctypes as C
    import dataclasses
    fe_path = os.path.join(ROOT, "frontend", "_ref", "libsummertts_frontend.so")
    if not os.path.exists(fe_path):
        pytest.skip("frontend/_ref/libsummertts_frontend.so did not travel (it is built from /root/reference)")
    exe = build_cli(tmp_path)
    cfg = dataclasses.replace(sb.tiny_cfg("mbb_fix"), lang=1, vocab=sb.ENG_IPA_SYMBOLS)
    ac = sb.make_blob(cfg, 31)
    blob = np.concatenate([ac, sb.eng_frontend_section()])
    (tmp_path / "m.bin").write_bytes(blob.tobytes())
    text = "hello world, this is 1 test of the text frontend."
    (tmp_path / "t.txt").write_text(text + "\n")
    out = tmp_path / "o.wav"
    r = subprocess.run([str(exe), str(tmp_path / "t.txt"), str(tmp_path / "m.bin"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    pcm = read_wav(out)
    lib = C.CDLL(fe_path)
    lib.stsfe_create.restype = C.c_void_p
    lib.stsfe_create.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
    lib.stsfe_text_to_ids.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_int32)]
    h = lib.stsfe_create(blob.ctypes.data, blob.nbytes, ac.size, 1)
    p, n = C.POINTER(C.c_int32)(), C.c_int32()
    assert lib.stsfe_text_to_ids(h, (text + "  ").encode(), C.byref(p), C.byref(n)) == 0      # the demo joins lines with "  "
    ids = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    assert ids.size > 40
    o = pyref.PortModel(blob).infer_ids(ids, 0, float(np.float32(1.0 * 0.83)))
    assert_pcm_close(pcm, o["pcm"], "text -> frontend -> HIP engine vs oracle")
    # the same model without the frontend library: text is refused with a message, ids still work
    env = dict(os.environ, SUMMERTTS_FRONTEND_LIB="/nonexistent.so")
    (tmp_path / "m2.bin").write_bytes(ac.tobytes())                 # weights-only blob: no frontend sections at all
    r = subprocess.run([str(exe), str(tmp_path / "t.txt"), str(tmp_path / "m2.bin"), str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 2 and "frontend" in r.stdout


def test_cli_matches_oracle(tmp_path):
    exe = build_cli(tmp_path)
    cfg = sb.tiny_cfg("mbb_fix")
    blob = sb.make_blob(cfg, 77)
    ids = sb.synthetic_ids(21, cfg.vocab, salt=4)
    (tmp_path / "m.bin").write_bytes(blob.tobytes())
    (tmp_path / "ids.txt").write_text(" ".join(str(int(i)) for i in ids[:10]) + "\n" + " ".join(str(int(i)) for i in ids[10:]) + "\n")
    out = tmp_path / "o.wav"
    r = subprocess.run([str(exe), str(tmp_path / "ids.txt"), str(tmp_path / "m.bin"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    pcm = read_wav(out)
    o = pyref.PortModel(blob).infer_ids(ids, 0, 1.0)
    assert_pcm_close(pcm, o["pcm"], "CLI vs oracle")
    # plain text is refused (frontend not wired), not guessed
    (tmp_path / "t.txt").write_text("hello world\n")
    r = subprocess.run([str(exe), str(tmp_path / "t.txt"), str(tmp_path / "m.bin"), str(out)], capture_output=True, text=True)
    assert r.returncode == 2 and "frontend" in r.stdout
