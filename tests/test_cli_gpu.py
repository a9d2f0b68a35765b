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


def test_cli_matches_oracle(tmp_path):
    exe = tmp_path / "tts_ids"
    subprocess.run(["g++", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "cli", "tts_ids.cpp"),
                    "-L", os.path.dirname(engine.LIB_PATH), "-lsummertts_hip", "-Wl,-rpath," + os.path.dirname(engine.LIB_PATH),
                    "-o", str(exe)], check=True)
    cfg = sb.tiny_cfg("mbb_fix")
    blob = sb.make_blob(cfg, 77)
    ids = sb.synthetic_ids(21, cfg.vocab, salt=4)
    (tmp_path / "m.bin").write_bytes(blob.tobytes())
    (tmp_path / "ids.txt").write_text(" ".join(str(int(i)) for i in ids[:10]) + "\n" + " ".join(str(int(i)) for i in ids[10:]) + "\n")
    out = tmp_path / "o.wav"
    r = subprocess.run([str(exe), str(tmp_path / "ids.txt"), str(tmp_path / "m.bin"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = out.read_bytes()
    assert raw[:4] == b"RIFF" and raw[8:16] == b"WAVEfmt " and struct.unpack("<I", raw[24:28])[0] == 16000
    pcm = np.frombuffer(raw[44:], dtype=np.int16)
    o = pyref.PortModel(blob).infer_ids(ids, 0, 1.0)
    assert_pcm_close(pcm, o["pcm"], "CLI vs oracle")
    # plain text is refused (frontend not wired), not guessed
    (tmp_path / "t.txt").write_text("hello world\n")
    r = subprocess.run([str(exe), str(tmp_path / "t.txt"), str(tmp_path / "m.bin"), str(out)], capture_output=True, text=True)
    assert r.returncode == 2 and "frontend" in r.stdout
