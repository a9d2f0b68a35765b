"""CPU suite: the oracle (C restatement) against the golden vectors produced by the real reference,
and against the real reference itself where oracle/_ref exists; the blob grammar."""
import numpy as np
import pytest

from conftest import (TAP_MAXABS_TOL, AMP_WAVE_MAXABS_TOL, AMP_WAVE_RMSE_TOL, assert_pcm_close, assert_pcm_close_wrapped, assert_wave_close, golden_files,
                      golden_files_v2, load_golden, load_golden_v2)
from oracle import pyref
from summertts_amd import synth_blob as sb

TINY = ["hifigan_sdp", "hifigan_fix", "mbb_fix", "ms_sdp", "istft_fix", "ms_hifigan_sdp", "ms_hifigan_fix", "odd"]


@pytest.mark.parametrize("path", golden_files(), ids=lambda p: p.split("/")[-1])
def test_port_matches_reference_golden(path, port_built):
    g, cfg, blob = load_golden(path)
    port = pyref.PortModel(blob)
    assert port.consumed == blob.size
    o = port.infer_ids(g["ids"], int(g["sid"]), float(g["length_scale"]), taps=True)
    assert (o["durations"] == g["durations"]).all(), "durations differ from the reference"
    assert_wave_close(o["wave"], g["wave"], "port vs reference golden")
    assert_pcm_close(o["pcm"], g["pcm"], "port vs reference golden")
    for k in ("m", "z", "logw"):
        assert np.abs(o[k] - g[k]).max() <= TAP_MAXABS_TOL, k


@pytest.mark.parametrize("kind", TINY)
def test_blob_grammar_consumed_exactly(kind, port_built):
    cfg = sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 7)
    assert pyref.PortModel(blob).consumed == blob.size
    if pyref.have_ref():
        assert pyref.RefModel(blob).consumed == blob.size   # the reference's own constructors


def test_full_size_blob_matches_survey_probe_sizes():
    # the reference constructors consumed exactly these float counts
    assert sb.make_blob(sb.full_cfg("hifigan_sdp"), 1).size == 29071424
    assert sb.make_blob(sb.full_cfg("mbb_fix"), 1).size == 27475791


@pytest.mark.skipif(not pyref.have_ref(), reason="oracle/_ref (real reference) not built on this machine")
@pytest.mark.parametrize("kind", TINY)
def test_port_matches_live_reference(kind, port_built):
    cfg = sb.tiny_cfg(kind)
    blob = sb.make_blob(cfg, 99)
    ids = sb.synthetic_ids(14, cfg.vocab, salt=5)
    r = pyref.RefModel(blob).infer_ids(ids, 1, 1.05, taps=True)
    p = pyref.PortModel(blob).infer_ids(ids, 1, 1.05, taps=True)
    assert (r["durations"] == p["durations"]).all()
    assert_wave_close(p["wave"], r["wave"], kind)
    assert_pcm_close(p["pcm"], r["pcm"], kind)
    for k in ("x_enc", "m", "logw", "z_p", "z"):
        assert np.abs(p[k] - r[k]).max() <= TAP_MAXABS_TOL, k


@pytest.mark.parametrize("path", [p for p in golden_files_v2("real_tiny_")], ids=lambda p: p.split("/")[-1])
def test_port_matches_reference_golden_on_realistic_weight_statistics(path, port_built):
    """Round 5: the restatement against outputs the compiled reference produced on blobs with the statistics of a trained,
    weight-normed checkpoint (synth_blob.py stats="realistic": log-normal per-channel gains, sparse 20x outliers, biases on every
    conv, LayerNorm gamma ~ U(0.5, 2) / beta ~ N(0, 0.3)) -- the oracle must not be pinned on i.i.d. Gaussian weights alone."""
    g, cfg, blob, utts, stride = load_golden_v2(path)
    assert cfg.stats == "realistic"
    port = pyref.PortModel(blob)
    assert port.consumed == blob.size
    for u, ids, sid, ls, dur, pcm, wave in utts:
        o = port.infer_ids(ids, sid, ls, taps=True)
        assert (o["durations"] == dur).all()
        assert_wave_close(o["wave"][::stride], wave, path)
        assert_pcm_close(o["pcm"], pcm, path)
        z = g[f"z_{u}"]
        assert np.abs(o["z"][:, ::int(g["z_stride"])] - z).max() <= TAP_MAXABS_TOL * max(1.0, float(np.abs(z).max()))


@pytest.mark.skipif(not pyref.have_ref(), reason="oracle/_ref (real reference) not built on this machine")
@pytest.mark.parametrize("kind", TINY)
def test_port_matches_live_reference_on_realistic_weight_statistics(kind, port_built):
    import dataclasses
    cfg = dataclasses.replace(sb.tiny_cfg(kind), stats="realistic")
    blob = sb.make_blob(cfg, 41)
    ids = sb.synthetic_ids(15, cfg.vocab, salt=2)
    r = pyref.RefModel(blob).infer_ids(ids, 1, 0.95, taps=True)
    p = pyref.PortModel(blob).infer_ids(ids, 1, 0.95, taps=True)
    assert (r["durations"] == p["durations"]).all()
    assert_wave_close(p["wave"], r["wave"], kind)
    assert_pcm_close(p["pcm"], r["pcm"], kind)
    for k in ("x_enc", "m", "logw", "z_p", "z"):
        assert np.abs(p[k] - r[k]).max() <= TAP_MAXABS_TOL * max(1.0, float(np.abs(r[k]).max())), k


def ms_blob_with_explicit_filter_bias(bias_value=0.0375, seed=17):
    """A full-size MS-iSTFT blob whose learned synthesis filter (multistream_conv_post: 1 x subbands x 63, nn_conv1d.cpp:32-46) carries an
    EXPLICIT bias: the has_bias flag of its header is set and the one bias float is written in place (the synthetic recipe only
    emits one under stats="realistic", with whatever value the generator draws)."""
    import dataclasses
    cfg = dataclasses.replace(sb.full_cfg("ms_fix"), stats="realistic")
    blob = sb.make_blob(cfg, seed).copy()
    hdr = np.array([1, cfg.subbands, 63, 31, 1, 1], np.float32)
    hits = [i for i in range(blob.size - 6) if blob[i] == 1.0 and np.array_equal(blob[i:i + 6], hdr)]
    assert len(hits) == 1, hits
    at = hits[0] + 6 + 63 * cfg.subbands
    blob[at] = np.float32(bias_value)
    return cfg, blob, at


@pytest.mark.skipif(not pyref.have_ref(), reason="oracle/_ref (real reference) not built on this machine")
def test_port_matches_live_reference_with_an_explicit_multistream_filter_bias(port_built):
    """Full-size MS-iSTFT decoder, the synthesis filter's bias set to 0.0375 (about 1 200 LSB): restatement == compiled reference, and the
    bias really is in the output (the same blob with the float zeroed gives a waveform lower by that amount)."""
    cfg, blob, at = ms_blob_with_explicit_filter_bias()
    ids = sb.synthetic_ids(9, cfg.vocab, salt=3)
    r = pyref.RefModel(blob).infer_ids(ids, 0, 1.0, taps=True)
    p = pyref.PortModel(blob).infer_ids(ids, 0, 1.0, taps=True)
    assert (r["durations"] == p["durations"]).all()
    assert_wave_close(p["wave"], r["wave"], "ms_fix, explicit filter bias")
    assert_pcm_close(p["pcm"], r["pcm"], "ms_fix, explicit filter bias")
    blob0 = blob.copy(); blob0[at] = 0.0
    p0 = pyref.PortModel(blob0).infer_ids(ids, 0, 1.0)
    assert np.abs((p["wave"] - p0["wave"]) - np.float32(0.0375)).max() <= 1e-6


def test_port_forced_durations_and_short_inputs(port_built):
    cfg = sb.tiny_cfg("hifigan_fix")
    blob = sb.make_blob(cfg, 3)
    port = pyref.PortModel(blob)
    for T in (1, 2, 4, 5):   # the reference asserts for T < window+1; the restatement stays total
        o = port.infer_ids(sb.synthetic_ids(T, cfg.vocab), 0, 1.0)
        assert o["wave"].size == int(o["durations"].sum()) * cfg.hop_total
    ids = sb.synthetic_ids(7, cfg.vocab)
    o = port.infer_ids(ids, 0, 1.0, forced_dur=[1, 0, 3, 2, 0, 1, 4])
    assert o["wave"].size == 11 * cfg.hop_total
    o0 = port.infer_ids(ids, 0, 1.0, forced_dur=[0] * 7)   # clamp_min(sum, 1): one zero frame
    assert o0["wave"].size == cfg.hop_total


def test_pcm_quantisation_rule(port_built):
    # (int16_t)(o * 32737): truncation toward zero with the literal 32737 (SynthesizerTrn.cpp:393-396)
    cfg = sb.tiny_cfg("hifigan_fix")
    o = pyref.PortModel(sb.make_blob(cfg, 3)).infer_ids(sb.synthetic_ids(6, cfg.vocab), 0, 1.0)
    expect = np.trunc(o["wave"].astype(np.float32) * np.float32(32737)).astype(np.int64)
    assert (o["pcm"].astype(np.int64) == expect).all()


@pytest.mark.parametrize("path", [p for p in golden_files_v2("amp_") if "full" not in p], ids=lambda p: p.split("/")[-1])
def test_port_matches_reference_at_the_amplitude_edge(path, port_built):
    """The restatement against the real reference's outputs at tanh saturation and beyond +-1.0 (int16 wrap-around of the
    unclipped cast, SynthesizerTrn.cpp:393-396)."""
    g, cfg, blob, utts, stride = load_golden_v2(path)
    port = pyref.PortModel(blob)
    for u, ids, sid, ls, dur, pcm, wave in utts:
        o = port.infer_ids(ids, sid, ls)
        assert (o["durations"] == dur).all()
        peak = max(1.0, float(np.abs(wave).max()))
        assert np.abs(o["wave"][::stride].astype(np.float64) - wave).max() <= AMP_WAVE_MAXABS_TOL * peak
        assert_pcm_close_wrapped(o["pcm"], pcm, path)


def test_full_size_golden_fixtures_are_well_formed():
    """The full-size fixtures are too slow for the restatement on a CPU-only box; check what can be checked without
    running a model: recipe hash, batch definition, PCM == trunc(wave * 32737) on the stored (strided) samples."""
    paths = golden_files_v2("full_")
    assert len(paths) >= 10     # round 3 added the bench's own T = 128 workload and the 32- / 64-utterance batches, round 5 the full-size iSTFT decoder
    real = [p for p in golden_files_v2("real_") if "real_tiny_" not in p]
    assert len(real) >= 4       # round 5: realistic weight statistics, incl. the iSTFT decoder
    for path in paths + real:
        g = np.load(path)
        assert str(g["size"]) == "full" and int(g["wave_stride"]) == 8
        for u in g["utts"]:
            pcm, wave, dur = g[f"pcm_{u}"], g[f"wave_{u}"], g[f"dur_{u}"]
            assert pcm.size % int(dur.sum()) == 0 and pcm.size // int(dur.sum()) == 256
            if f"z_{u}" in g:
                assert g[f"z_{u}"].shape[1] == (int(dur.sum()) + int(g["z_stride"]) - 1) // int(g["z_stride"])
            expect = np.trunc(wave.astype(np.float32) * np.float32(32737)).astype(np.int64)
            assert (pcm[::8].astype(np.int64) == expect).all()
            assert g[f"ids_{u}"].size >= 64
        if "batch_lens" in g:
            assert len(g["batch_lens"]) in (8, 32, 64) and 64 <= min(g["batch_lens"]) and max(g["batch_lens"]) <= 256
