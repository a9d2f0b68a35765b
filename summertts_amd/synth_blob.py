"""Synthetic SummerTTS ``.bin`` model-blob writer.

The reference ships its five ``models/*.bin`` files out-of-tree (they are absent
from /root/reference, see ``.MISSING_LARGE_BLOBS``), so every test and bench in
this repo runs on seeded random-weight blobs written in the reference's own
float-stream grammar.  The grammar is read off the reference constructors; each
writer below cites the constructor it mirrors (paths relative to /root/reference):

* header + section order ....... src/models/SynthesizerTrn.cpp:103-163
* TextEncoder .................. src/models/TextEncoder.cpp:32-44
* attention_encoder ............ src/modules/attention_encoder.cpp:30-55  (grouped by kind!)
* multi_head_attention ......... src/modules/multi_head_attention.cpp:40-90
* FFN .......................... src/modules/ffn.cpp:27-30
* nn_layer_norm ................ src/nn_op/nn_layer_norm.cpp:17-31
* nn_conv1d .................... src/nn_op/nn_conv1d.cpp:25-52   (W memory order [out][k][in])
* nn_conv1d_transposed ......... src/nn_op/nn_conv1d_transposed.cpp:24-53 ([out][k][in] too)
* Generator_hifiGan ............ src/models/Generator_hifigan.cpp:44-101
* Generator_MS/Istft/MBB ....... src/models/Generator_MS.cpp:51-127, Generator_Istft.cpp:49-113,
                                 Generator_MBB.cpp:51-106
* ResBlock1 .................... src/modules/ResBlock1.cpp:27-38
* ResidualCouplingBlock/Layer .. src/models/ResidualCouplingBlock.cpp:29-39,
                                 src/modules/ResidualCouplingLayer.cpp:28-30
* WN ........................... src/modules/WN.cpp:32-60
* StochasticDurationPredictor .. src/models/StochasticDurationPredictor.cpp:41-70
* FixDurationPredictor ......... src/models/FixDurationPredictor.cpp:33-44
* ElementwiseAffine / ConvFlow / DDSConv
                                 src/modules/ElementwiseAffine.cpp:28-32, ConvFlow.cpp:37-41,
                                 DDSConv.cpp:29-59

All values are float32 (integers stored as floats).  The weights are drawn so
that activations stay O(1) through ~100 layers and the waveform stays inside
(-1, 1): conv weights ~ N(0, gain/sqrt(fan_in)), LayerNorm gamma=1/beta=0, the
flow ``post`` convs and the decoder's last conv scaled down.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Sequence, Tuple

import numpy as np

DEC_HIFIGAN, DEC_MS, DEC_ISTFT, DEC_MBB = 0, 1, 2, 3
DUR_STOCHASTIC, DUR_FIX = 0, 1


@dataclass
class ModelCfg:
    # header (SynthesizerTrn.cpp:103-106)
    is_ms: int = 0
    lang: int = 0
    dur_type: int = DUR_STOCHASTIC
    dec_type: int = DEC_HIFIGAN
    # text encoder
    hidden: int = 192
    inter: int = 192          # flow/latent channels (proj emits 2*inter)
    ffn: int = 768
    ffn_k: int = 3
    n_layers: int = 6
    window: int = 4
    vocab: int = 219
    # flow
    flow_n: int = 4
    flow_hidden: int = 192
    flow_layers: int = 4
    flow_k: int = 5
    # stochastic duration predictor
    sdp_filter: int = 192
    sdp_k: int = 3
    sdp_flows: int = 4
    dds_layers: int = 3
    # fixed duration predictor
    fix_filter: int = 256
    fix_k: int = 3
    # decoder
    up_rates: Tuple[int, ...] = (8, 8, 2, 2)
    up_init: int = 512
    up_k: Tuple[int, ...] = (16, 16, 4, 4)
    res_k: Tuple[int, ...] = (3, 7, 11)
    res_d: Tuple[Tuple[int, int, int], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    subbands: int = 4
    nfft: int = 16
    hop: int = 4
    # speakers
    spk_num: int = 0
    gin: int = 0
    # synthetic-duration control: logw ~= dur_bias (+- small)
    dur_bias: float = 1.55
    # output-amplitude controls (defaults keep |o| ~ 0.05-0.2; the amplitude-edge tests raise them):
    # gain of the HiFi-GAN conv_post weights (tanh saturation) / of the iSTFT heads, and the bias added to the
    # log-magnitude rows of the MB-iSTFT heads (|X_k| ~ e^mag_bias: > 0 drives |o| past 1.0 -> int16 wrap-around)
    post_gain: float = 0.35
    head_gain: float = 0.25
    mag_bias: float = -1.0
    istft_mag_bias: float = 0.0
    # weight statistics: "gaussian" = i.i.d. N(0, gain / sqrt(fan_in)) (rounds 1-4); "realistic" (round 5) = what a
    # weight-normed, trained checkpoint looks like to the kernels' range logic: a log-normal gain per output channel (sigma 1), one
    # weight in a thousand 20x larger, every conv with a bias of O(0.1), LayerNorm gamma ~ U(0.5, 2) and beta ~ N(0, 0.3) -- each
    # tensor rescaled to the rms the Gaussian recipe gives it, so that activations stay O(1) through the ~100 layers
    stats: str = "gaussian"

    @property
    def hop_total(self) -> int:
        up = int(np.prod(self.up_rates))
        if self.dec_type == DEC_HIFIGAN:
            return up
        if self.dec_type == DEC_ISTFT:
            return up * 4
        return up * 4 * 4


def full_cfg(kind: str) -> ModelCfg:
    """Upstream-default sized configurations (dims marked as assumed)."""
    if kind == "hifigan_sdp":           # VITS: HiFi-GAN decoder + stochastic duration predictor
        return ModelCfg(dur_bias=2.4)   # calibrated so that F/T ~ 5.5 frames per phoneme
    if kind == "mbb_fix":               # MB-iSTFT-VITS (PQMF) + deterministic duration predictor
        return ModelCfg(dur_type=DUR_FIX, dec_type=DEC_MBB, up_rates=(4, 4), up_k=(16, 16))
    if kind == "ms_fix":                # MS-iSTFT-VITS (learned synthesis filter)
        return ModelCfg(dur_type=DUR_FIX, dec_type=DEC_MS, up_rates=(4, 4), up_k=(16, 16))
    if kind == "ms_sdp":                # MS-iSTFT-VITS decoder + stochastic duration predictor
        return ModelCfg(dec_type=DEC_MS, up_rates=(4, 4), up_k=(16, 16), dur_bias=1.0)   # ~5.4 frames per phoneme
    if kind == "istft_fix":
        return ModelCfg(dur_type=DUR_FIX, dec_type=DEC_ISTFT, up_rates=(8, 8), up_k=(16, 16))
    if kind == "ms_hifigan_sdp":        # multi-speaker (aishell3-like)
        return ModelCfg(is_ms=1, spk_num=174, gin=256, dur_bias=1.4)
    raise ValueError(kind)


def tiny_cfg(kind: str) -> ModelCfg:
    """Small configurations for parity tests (seconds on one CPU core).  Channel counts are
    multiples of 32 where the matrix-core conv path should be exercised and deliberately
    odd elsewhere."""
    base = dict(hidden=64, inter=64, ffn=96, n_layers=2, vocab=40, flow_n=4, flow_hidden=64,
                flow_layers=2, sdp_filter=32, fix_filter=48, up_init=64, res_k=(3, 5),
                res_d=((1, 3, 5), (1, 2, 3)))
    if kind == "hifigan_sdp":
        return ModelCfg(up_rates=(4, 2), up_k=(8, 4), **base)
    if kind == "hifigan_fix":
        return ModelCfg(dur_type=DUR_FIX, up_rates=(4, 2), up_k=(8, 4), **base)
    if kind == "mbb_fix":
        return ModelCfg(dur_type=DUR_FIX, dec_type=DEC_MBB, up_rates=(2, 2), up_k=(4, 4), **base)
    if kind == "ms_sdp":
        return ModelCfg(dec_type=DEC_MS, up_rates=(2, 2), up_k=(4, 4), **base)
    if kind == "istft_fix":
        return ModelCfg(dur_type=DUR_FIX, dec_type=DEC_ISTFT, up_rates=(2, 2), up_k=(4, 4), **base)
    if kind == "ms_hifigan_sdp":
        return ModelCfg(is_ms=1, spk_num=5, gin=16, up_rates=(4, 2), up_k=(8, 4), **base)
    if kind == "ms_hifigan_fix":
        return ModelCfg(is_ms=1, spk_num=3, gin=16, dur_type=DUR_FIX, up_rates=(4, 2), up_k=(8, 4), **base)
    if kind == "odd":  # channel counts that are not multiples of 32 anywhere
        return ModelCfg(hidden=24, inter=20, ffn=40, n_layers=1, vocab=17, flow_n=3, flow_hidden=24,
                        flow_layers=2, sdp_filter=12, fix_filter=20, up_init=24, up_rates=(3, 2),
                        up_k=(7, 4), res_k=(3,), res_d=((1, 2, 3),), dur_type=DUR_FIX)
    raise ValueError(kind)


class _W:
    """Append-only float32 stream."""

    def __init__(self, seed: int, stats: str = "gaussian"):
        if stats not in ("gaussian", "realistic"):
            raise ValueError(stats)
        self.rng = np.random.default_rng(seed)
        self.realistic = stats == "realistic"
        self.parts: List[np.ndarray] = []
        self.n = 0

    def shape_weights(self, wt: np.ndarray) -> np.ndarray:
        """realistic: per-output-channel log-normal gain, sparse outliers, the tensor's rms kept (wt is [out][k][in])."""
        if not self.realistic or wt.size < 2:
            return wt
        rms0 = float(np.sqrt((wt.astype(np.float64) ** 2).mean()))
        g = np.exp(self.rng.standard_normal((wt.shape[0],) + (1,) * (wt.ndim - 1)))
        out = wt.astype(np.float64) * g
        out[self.rng.random(wt.shape) < 1e-3] *= 20.0
        rms1 = float(np.sqrt((out ** 2).mean()))
        return (out * (rms0 / rms1 if rms1 > 0 else 1.0)).astype(np.float32)

    def bias(self, n: int, std: float) -> np.ndarray:
        return self.normal((n,), max(std, 0.1) if self.realistic else std)

    def ints(self, *vals):
        a = np.asarray(vals, dtype=np.float32).ravel()
        self.parts.append(a)
        self.n += a.size

    def arr(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        self.parts.append(a)
        self.n += a.size

    def normal(self, shape, std):
        return (self.rng.standard_normal(shape) * std).astype(np.float32)

    def blob(self) -> np.ndarray:
        return np.concatenate(self.parts) if self.parts else np.zeros(0, np.float32)


def _conv1d(w: _W, out_ch, in_ch, k, pad=0, dil=1, bias=True, gain=0.7, bias_std=0.02):
    """nn_conv1d.cpp:32-46.  W memory order [out][k][in]; depthwise convs carry in_ch=1."""
    bias = bias or w.realistic          # ("biases everywhere": also the tails upstream VITS builds without one)
    w.ints(out_ch, in_ch, k, pad, dil, 1 if bias else 0)
    w.arr(w.shape_weights(w.normal((out_ch, k, in_ch), gain / np.sqrt(in_ch * k))))
    if bias:
        w.arr(w.bias(out_ch, bias_std))


def _convT1d(w: _W, out_ch, in_ch, k, stride, pad, bias=True, gain=0.7):
    """nn_conv1d_transposed.cpp:33-48.  W memory order [out][k][in]."""
    w.ints(out_ch, in_ch, k, pad, 1, 1 if bias else 0, stride)
    # each output sample receives ~k/stride taps * in_ch products
    w.arr(w.shape_weights(w.normal((out_ch, k, in_ch), gain / np.sqrt(in_ch * max(1.0, k / stride)))))
    if bias:
        w.arr(w.bias(out_ch, 0.02))


def _ln(w: _W, size):
    """nn_layer_norm.cpp:17-31."""
    w.ints(size)
    if w.realistic:
        w.arr(w.rng.uniform(0.5, 2.0, size).astype(np.float32))
        w.arr(w.normal((size,), 0.3))
        return
    w.arr(1.0 + w.normal((size,), 0.05))
    w.arr(w.normal((size,), 0.05))


def _mha(w: _W, cfg: ModelCfg):
    """multi_head_attention.cpp:40-90 (2 heads are hard-coded in the reference's output assembly)."""
    h = cfg.hidden
    kc = h // 2
    w.ints(h, h, 2, cfg.window)
    if cfg.window:
        px = 2 * cfg.window + 1
        for _ in range(2):  # relK, relV ; Eigen column-major [px, kc]
            w.ints(px, kc)
            w.arr(w.normal((kc, px), kc ** -0.5))  # memory = column-major => [col=kc][row=px]
    for _ in range(4):  # q, k, v, o
        _conv1d(w, h, h, 1, gain=1.0)


def _text_encoder(w: _W, cfg: ModelCfg):
    h = cfg.hidden
    w.ints(h, cfg.vocab, h)
    # emb_(v,e) = ptr[e*vocab + v]  (TextEncoder.cpp:36-38)
    w.arr(w.normal((h, cfg.vocab), h ** -0.5))
    w.ints(cfg.n_layers)
    for _ in range(cfg.n_layers):
        _mha(w, cfg)
    for _ in range(cfg.n_layers):
        _ln(w, h)
    for _ in range(cfg.n_layers):
        w.ints(cfg.ffn_k)
        _conv1d(w, cfg.ffn, h, cfg.ffn_k, gain=1.0)
        _conv1d(w, h, cfg.ffn, cfg.ffn_k, gain=1.0)
    for _ in range(cfg.n_layers):
        _ln(w, h)
    _conv1d(w, 2 * cfg.inter, h, 1, gain=1.0)


def _resblock1(w: _W, ch, k, dils):
    """ResBlock1.cpp:27-38: nBlocks, convs1 (dilated) x n, convs2 (d=1) x n; header pad/dil are used."""
    w.ints(len(dils))
    for d in dils:
        _conv1d(w, ch, ch, k, pad=(k * d - d) // 2, dil=d, gain=0.6)
    for _ in dils:
        _conv1d(w, ch, ch, k, pad=(k - 1) // 2, dil=1, gain=0.35)


def _gen_hdr(w: _W, cfg: ModelCfg):
    w.ints(len(cfg.up_rates), *cfg.up_rates, cfg.up_init, len(cfg.up_k), *cfg.up_k,
           len(cfg.res_k), *cfg.res_k, len(cfg.res_d))
    for d in cfg.res_d:
        w.ints(*d)


def _gen_body(w: _W, cfg: ModelCfg):
    _conv1d(w, cfg.up_init, cfg.inter, 7, pad=3, gain=1.0)
    ch = cfg.up_init
    for u, k in zip(cfg.up_rates, cfg.up_k):
        _convT1d(w, ch // 2, ch, k, u, (k - u) // 2)
        ch //= 2
    ch = cfg.up_init
    for _ in cfg.up_rates:
        ch //= 2
        for j, k in enumerate(cfg.res_k):
            _resblock1(w, ch, k, cfg.res_d[j])
    return ch


def _decoder(w: _W, cfg: ModelCfg):
    if cfg.dec_type == DEC_HIFIGAN:
        _gen_hdr(w, cfg)
        ch = _gen_body(w, cfg)
        _conv1d(w, 1, ch, 7, pad=3, bias=False, gain=cfg.post_gain)
        if cfg.is_ms:
            _conv1d(w, cfg.up_init, cfg.gin, 1, gain=0.3)
        return
    w.ints(cfg.subbands, cfg.nfft, cfg.hop)
    _gen_hdr(w, cfg)
    ch = _gen_body(w, cfg)
    nbin = cfg.nfft // 2 + 1
    if cfg.dec_type == DEC_ISTFT:
        # same stream as _conv1d(w, 2 * nbin, ch, 7, pad=3, gain=head_gain, bias_std=0.01), plus the log-magnitude shift
        w.ints(2 * nbin, ch, 7, 3, 1, 1)
        w.arr(w.shape_weights(w.normal((2 * nbin, 7, ch), cfg.head_gain / np.sqrt(ch * 7))))
        bs = w.normal((2 * nbin,), 0.01)
        bs[:nbin] += np.float32(cfg.istft_mag_bias)
        w.arr(bs)
        return
    # log-magnitude / phase heads; small so exp() stays O(1) and the waveform inside (-1,1)
    out = cfg.subbands * 2 * nbin
    w.ints(out, ch, 7, 3, 1, 1)
    wt = w.shape_weights(w.normal((out, 7, ch), cfg.head_gain / np.sqrt(ch * 7)))
    bs = w.normal((out,), 0.01)
    for b in range(cfg.subbands):       # push log-magnitudes down: |X_k| ~ e^-1
        bs[b * 2 * nbin: b * 2 * nbin + nbin] += np.float32(cfg.mag_bias)
    w.arr(wt)
    w.arr(bs)
    if cfg.dec_type == DEC_MS:
        _conv1d(w, 1, cfg.subbands, 63, pad=31, bias=False, gain=0.5)


def _wn(w: _W, cfg: ModelCfg):
    h = cfg.flow_hidden
    w.ints(cfg.flow_layers, cfg.flow_k)
    for _ in range(cfg.flow_layers):   # header pad/dil are overridden by WN.cpp:36-40
        _conv1d(w, 2 * h, h, cfg.flow_k, pad=(cfg.flow_k - 1) // 2, dil=1, gain=0.8)
    for i in range(cfg.flow_layers):
        last = i == cfg.flow_layers - 1
        _conv1d(w, h if last else 2 * h, h, 1, gain=0.8)
    if cfg.is_ms:
        _conv1d(w, 2 * h * cfg.flow_layers, cfg.gin, 1, gain=0.3)


def _flow(w: _W, cfg: ModelCfg):
    w.ints(cfg.flow_n, cfg.flow_layers)
    half = cfg.inter // 2
    for _ in range(cfg.flow_n):
        _conv1d(w, cfg.flow_hidden, half, 1, gain=1.0)
        _wn(w, cfg)
        _conv1d(w, half, cfg.flow_hidden, 1, gain=0.2, bias_std=0.01)


def _dds(w: _W, ch, k, layers):
    w.ints(layers, k)
    for i in range(layers):            # depthwise; header in_ch = 1
        d = k ** i
        _conv1d(w, ch, 1, k, pad=(k * d - d) // 2, dil=d, gain=1.0)
    for _ in range(layers):
        _conv1d(w, ch, ch, 1, gain=1.0)
    for _ in range(2 * layers):
        _ln(w, ch)


def _ea(w: _W, m, logs):
    w.arr(np.asarray(m, np.float32))
    w.arr(np.asarray(logs, np.float32))


def _convflow(w: _W, cfg: ModelCfg):
    f = cfg.sdp_filter
    _conv1d(w, f, 1, 1, gain=1.0)
    _dds(w, f, cfg.sdp_k, cfg.dds_layers)
    _conv1d(w, 29, f, 1, gain=0.5 * np.sqrt(f))


def _dur(w: _W, cfg: ModelCfg):
    if cfg.dur_type == DUR_STOCHASTIC:
        f = cfg.sdp_filter
        w.ints(cfg.sdp_flows)
        # inference output logw = (z0 - m) * exp(-logs): centre the durations on exp(dur_bias)
        _ea(w, [-cfg.dur_bias * 4.0, 0.1], [np.log(4.0), -0.2])
        for _ in range(cfg.sdp_flows):
            _convflow(w, cfg)
        _conv1d(w, f, 1, 1)            # post_pre   (posterior side: loaded, never run)
        _conv1d(w, f, f, 1)            # post_proj
        _dds(w, f, cfg.sdp_k, cfg.dds_layers)
        _ea(w, [0.0, 0.0], [0.0, 0.0])
        for _ in range(4):
            _convflow(w, cfg)
        _conv1d(w, f, cfg.hidden, 1, gain=1.0)   # pre
        _conv1d(w, f, f, 1, gain=1.0)            # proj
        _dds(w, f, cfg.sdp_k, cfg.dds_layers)
        if cfg.is_ms:
            _conv1d(w, f, cfg.gin, 1, gain=0.3)
    else:
        f = cfg.fix_filter
        k = cfg.fix_k
        _conv1d(w, f, cfg.hidden, k, pad=k // 2, gain=1.0)
        _ln(w, f)
        _conv1d(w, f, f, k, pad=k // 2, gain=1.0)
        _ln(w, f)
        # proj: small weights, bias = dur_bias
        w.ints(1, f, 1, 0, 1, 1)
        w.arr(w.shape_weights(w.normal((1, 1, f), 0.35 / np.sqrt(f))))
        w.arr(np.asarray([cfg.dur_bias], np.float32))
        if cfg.is_ms:
            _conv1d(w, cfg.hidden, cfg.gin, 1, gain=0.3)


def make_blob(cfg: ModelCfg, seed: int = 1234) -> np.ndarray:
    """Return the float32 model blob (acoustic sections only; no text-frontend sections)."""
    w = _W(seed, cfg.stats)
    w.ints(cfg.is_ms, cfg.lang, cfg.dur_type, cfg.dec_type)
    _text_encoder(w, cfg)
    _decoder(w, cfg)
    _flow(w, cfg)
    _dur(w, cfg)
    if cfg.is_ms:
        w.ints(cfg.spk_num, cfg.gin)
        # emb_g(s,c) = ptr[c*spkNum + s]  (SynthesizerTrn.cpp:159)
        w.arr(w.normal((cfg.gin, cfg.spk_num), 1.0))
    return w.blob()


ENG_IPA_SYMBOLS = 178     # len(ipaSymbols_) in /root/reference/src/engipa/EnglishText2Id.cpp:65: the id range of the English frontend
ENG_G2P_PHONES = 74       # id2Phone_ entries (EnglishText2Id.cpp:161-234)


def eng_frontend_section(seed: int = 7, hidden: int = 24) -> np.ndarray:
    """The frontend section that follows the acoustic sections of an ENGLISH model blob (lang = 1): the twelve matrices of the
    g2p GRU seq2seq fallback, in the order /root/reference/src/engipa/EnglishText2Id.cpp:74-131 reads them (matrices carry
    [rows, cols] headers, biases [n]; Eigen column-major; embedding width == hidden width, required by gru() :294-306).
    Random weights: dictionary words never touch them (the 125 k-word table is compiled into the frontend), only
    out-of-vocabulary words of >= 4 letters do."""
    w = _W(seed)
    h = hidden

    def mat(r, c):
        w.ints(r, c)
        w.arr(w.normal((c, r), 0.3))

    def vec(n):
        w.ints(n)
        w.arr(w.normal((n,), 0.1))

    mat(29, h); mat(3 * h, h); mat(3 * h, h); vec(3 * h); vec(3 * h)          # encoder: emb, w_ih, w_hh, b_ih, b_hh
    mat(ENG_G2P_PHONES, h); mat(3 * h, h); mat(3 * h, h); vec(3 * h); vec(3 * h)   # decoder
    mat(ENG_G2P_PHONES, h); vec(ENG_G2P_PHONES)                               # fc
    return w.blob()


def chs_frontend_sections(tagger: bytes, verbalizer: bytes, jieba: Sequence[bytes], poly_words: bytes, poly_pinyin: bytes,
                          float_offset: int) -> Tuple[np.ndarray, dict]:
    """The three frontend sections of a CHINESE model blob as /root/reference/src/models/SynthesizerTrn.cpp:181-297 walks
    them: [tagger size, verbalizer size, bytes...] [5 jieba sizes, bytes...] [2 polyphone sizes, bytes...], each section
    followed by the reference's alignment step  off_char += off_char % 4  (NOT a round-up: remainder 1 -> +1, 3 -> +3),
    then off = off_char / 4.  float_offset = float index at which the first section starts (the acoustic sections' end).
    Returns (float32 array to append, {section: float offset of its payload, 'end': float offset after the walk})."""
    out = bytearray()
    info = {}

    def cur_float():
        return float_offset + len(out) // 4

    def put_ints(*v):
        out.extend(np.asarray(v, np.float32).tobytes())

    def put_payload(chunks):
        start_char = float_offset * 4 + len(out)
        tot = sum(len(c) for c in chunks)
        for c in chunks:
            out.extend(c)
        off_char = start_char + tot
        if off_char % 4 > 0:
            off_char += off_char % 4
        nxt = off_char // 4                                      # where the reference continues reading floats
        need = nxt * 4 - (float_offset * 4 + len(out))
        if need > 0:
            out.extend(b"\0" * need)
        elif need < 0:                                           # the walk lands INSIDE the payload's last float: pad to it
            out.extend(b"\0" * ((-need) % 4))
        while len(out) % 4:
            out.extend(b"\0")
        return nxt

    put_ints(len(tagger), len(verbalizer)); info["tn"] = cur_float()
    nxt = put_payload([tagger, verbalizer]); info["after_tn"] = nxt
    assert nxt == cur_float(), "synthetic sizes must keep the walk on the written stream"
    put_ints(*[len(c) for c in jieba]); info["jieba"] = cur_float()
    nxt = put_payload(list(jieba)); info["after_jieba"] = nxt
    assert nxt == cur_float()
    put_ints(len(poly_words), len(poly_pinyin)); info["poly"] = cur_float()
    nxt = put_payload([poly_words, poly_pinyin]); info["end"] = nxt
    return np.frombuffer(bytes(out), dtype=np.float32).copy(), info


def synthetic_ids(n: int, vocab: int, salt: int = 0, family: int = 0) -> np.ndarray:
    """Fixed seeded phoneme-id sequence: ids[i] = (i*37 + 11 + salt) mod vocab.  family > 0: another sequence of the same
    length, ids[i] = (i*37 + 11 + salt + family * (i*i mod 101)) mod vocab -- bench.py's supply of requests an engine has not served before."""
    i = np.arange(n, dtype=np.int64)
    return ((i * 37 + 11 + salt + family * ((i * i) % 101)) % vocab).astype(np.int32)


def cfg_dict(cfg: ModelCfg) -> dict:
    return asdict(cfg)
