// capi.hip -- extern "C" surface declared in include/summertts_hip.h.
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>

#include "engine.hpp"

using namespace sts;

struct sts_engine { Engine eng; };

static thread_local std::string g_err;
static int set_err(int code, const std::string& s) { g_err = s; return code; }

extern "C" {

const char* sts_last_error(void) { return g_err.c_str(); }
void sts_free(void* p) { free(p); }

int sts_create(const float* blob, int64_t blob_bytes, int device, sts_engine** out) {
    if (!out) return set_err(STS_EINVAL, "sts_create: null out pointer");
    *out = nullptr;
    sts_engine* e = new (std::nothrow) sts_engine();
    if (!e) return set_err(STS_EDEVICE, "out of host memory");
    int rc = e->eng.init(blob, blob_bytes, device);
    if (rc != STS_OK) { set_err(rc, e->eng.error()); delete e; return rc; }
    *out = e;
    return STS_OK;
}

void sts_destroy(sts_engine* e) { delete e; }

int sts_speaker_num(const sts_engine* e) {
    if (!e) return 1;
    return e->eng.model.spk_num == 0 ? 1 : e->eng.model.spk_num;   // SynthesizerTrn.cpp:79-89
}

int sts_get_info(const sts_engine* e, sts_model_info* info) {
    if (!e || !info) return set_err(STS_EINVAL, "null argument");
    const Model& m = e->eng.model;
    info->is_multi_speaker = m.is_ms; info->lang_type = m.lang; info->dur_pred_type = m.dur_type; info->dec_type = m.dec_type;
    info->vocab = m.vocab; info->hidden = m.hidden; info->inter_channels = m.inter;
    info->speaker_num = m.spk_num == 0 ? 1 : m.spk_num; info->gin_channels = m.gin;
    info->samples_per_frame = m.hop_total; info->sample_rate = 16000;
    info->blob_floats_consumed = m.consumed;
    return STS_OK;
}

int sts_run_batch(sts_engine* e, int32_t B, const int32_t* const* ids, const int32_t* n, const int32_t* sid,
                  const float* length_scale, int32_t* n_out, int64_t* total_out) {
    if (!e) return set_err(STS_EINVAL, "null engine");
    int rc = e->eng.run(B, ids, n, sid, length_scale);
    if (rc != STS_OK) return set_err(rc, e->eng.error());
    if (n_out) for (int b = 0; b < B; b++) n_out[b] = e->eng.n_samples[b];
    if (total_out) *total_out = e->eng.total_samples;
    return STS_OK;
}

int sts_copy_pcm_device(sts_engine* e, void* dst, int64_t cap) {
    if (!e || !dst) return set_err(STS_EINVAL, "null argument");
    if (cap < e->eng.total_samples || !e->eng.d_pcm) return set_err(STS_ESTATE, "destination too small or no run yet");
    // (d_pcm is the mapped pinned host buffer when the run's last kernel wrote the PCM there: Engine::pcm_in_host_)
    if (hipMemcpyAsync(dst, e->eng.pcm_in_host_ ? (const void*)e->eng.h_pcm : (const void*)e->eng.d_pcm, (size_t)e->eng.total_samples * 2,
                       e->eng.pcm_in_host_ ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, e->eng.stream) != hipSuccess ||
        hipStreamSynchronize(e->eng.stream) != hipSuccess)
        return set_err(STS_EDEVICE, "device copy failed");
    return STS_OK;
}

int sts_copy_pcm_host(sts_engine* e, int16_t* dst, int64_t cap) {
    if (!e || !dst) return set_err(STS_EINVAL, "null argument");
    if (cap < e->eng.total_samples || !e->eng.d_pcm) return set_err(STS_ESTATE, "destination too small or no run yet");
    if (e->eng.h_pcm) { memcpy(dst, e->eng.h_pcm, (size_t)e->eng.total_samples * 2); return STS_OK; }   // downloaded inside the run
    if (hipMemcpyAsync(dst, e->eng.d_pcm, (size_t)e->eng.total_samples * 2, hipMemcpyDeviceToHost, e->eng.stream) != hipSuccess ||
        hipStreamSynchronize(e->eng.stream) != hipSuccess)
        return set_err(STS_EDEVICE, "device-to-host copy failed");
    return STS_OK;
}

int sts_pcm_host_view(sts_engine* e, const int16_t** pcm, int64_t* count) {
    if (!e || !pcm || !count) return set_err(STS_EINVAL, "null argument");
    if (!e->eng.h_pcm || !e->eng.d_pcm) return set_err(STS_ESTATE, "no host copy of the PCM (sts_set_host_pcm(e, 1) before the run)");
    *pcm = e->eng.h_pcm; *count = e->eng.total_samples;
    return STS_OK;
}

int sts_infer_ids_batch(sts_engine* e, int32_t B, const int32_t* const* ids, const int32_t* n, const int32_t* sid,
                        const float* length_scale, int16_t** pcm_out, int32_t* n_out) {
    if (!pcm_out || !n_out) return set_err(STS_EINVAL, "null output");
    int64_t total = 0;
    if (!e) return set_err(STS_EINVAL, "null engine");
    const bool was = e->eng.host_pcm;
    e->eng.host_pcm = true;               // the PCM download rides at the end of the run: one stream sync for the whole call
    int rc = sts_run_batch(e, B, ids, n, sid, length_scale, n_out, &total);
    e->eng.host_pcm = was;
    if (rc != STS_OK) return rc;
    int16_t* all = (int16_t*)malloc((size_t)(total > 0 ? total : 1) * 2);
    if (!all) return set_err(STS_EDEVICE, "out of host memory");
    rc = sts_copy_pcm_host(e, all, total);
    if (rc != STS_OK) { free(all); return rc; }
    int64_t off = 0;
    for (int b = 0; b < B; b++) {
        if (B == 1) { pcm_out[0] = all; break; }   // single utterance: hand over the buffer itself
        pcm_out[b] = (int16_t*)malloc((size_t)(n_out[b] > 0 ? n_out[b] : 1) * 2);
        if (!pcm_out[b]) {
            for (int q = 0; q < b; q++) { free(pcm_out[q]); pcm_out[q] = nullptr; }
            free(all);
            return set_err(STS_EDEVICE, "out of host memory");
        }
        memcpy(pcm_out[b], all + off, (size_t)n_out[b] * 2);
        off += n_out[b];
    }
    if (B != 1) free(all);
    return STS_OK;
}

int sts_infer_ids(sts_engine* e, const int32_t* ids, int32_t n, int32_t sid, float length_scale, int16_t** pcm_out,
                  int32_t* n_out) {
    const int32_t* idp[1] = {ids};
    return sts_infer_ids_batch(e, 1, idp, &n, &sid, &length_scale, pcm_out, n_out);
}

int sts_infer_ids_stream(sts_engine* e, const int32_t* ids, int32_t n, int32_t sid, float length_scale, int32_t chunk_frames,
                         sts_chunk_cb cb, void* user, int32_t* n_total) {
    if (!e || !ids || !cb) return set_err(STS_EINVAL, "null argument");
    const int32_t* idp[1] = {ids};
    StreamSpec ss{chunk_frames, cb, user};
    const int rc = e->eng.run(1, idp, &n, &sid, &length_scale, &ss);
    if (rc != STS_OK) return set_err(rc, e->eng.error());
    if (n_total) *n_total = (int32_t)e->eng.total_samples;
    return STS_OK;
}

int sts_stream_halo_frames(const sts_engine* e) {
    if (!e) return set_err(STS_EINVAL, "null engine");
    return decoder_halo_frames(e->eng.model);
}

int sts_set_forced_durations(sts_engine* e, const int32_t* dur, int64_t count) {
    if (!e) return set_err(STS_EINVAL, "null engine");
    if (!dur || count <= 0) { e->eng.have_forced = false; return STS_OK; }
    e->eng.forced_dur.assign(dur, dur + count);
    e->eng.have_forced = true;
    return STS_OK;
}

int sts_set_record_taps(sts_engine* e, int enable) { if (!e) return set_err(STS_EINVAL, "null engine"); e->eng.record_taps = enable != 0; return STS_OK; }
int sts_set_conv_math(sts_engine* e, int mode) {
    if (!e) return set_err(STS_EINVAL, "null engine");
    if (mode < 0 || mode > 3) return set_err(STS_EINVAL, "conv math: 0 = split-bf16 (default), 1 = exact fp32, 2 = split-bf16 wherever eligible, 3 = two-term fp16");
    e->eng.conv_math = mode;
    e->eng.h2_consecutive = 0; e->eng.h2_disabled = false;
    return STS_OK;
}
int sts_set_conv_mode(sts_engine* e, int mode) { if (!e) return set_err(STS_EINVAL, "null engine"); e->eng.conv_mode = mode; return STS_OK; }
int sts_debug_set(sts_engine* e, int key, int value) {
    if (!e) return set_err(STS_EINVAL, "null engine");
    switch (key) {
        case STS_DBG_ATTN_BLOCK_MIN_WGS: if (value < 1) return set_err(STS_EINVAL, "threshold must be >= 1"); e->eng.attn_block_min_wgs = value; return STS_OK;
        case STS_DBG_FLOW_FUSED: e->eng.flow_fused = value != 0; return STS_OK;
        case STS_DBG_LAUNCH_AHEAD: if (value < 0 || value > 2) return set_err(STS_EINVAL, "launch_ahead must be 0, 1 or 2"); if ((value == 2) != (e->eng.launch_ahead == 2)) { e->eng.seen_tf_.clear(); e->eng.seen_order_.clear(); } e->eng.launch_ahead = value; return STS_OK;
        case STS_DBG_H2P: e->eng.h2p = value < 0 ? 0 : (value > 4 ? 4 : value); return STS_OK;
        case STS_DBG_H2P_TILE: e->eng.h2p_tile = value; return STS_OK;
        case STS_DBG_CHAIN_STREAMS: e->eng.chain_streams_dbg = value; return STS_OK;
        case STS_DBG_TAIL_FUSED: e->eng.tail_fused = value != 0; return STS_OK;
        case STS_DBG_UPS_ROWPH: e->eng.ups_rowph = value != 0; return STS_OK;
        case STS_DBG_MEMO_CLEAR: e->eng.seen_tf_.clear(); e->eng.seen_order_.clear(); return STS_OK;
        case STS_DBG_ATTN_REG: e->eng.attn_reg = value != 0; return STS_OK;
        case STS_DBG_DDS_TAIL: e->eng.dds_tail = value != 0; return STS_OK;
        case STS_DBG_PCM_DIRECT: e->eng.pcm_direct = value != 0; return STS_OK;
        default: return set_err(STS_EINVAL, "unknown debug key");
    }
}
int sts_set_host_pcm(sts_engine* e, int enable) { if (!e) return set_err(STS_EINVAL, "null engine"); e->eng.host_pcm = enable != 0; return STS_OK; }
int sts_set_profiling(sts_engine* e, int enable) { if (!e) return set_err(STS_EINVAL, "null engine"); e->eng.profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0); return STS_OK; }

int sts_abi_version(void) { return STS_ABI_VERSION; }
int sts_build_flags(void) {
#ifdef STS_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}
int sts_get_profile_ex(const sts_engine* e, void* p, int64_t size_bytes) {
    if (!e || !p || size_bytes <= 0) return set_err(STS_EINVAL, "null argument");
    // the struct only grows at its end: a client built against an older header hands in its own sizeof and gets that prefix
    const size_t n = (size_t)size_bytes < sizeof(sts_profile) ? (size_t)size_bytes : sizeof(sts_profile);
    memcpy(p, &e->eng.prof, n);
    return STS_OK;
}
int sts_get_profile(const sts_engine* e, sts_profile* p) { return sts_get_profile_ex(e, p, (int64_t)sizeof(sts_profile)); }

int sts_get_tap(sts_engine* e, const char* name, float** data, int32_t* channels, int64_t* length) {
    if (!e || !name || !data || !channels || !length) return set_err(STS_EINVAL, "null argument");
    auto it = e->eng.taps.find(name);
    if (it == e->eng.taps.end()) return set_err(STS_ESTATE, std::string("tap not recorded: ") + name);
    const Tap& t = it->second;
    float* p = (float*)malloc(t.data.size() * sizeof(float) + 4);
    if (!p) return set_err(STS_EDEVICE, "out of host memory");
    memcpy(p, t.data.data(), t.data.size() * sizeof(float));
    *data = p; *channels = t.channels; *length = t.length;
    return STS_OK;
}

int sts_get_durations(sts_engine* e, int32_t* dur, int64_t cap) {
    if (!e || !dur) return set_err(STS_EINVAL, "null argument");
    if ((int64_t)e->eng.durations_h.size() > cap) return set_err(STS_ESTATE, "destination too small");
    memcpy(dur, e->eng.durations_h.data(), e->eng.durations_h.size() * sizeof(int32_t));
    return STS_OK;
}

// ------------------------------------------------------------------------------------------------
// op-level entry for the parity tests: one conv on host arrays through the same kernels
int sts_debug_wino_pack(const float* w, int32_t Cout, int32_t k, int32_t Cin, float* out, int64_t out_floats) {
    if (!w || !out || Cout <= 0 || Cin <= 0 || k < 2) return set_err(STS_EINVAL, "bad arguments");
    int n3, n2; wino_split(k, &n3, &n2);
    const int Cin_pad = (Cin + 15) / 16 * 16, Cout_pad = (Cout + 31) / 32 * 32;
    if (out_floats < (int64_t)(n3 + n2) * 4 * Cin_pad * Cout_pad) return set_err(STS_EINVAL, "output buffer too small");
    wino_pack(w, (long)k * Cin, Cin, 1, Cout, k, Cin, Cin_pad, Cout_pad, out);
    return n3 + n2;
}

int sts_debug_conv1d(int device, const float* x, int32_t Cin, int32_t L, const float* w, const float* bias, int32_t Cout,
                     int32_t k, int32_t pad, int32_t dil, int32_t stride_t, int32_t depthwise, float in_slope, int32_t in_act,
                     int mode, float** y_out, int32_t* Lout_out) {
    return sts_debug_conv1d_bench(device, x, Cin, L, w, bias, Cout, k, pad, dil, stride_t, depthwise, in_slope, in_act, mode,
                                  y_out, Lout_out, 0, nullptr);
}

int sts_debug_conv1d_bench(int device, const float* x, int32_t Cin, int32_t L, const float* w, const float* bias, int32_t Cout,
                           int32_t k, int32_t pad, int32_t dil, int32_t stride_t, int32_t depthwise, float in_slope,
                           int32_t in_act, int mode, float** y_out, int32_t* Lout_out, int32_t iters, float* ms_out) {
    if (!x || !w || !y_out || !Lout_out || Cin <= 0 || Cout <= 0 || L <= 0 || k <= 0) return set_err(STS_EINVAL, "bad conv arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return set_err(STS_EDEVICE, "no HIP device visible (no CPU fallback)");
    if (hipSetDevice(device) != hipSuccess) return set_err(STS_EDEVICE, "hipSetDevice failed");
    const bool tr = stride_t > 0;
    const int Lout = tr ? (L - 1) * stride_t - 2 * pad + (k - 1) + 1 : L + 2 * pad - dil * (k - 1);
    if (Lout <= 0) return set_err(STS_EINVAL, "empty output");
    auto r_up = [](int v, int m) { return (v + m - 1) / m * m; };
    const int cin_eff = depthwise ? 1 : Cin;
    const int Cin_pad = r_up(cin_eff, 16), Cout_pad = r_up(Cout, 32);
    const int J = tr ? (k + stride_t - 1) / stride_t : 1;
    const size_t wn = depthwise ? (size_t)k * Cout_pad : (tr ? (size_t)stride_t * J : (size_t)k) * Cin_pad * Cout_pad;
    std::vector<float> wp(wn, 0.f), bp(Cout_pad, 0.f);
    if (depthwise) {
        for (int o = 0; o < Cout; o++) for (int t = 0; t < k; t++) wp[(size_t)t * Cout_pad + o] = w[(size_t)o * k + t];
    } else if (tr) {
        for (int ph = 0; ph < stride_t; ph++) for (int j = 0; j < J; j++) {
            int kk = ph + j * stride_t; if (kk >= k) continue;
            for (int ci = 0; ci < Cin; ci++) for (int o = 0; o < Cout; o++)
                wp[(((size_t)ph * J + j) * Cin_pad + ci) * Cout_pad + o] = w[((size_t)o * k + kk) * Cin + ci];
        }
    } else {
        for (int o = 0; o < Cout; o++) for (int t = 0; t < k; t++) for (int ci = 0; ci < Cin; ci++)
            wp[((size_t)t * Cin_pad + ci) * Cout_pad + o] = w[((size_t)o * k + t) * Cin + ci];
    }
    if (bias) memcpy(bp.data(), bias, sizeof(float) * Cout);
    // mode 12: the Winograd-domain kernel (weights transformed here the way load_model does it)
    std::vector<float> wu;
    int wn3 = 0, wn2 = 0;
    if (mode == 12 && !depthwise && !tr) {
        wino_split(k, &wn3, &wn2);
        wu.resize((size_t)(wn3 + wn2) * 4 * Cin_pad * Cout_pad);
        wino_pack(w, (long)k * Cin, Cin, 1, Cout, k, Cin, Cin_pad, Cout_pad, wu.data());
    }
    // mode + 100 (transposed convs of stride 2 / 4 / 8 through the split-operand kernels): the row-interleaved-phase packing (ConvArgs::rowph)
    const bool rowph = mode >= 100;
    if (rowph) {
        mode -= 100;
        if (!tr || depthwise || Cout != Cout_pad || (stride_t != 2 && stride_t != 4 && stride_t != 8)) return set_err(STS_EINVAL, "row-interleaved phases: transposed conv, stride 2 / 4 / 8, Cout % 32 == 0");
    }
    // modes 13 / 20..25 / 28..33: the split-bf16 kernel (conv_bf3.hip), automatic tile / tile code (mode - 20)
    const bool h2 = (mode == 50 || (mode >= 60 && mode < 85)) && !depthwise;       // the two-term fp16 form of the same kernel
    if (h2) mode = mode == 50 ? 13 : mode - 40;
    const bool bf3 = (mode == 13 || (mode >= 20 && mode < 45)) && !depthwise;
    std::vector<unsigned char> wb3;
    float h2_scale = 1.0f;
    if (rowph && !bf3) return set_err(STS_EINVAL, "row-interleaved phases: split-operand modes only");
    if (bf3 && rowph) {      // merged row rho = cout * stride + phase, one "phase" of J taps (as model.hip pack_bf3_rowph)
        const int rows = Cout_pad * stride_t;
        std::vector<float> wr((size_t)J * Cin_pad * rows, 0.f);
        for (int ph = 0; ph < stride_t; ph++) for (int j = 0; j < J; j++) for (int ci = 0; ci < Cin_pad; ci++) for (int o = 0; o < Cout_pad; o++)
            wr[((size_t)j * Cin_pad + ci) * rows + (size_t)o * stride_t + ph] = wp[(((size_t)ph * J + j) * Cin_pad + ci) * Cout_pad + o];
        wb3.resize(bf3_pack(wr.data(), 1, J, Cin_pad, rows, nullptr, false, h2 ? 1 : 0));
        bf3_pack(wr.data(), 1, J, Cin_pad, rows, wb3.data(), false, h2 ? 1 : 0, &h2_scale);
        std::vector<float> br((size_t)rows);
        for (int r = 0; r < rows; r++) br[r] = bp[r / stride_t];
        bp.swap(br);
    } else if (bf3) {
        wb3.resize(bf3_pack(wp.data(), tr ? stride_t : 1, tr ? J : k, Cin_pad, Cout_pad, nullptr, false, h2 ? 1 : 0));
        bf3_pack(wp.data(), tr ? stride_t : 1, tr ? J : k, Cin_pad, Cout_pad, wb3.data(), false, h2 ? 1 : 0, &h2_scale);
    }
    void* dwb3 = nullptr;
    float *dx = nullptr, *dw = nullptr, *db = nullptr, *dy = nullptr, *dwu = nullptr; int* dseg = nullptr;
    int seg[2] = {0, 1};
    bool ok = hipMalloc((void**)&dx, (size_t)Cin * L * 4) == hipSuccess && hipMalloc((void**)&dw, (wn + 1024) * 4) == hipSuccess &&
              hipMalloc((void**)&db, bp.size() * 4) == hipSuccess && hipMalloc((void**)&dy, (size_t)Cout * Lout * 4) == hipSuccess &&
              hipMalloc((void**)&dseg, 32) == hipSuccess;
    int rc = STS_OK;
    if (ok && !wu.empty()) ok = hipMalloc((void**)&dwu, (wu.size() + 1024) * 4) == hipSuccess;
    if (ok && bf3) ok = hipMalloc(&dwb3, wb3.size() + 4096) == hipSuccess;
    if (!ok) rc = set_err(STS_EDEVICE, "hipMalloc failed");
    if (rc == STS_OK) {
        (void)hipMemcpy(dx, x, (size_t)Cin * L * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dw, wp.data(), wn * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, bp.data(), bp.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemset(dseg, 0, 32);
        (void)hipMemcpy(dseg, seg, 8, hipMemcpyHostToDevice);
        (void)hipMemset(dy, 0, (size_t)Cout * Lout * 4);
        ConvArgs a;
        memset(&a, 0, sizeof(a));
        a.x = dx; a.x_ld = L; a.y = dy; a.y_ld = Lout; a.w = dw; a.bias = bias ? db : nullptr;
        a.Cin = cin_eff; a.Cout = Cout; a.Cin_pad = Cin_pad; a.Cout_pad = Cout_pad;
        a.depthwise = depthwise ? 1 : 0;
        if (tr) { a.ntap = J; a.tap_step = -1; a.tap_off = 0; a.out_stride = stride_t; a.out_off = -pad; a.transposed = 1; a.n_extra = J - 1; a.max_n = L + J - 1; }
        else { a.ntap = k; a.tap_step = dil; a.tap_off = -pad; a.out_stride = 1; a.out_off = 0; a.max_n = Lout; }
        a.in_act = in_act; a.in_slope = in_slope; a.epi = EPI_STORE;
        // segment lengths in base units: in = L, out = Lout -> two views over the same {off=0,len=1} table
        a.in_seg = SegView{dseg, dseg + 1, L, 0}; a.out_seg = SegView{dseg, dseg + 1, Lout, 0}; a.B = 1;
        if (dwu) { (void)hipMemcpy(dwu, wu.data(), wu.size() * 4, hipMemcpyHostToDevice); a.wu = dwu; a.wino_n3 = wn3; a.wino_n2 = wn2; }
        if (dwb3) { (void)hipMemcpy(dwb3, wb3.data(), wb3.size(), hipMemcpyHostToDevice); a.wb3 = dwb3; }
        if (h2) { a.math = 1; a.wscale = h2_scale; a.ovf = (unsigned*)dseg + 4; }     // (overflow word: behind the segment table)
        if (rowph) { a.rowph = stride_t; a.Cout = a.Cout_pad = Cout_pad * stride_t; }
        auto launch = [&]() {
            if (bf3) conv_bf3(a, nullptr, mode == 13 ? -1 : mode - 20);
            else if (mode == 12) conv_wino(a, nullptr);
            else if (mode != 1 && conv_mfma_eligible(a)) conv_mfma(a, nullptr, mode >= 2 ? mode - 2 : -1);
            else conv_generic(a, nullptr);
        };
        if (bf3 && !conv_bf3_eligible(a)) rc = set_err(STS_EINVAL, "shape not eligible for the split-bf16 kernel");
        else if (mode == 12 && !conv_wino_eligible(a)) rc = set_err(STS_EINVAL, "shape not eligible for the Winograd kernel");
        else if (!bf3 && mode >= 2 && mode != 12 && !conv_mfma_eligible(a)) rc = set_err(STS_EINVAL, "shape not eligible for the matrix-core kernel");
        else launch();
        if (rc == STS_OK && iters > 0 && ms_out) {   // steady-state timing of the same launch (HIP events, null stream)
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0, nullptr);
            for (int it = 0; it < iters; it++) launch();
            (void)hipEventRecord(e1, nullptr);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            *ms_out = ms / (float)iters;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
        if (rc == STS_OK && (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess)) rc = set_err(STS_EDEVICE, "conv kernel failed");
        if (rc == STS_OK) {
            float* y = (float*)malloc((size_t)Cout * Lout * 4);
            (void)hipMemcpy(y, dy, (size_t)Cout * Lout * 4, hipMemcpyDeviceToHost);
            *y_out = y; *Lout_out = Lout;
        }
    }
    (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(db); (void)hipFree(dy); (void)hipFree(dseg); if (dwu) (void)hipFree(dwu); if (dwb3) (void)hipFree(dwb3);
    return rc;
}

// One "same"-padded conv through the pre-split path (conv_h2p.hip): x fp32 [Cin][L] -> split_planes -> conv_h2p_group (`members` identical
// members in one grid; member 0 is returned) -> all three output forms decoded to fp32 [Cout][L] on the host.
int sts_debug_conv_h2p(int device, const float* x, int32_t Cin, int32_t L, const float* w, const float* bias, int32_t Cout, int32_t k, int32_t dil,
                       const float* res, float in_slope, float out_slope, int tile, int members, float* y_out, float* y16_out, float* yp_out,
                       int32_t iters, float* ms_out) {
    if (!x || !w || Cin <= 0 || Cout <= 0 || L <= 0 || k <= 0 || !(k & 1) || dil < 1 || members < 1 || members > kMaxGroup) return set_err(STS_EINVAL, "bad conv arguments");
    if (Cin % 16 || Cout % 32) return set_err(STS_EINVAL, "pre-split conv: Cin % 16 == 0 and Cout % 32 == 0");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return set_err(STS_EDEVICE, "no HIP device visible (no CPU fallback)");
    if (hipSetDevice(device) != hipSuccess) return set_err(STS_EDEVICE, "hipSetDevice failed");
    const int pad = dil * (k - 1) / 2;
    std::vector<float> wp((size_t)k * Cin * Cout, 0.f);
    for (int o = 0; o < Cout; o++) for (int t = 0; t < k; t++) for (int ci = 0; ci < Cin; ci++)
        wp[((size_t)t * Cin + ci) * Cout + o] = w[((size_t)o * k + t) * Cin + ci];
    std::vector<unsigned char> wb(bf3_pack(wp.data(), 1, k, Cin, Cout, nullptr, true, 1));
    float wscale = 1.0f;
    bf3_pack(wp.data(), 1, k, Cin, Cout, wb.data(), true, 1, &wscale);
    const size_t in_b = (size_t)Cin * L * 4, out_b = (size_t)Cout * L * 4;
    float *dx = nullptr, *db = nullptr, *dres = nullptr, *dres16 = nullptr; void *dxp = nullptr, *dwb = nullptr, *dtmp = nullptr; unsigned* dovf = nullptr;
    float* dy[kMaxGroup] = {}; float* dy16[kMaxGroup] = {}; void* dyp[kMaxGroup] = {};
    bool ok = hipMalloc((void**)&dx, in_b) == hipSuccess && hipMalloc(&dxp, in_b) == hipSuccess && hipMalloc(&dwb, wb.size() + 8192) == hipSuccess &&
              hipMalloc((void**)&db, (size_t)Cout * 4) == hipSuccess && hipMalloc((void**)&dovf, 64) == hipSuccess;
    if (ok && res) ok = hipMalloc((void**)&dres, out_b) == hipSuccess && hipMalloc((void**)&dres16, out_b) == hipSuccess && hipMalloc(&dtmp, out_b) == hipSuccess;
    for (int m = 0; m < members && ok; m++)
        ok = hipMalloc((void**)&dy[m], out_b) == hipSuccess && hipMalloc((void**)&dy16[m], out_b) == hipSuccess && hipMalloc(&dyp[m], out_b) == hipSuccess;
    int rc = ok ? STS_OK : set_err(STS_EDEVICE, "hipMalloc failed");
    if (rc == STS_OK) {
        (void)hipMemcpy(dx, x, in_b, hipMemcpyHostToDevice);
        (void)hipMemset(dwb, 0, wb.size() + 8192);
        (void)hipMemcpy(dwb, wb.data(), wb.size(), hipMemcpyHostToDevice);
        (void)hipMemset(db, 0, (size_t)Cout * 4);
        if (bias) (void)hipMemcpy(db, bias, (size_t)Cout * 4, hipMemcpyHostToDevice);
        (void)hipMemset(dovf, 0, 64);
        split_planes(dx, L, Cin, L, in_slope, dxp, nullptr, L, dovf, nullptr);
        if (res) {
            (void)hipMemcpy(dres, res, out_b, hipMemcpyHostToDevice);
            split_planes(dres, L, Cout, L, 1.0f, dtmp, dres16, L, nullptr, nullptr);
        }
        H2PGroup G;
        memset(&G, 0, sizeof(G));
        G.n = members; G.seg = SegView{nullptr, nullptr, 1, 0, 0, L}; G.B = 1; G.max_n = L; G.ovf = dovf;
        for (int m = 0; m < members; m++) {
            H2PArgs& a = G.g[m];
            a.xp = dxp; a.xp_ld = L; a.wb = dwb; a.wscale = wscale; a.bias = bias ? db : nullptr; a.res16 = dres16; a.res_ld = L;
            a.y = iters < 0 ? nullptr : dy[m]; a.y_ld = L; a.y16 = dy16[m]; a.y16_ld = L; a.yp = dyp[m]; a.yp_ld = L; a.yp_slope = out_slope;    // (iters < 0: timing with a layer's second conv's outputs only)
            a.Cin = Cin; a.Cout = Cout; a.ntap = k; a.tap_step = dil; a.tap_off = -pad;
        }
        if (!conv_h2p_group_eligible(G)) rc = set_err(STS_EINVAL, "shape not eligible for the pre-split kernel");
        else {
            conv_h2p_group(G, nullptr, tile);
            if (iters != 0 && ms_out) {
                if (iters < 0) iters = -iters;
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, nullptr);
                for (int it = 0; it < iters; it++) conv_h2p_group(G, nullptr, tile);
                (void)hipEventRecord(e1, nullptr);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                *ms_out = ms / (float)iters;
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            }
            if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = set_err(STS_EDEVICE, "conv kernel failed");
        }
        if (rc == STS_OK) {
            if (y_out) (void)hipMemcpy(y_out, dy[0], out_b, hipMemcpyDeviceToHost);
            std::vector<float> t16((size_t)Cout * L);
            std::vector<uint16_t> tp((size_t)Cout * L * 2);
            (void)hipMemcpy(t16.data(), dy16[0], out_b, hipMemcpyDeviceToHost);
            (void)hipMemcpy(tp.data(), dyp[0], out_b, hipMemcpyDeviceToHost);
            const size_t ps = (size_t)Cout * L;       // fp16 values per plane
            for (int c = 0; c < Cout / 16; c++) for (long t = 0; t < L; t++) for (int h = 0; h < 2; h++) for (int e = 0; e < 8; e++) {
                const int ch = 16 * c + 8 * (e >> 2) + 4 * h + (e & 3);
                const size_t u = ((size_t)c * L + t) * 16 + h * 8 + e;
                if (y16_out) y16_out[(size_t)ch * L + t] = t16[u];
                if (yp_out) {
                    _Float16 hi, lo;
                    memcpy(&hi, &tp[u], 2); memcpy(&lo, &tp[ps + u], 2);
                    yp_out[(size_t)ch * L + t] = (float)hi + (float)lo * (1.0f / 2048.0f);
                }
            }
        }
    }
    (void)hipFree(dx); (void)hipFree(dxp); (void)hipFree(dwb); (void)hipFree(db); (void)hipFree(dovf);
    if (dres) (void)hipFree(dres); if (dres16) (void)hipFree(dres16); if (dtmp) (void)hipFree(dtmp);
    for (int m = 0; m < kMaxGroup; m++) { if (dy[m]) (void)hipFree(dy[m]); if (dy16[m]) (void)hipFree(dy16[m]); if (dyp[m]) (void)hipFree(dyp[m]); }
    return rc;
}

// One "same"-padded conv through the Winograd-domain lab path (conv_h2w.hip): x fp32 [C][L] -> to_x16 -> conv_h2w_group (`members` identical
// members) -> member 0's fp32 [C][L] output and its x16 output (lrelu(out, out_slope)) decoded to fp32 [C][L].
int sts_debug_conv_h2w(int device, const float* x, int32_t C, int32_t L, const float* w, const float* bias, int32_t k, int32_t dil, const float* res,
                       float in_slope, float out_slope, int members, float* y_out, float* y16_out, int32_t iters, float* ms_out) {
    if (!x || !w || C <= 0 || L <= 0 || k <= 0 || !(k & 1) || dil < 1 || members < 1 || members > kMaxGroup || C % 128) return set_err(STS_EINVAL, "bad conv arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return set_err(STS_EDEVICE, "no HIP device visible (no CPU fallback)");
    if (hipSetDevice(device) != hipSuccess) return set_err(STS_EDEVICE, "hipSetDevice failed");
    std::vector<float> wp((size_t)k * C * C, 0.f);
    for (int o = 0; o < C; o++) for (int t = 0; t < k; t++) for (int ci = 0; ci < C; ci++) wp[((size_t)t * C + ci) * C + o] = w[((size_t)o * k + t) * C + ci];
    int n3, n2; wino_split(k, &n3, &n2);
    const int ntapw = 4 * n3 + 3 * n2;
    std::vector<float> wu((size_t)ntapw * C * C);
    h2w_transform_weights(wp.data(), k, C, C, wu.data());
    std::vector<unsigned char> wb(bf3_pack(wu.data(), 1, ntapw, C, C, nullptr, true, 1));
    float wscale = 1.0f;
    bf3_pack(wu.data(), 1, ntapw, C, C, wb.data(), true, 1, &wscale);
    const size_t tb = (size_t)C * L * 4;
    float *dx = nullptr, *dx16 = nullptr, *db = nullptr, *dres = nullptr, *dres16 = nullptr; void* dwb = nullptr; unsigned* dovf = nullptr;
    float* dy[kMaxGroup] = {}; float* dy16[kMaxGroup] = {};
    bool ok = hipMalloc((void**)&dx, tb) == hipSuccess && hipMalloc((void**)&dx16, tb) == hipSuccess && hipMalloc(&dwb, wb.size() + 8192) == hipSuccess &&
              hipMalloc((void**)&db, (size_t)C * 4) == hipSuccess && hipMalloc((void**)&dovf, 64) == hipSuccess;
    if (ok && res) ok = hipMalloc((void**)&dres, tb) == hipSuccess && hipMalloc((void**)&dres16, tb) == hipSuccess;
    for (int m = 0; m < members && ok; m++) ok = hipMalloc((void**)&dy[m], tb) == hipSuccess && hipMalloc((void**)&dy16[m], tb) == hipSuccess;
    int rc = ok ? STS_OK : set_err(STS_EDEVICE, "hipMalloc failed");
    if (rc == STS_OK) {
        (void)hipMemcpy(dx, x, tb, hipMemcpyHostToDevice);
        (void)hipMemset(dwb, 0, wb.size() + 8192);
        (void)hipMemcpy(dwb, wb.data(), wb.size(), hipMemcpyHostToDevice);
        (void)hipMemset(db, 0, (size_t)C * 4);
        if (bias) (void)hipMemcpy(db, bias, (size_t)C * 4, hipMemcpyHostToDevice);
        (void)hipMemset(dovf, 0, 64);
        to_x16(dx, L, C, L, dx16, L, nullptr);
        if (res) { (void)hipMemcpy(dres, res, tb, hipMemcpyHostToDevice); to_x16(dres, L, C, L, dres16, L, nullptr); }
        H2WGroup G;
        memset(&G, 0, sizeof(G));
        G.n = members; G.seg = SegView{nullptr, nullptr, 1, 0, 0, L}; G.B = 1; G.max_n = L; G.ovf = dovf;
        for (int m = 0; m < members; m++) {
            H2WArgs& a = G.g[m];
            a.x16 = dx16; a.x_ld = L; a.wu = dwb; a.wscale = wscale; a.bias = bias ? db : nullptr; a.res16 = dres16; a.res_ld = L;
            a.y16 = dy16[m]; a.y16_ld = L; a.y = iters < 0 ? nullptr : dy[m]; a.y_ld = L; a.in_slope = in_slope; a.out_slope = out_slope; a.C = C; a.k = k; a.dil = dil;
        }
        if (!conv_h2w_group_eligible(G)) rc = set_err(STS_EINVAL, "shape not eligible for the Winograd-domain kernel");
        else {
            conv_h2w_group(G, nullptr);
            if (iters != 0 && ms_out) {
                if (iters < 0) iters = -iters;
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, nullptr);
                for (int it = 0; it < iters; it++) conv_h2w_group(G, nullptr);
                (void)hipEventRecord(e1, nullptr);
                (void)hipEventSynchronize(e1);
                float ms = 0.f;
                (void)hipEventElapsedTime(&ms, e0, e1);
                *ms_out = ms / (float)iters;
                (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            }
            if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = set_err(STS_EDEVICE, "conv kernel failed");
        }
        if (rc == STS_OK) {
            if (y_out) (void)hipMemcpy(y_out, dy[0], tb, hipMemcpyDeviceToHost);
            if (y16_out) {
                std::vector<float> t16((size_t)C * L);
                (void)hipMemcpy(t16.data(), dy16[0], tb, hipMemcpyDeviceToHost);
                for (int c = 0; c < C / 16; c++) for (long t = 0; t < L; t++) for (int h = 0; h < 2; h++) for (int e = 0; e < 8; e++)
                    y16_out[(size_t)(16 * c + 8 * (e >> 2) + 4 * h + (e & 3)) * L + t] = t16[((size_t)c * L + t) * 16 + h * 8 + e];
            }
        }
    }
    (void)hipFree(dx); (void)hipFree(dx16); (void)hipFree(dwb); (void)hipFree(db); (void)hipFree(dovf);
    if (dres) (void)hipFree(dres); if (dres16) (void)hipFree(dres16);
    for (int m = 0; m < kMaxGroup; m++) { if (dy[m]) (void)hipFree(dy[m]); if (dy16[m]) (void)hipFree(dy16[m]); }
    return rc;
}

}  // extern "C"
