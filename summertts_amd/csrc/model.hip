// model.hip -- blob parser + weight repacker (see model.hpp).  Host code; compiled by hipcc for the
// HIP runtime calls only.
#include "model.hpp"
#include "kernels.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

namespace sts {

namespace {

struct Reader {
    const float* p; int64_t n; int64_t o = 0; bool ok = true;
    int geti() { if (o >= n) { ok = false; return 0; } return (int)p[o++]; }
    const float* take(int64_t cnt) {
        if (cnt < 0 || o + cnt > n) { ok = false; return p; }
        const float* r = p + o; o += cnt; return r;
    }
};

// /root/reference/src/nn_op/nn_conv1d.cpp:25-52
HConv parse_conv(Reader& r) {
    HConv c;
    c.out_ch = r.geti(); c.in_ch = r.geti(); c.k = r.geti(); c.pad = r.geti(); c.dil = r.geti(); c.has_bias = r.geti();
    if (c.out_ch <= 0 || c.in_ch <= 0 || c.k <= 0 || c.out_ch > (1 << 20) || c.in_ch > (1 << 20) || c.k > 4096) { r.ok = false; return c; }
    c.w = r.take((int64_t)c.in_ch * c.k * c.out_ch);
    if (c.has_bias == 1) c.b = r.take(c.out_ch);
    return c;
}
// /root/reference/src/nn_op/nn_conv1d_transposed.cpp:24-53
HConv parse_convT(Reader& r) {
    HConv c;
    c.out_ch = r.geti(); c.in_ch = r.geti(); c.k = r.geti(); c.pad = r.geti(); c.dil = r.geti(); c.has_bias = r.geti();
    c.stride = r.geti();
    if (c.out_ch <= 0 || c.in_ch <= 0 || c.k <= 0 || c.out_ch > (1 << 20) || c.in_ch > (1 << 20) || c.k > 4096) { r.ok = false; return c; }
    c.w = r.take((int64_t)c.in_ch * c.k * c.out_ch);
    if (c.has_bias == 1) c.b = r.take(c.out_ch);
    return c;
}
// /root/reference/src/nn_op/nn_layer_norm.cpp:17-31
HLn parse_ln(Reader& r) {
    HLn l; l.size = r.geti();
    if (l.size <= 0 || l.size > (1 << 20)) { r.ok = false; return l; }
    l.g = r.take(l.size); l.b = r.take(l.size);
    return l;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// packed weight store: one device allocation, one upload
// Packed-weight store.  Kernel-ready pointers are handed out WHILE the blob is parsed, so the device base address must exist before
// the final size is known.  Where the virtual-memory API works (probed; ADVICE r02: a 3x-blob hipMalloc kept ~370 MB per engine
// for a 112 MB model, and its heuristic bound could overflow on a valid model), only ADDRESS SPACE is reserved up front --
// generously: 8x the blob + 64 MB -- and physical memory of exactly the used size is mapped behind it after packing; otherwise a
// plain hipMalloc of the bound.  The host mirror is calloc'ed (untouched pages cost nothing).
struct HostArena {
    float* p = nullptr;
    float* data() const { return p; }
};
struct Store {
    HostArena host; float* dev = nullptr; size_t cap = 0, used = 0;
    bool vmm = false; size_t gran = 0, reserved = 0;
    hipMemGenericAllocationHandle_t handle{}; size_t mapped = 0;
    ~Store() { free(host.p); }
    static bool vmm_probe(int device, size_t* gran_out) {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        size_t g = 0;
        if (hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || g == 0) { (void)hipGetLastError(); return false; }
        void* va = nullptr; hipMemGenericAllocationHandle_t h{};
        bool ok = hipMemAddressReserve(&va, 2 * g, 0, nullptr, 0) == hipSuccess;
        if (ok) {
            ok = hipMemCreate(&h, g, &prop, 0) == hipSuccess;
            if (ok) {
                ok = hipMemMap(va, g, 0, h, 0) == hipSuccess;
                if (ok) {
                    hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
                    ok = hipMemSetAccess(va, g, &ad, 1) == hipSuccess && hipMemset(va, 0, 256) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
                    (void)hipMemUnmap(va, g);
                }
                (void)hipMemRelease(h);
            }
            (void)hipMemAddressFree(va, 2 * g);
        }
        if (!ok) (void)hipGetLastError();
        *gran_out = g;
        return ok;
    }
    bool init(size_t blob_floats) {
        int device = 0;
        (void)hipGetDevice(&device);
        vmm = vmm_probe(device, &gran);
        cap = vmm ? blob_floats * 9 + (16u << 20) : blob_floats * 7 + (8u << 20);
        host.p = (float*)calloc(cap, sizeof(float));
        if (!host.p) return false;
        if (vmm) {
            reserved = (cap * sizeof(float) + gran - 1) / gran * gran;
            void* va = nullptr;
            if (hipMemAddressReserve(&va, reserved, 0, nullptr, 0) == hipSuccess) { dev = (float*)va; return true; }
            (void)hipGetLastError();
            vmm = false; cap = blob_floats * 7 + (8u << 20);
        }
        return hipMalloc((void**)&dev, cap * sizeof(float)) == hipSuccess;
    }
    // physical memory behind the used part of the reservation (VMM) -- then the caller uploads `used` floats
    bool commit() {
        if (!vmm) return true;
        int device = 0;
        (void)hipGetDevice(&device);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = device;
        mapped = (used * sizeof(float) + 4096 + gran - 1) / gran * gran;       // + slack: unconditional prefetches read a little past the end
        if (hipMemCreate(&handle, mapped, &prop, 0) != hipSuccess) return false;
        if (hipMemMap(dev, mapped, 0, handle, 0) != hipSuccess) { (void)hipMemRelease(handle); return false; }
        hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(dev, mapped, &ad, 1) != hipSuccess) { (void)hipMemUnmap(dev, mapped); (void)hipMemRelease(handle); return false; }
        return hipMemset(dev, 0, mapped) == hipSuccess;
    }
    void release_device() {       // on a failed load
        if (!dev) return;
        if (vmm) { if (mapped) { (void)hipMemUnmap(dev, mapped); (void)hipMemRelease(handle); } (void)hipMemAddressFree(dev, reserved); }
        else (void)hipFree(dev);
        dev = nullptr;
    }
    // returns host pointer to fill and the matching device pointer
    float* alloc(size_t n, const float** dptr) {
        size_t a = (used + 63) / 64 * 64;
        if (a + n > cap) return nullptr;
        used = a + n;
        *dptr = dev + a;
        return host.data() + a;
    }
};

// packed row -> source row for the gate pairing: every 32-row MFMA tile holds 16 tanh rows followed by the 16 sigmoid
// rows of the SAME channels, so that one accumulator tile carries complete (t, s) pairs in each lane (rows r and r + 16
// of a 32x32 tile live in the same lane) and a 32-row tile is a legal unit of work for the gated conv
int gate_perm_row(int pr, int H) {
    int g = pr / 32, r = pr % 32;
    return r < 16 ? g * 16 + r : H + g * 16 + (r - 16);
}

bool pack_conv(Store& st, const HConv& h, const PackOpts& o, DConv& d) {
    const int rows = o.out_rows > 0 ? o.out_rows : h.out_ch;
    d = DConv();
    d.depthwise = o.depthwise ? 1 : 0;
    d.Cin = o.depthwise ? 1 : h.in_ch;
    d.Cout = rows;
    d.k = h.k; d.pad = h.pad; d.dil = h.dil;
    d.Cin_pad = round_up(d.Cin, 16);
    const bool perm = o.gate && (o.gate_H % 16 == 0);
    d.gate_perm = perm ? 1 : 0;
    d.H = o.gate_H;
    d.Cout_pad = round_up(rows, 32);
    d.macs_per_out = o.depthwise ? (double)rows * h.k : (double)d.Cin * rows * h.k;
    auto src_row = [&](int pr) -> int {
        if (pr >= rows) return -1;
        if (o.reverse_out) return rows - 1 - pr;
        if (perm) return gate_perm_row(pr, o.gate_H);
        if (o.gate_blocks && o.gate_H % 16 == 0) {
            int blk = pr / (2 * o.gate_H), within = pr % (2 * o.gate_H);
            return blk * 2 * o.gate_H + gate_perm_row(within, o.gate_H);
        }
        return pr;
    };
    const size_t wn = o.depthwise ? (size_t)h.k * d.Cout_pad : (size_t)h.k * d.Cin_pad * d.Cout_pad;
    float* w = st.alloc(wn, &d.w);
    if (!w) return false;
    for (int pr = 0; pr < d.Cout_pad; pr++) {
        const int sr = src_row(pr);
        if (sr < 0) continue;
        for (int t = 0; t < h.k; t++) {
            if (o.depthwise) { w[(size_t)t * d.Cout_pad + pr] = h.w[(size_t)sr * h.k + t]; continue; }
            for (int ci = 0; ci < d.Cin; ci++) {
                const int cs = o.reverse_in ? d.Cin - 1 - ci : ci;
                w[((size_t)t * d.Cin_pad + ci) * d.Cout_pad + pr] = h.w[((size_t)sr * h.k + t) * h.in_ch + cs];
            }
        }
    }
    if (h.has_bias == 1 && h.b) {
        float* b = st.alloc(d.Cout_pad, &d.bias);
        if (!b) return false;
        for (int pr = 0; pr < d.Cout_pad; pr++) { const int sr = src_row(pr); if (sr >= 0) b[pr] = h.b[sr]; }
    }
    return true;
}

// Winograd-domain copy U = G g of a plain conv's weights (kernels.hpp: wino_pack) for resblock_wino_kernel;
// only the ResBlock convs of the <= 128-channel decoder stages get one (the kernel's LDS tile bounds C)
bool pack_wino(Store& st, const HConv& h, DConv& d) {
    if (d.depthwise || d.transposed || h.k < 2 || h.k > 15 || d.Cin != d.Cout || d.Cin != d.Cin_pad || d.Cout != d.Cout_pad) return true;
    if (d.Cin != 32 && d.Cin != 64 && d.Cin != 128) return true;
    if (h.dil < 1 || h.dil > 6 || 30 % h.dil != 0 || (h.k - 1) * h.dil > 64) return true;
    int n3, n2; wino_split(h.k, &n3, &n2);
    const size_t n = (size_t)(n3 + n2) * 4 * d.Cin_pad * d.Cout_pad;
    float* u = st.alloc(n, &d.wu);
    if (!u) { d.wu = nullptr; return false; }
    wino_pack(h.w, (long)h.k * h.in_ch, h.in_ch, 1, d.Cout, h.k, d.Cin, d.Cin_pad, d.Cout_pad, u);
    return true;
}

// split-bf16 copy of an already packed conv (kernels.hpp: bf3_pack) for the bf16-matrix-core kernels of conv_bf3.hip
bool pack_bf3(Store& st, DConv& d, bool perm_k = false) {
    if (d.depthwise || d.Cin != d.Cin_pad || d.Cin < 32 || !d.w) return true;
    if (perm_k && (d.transposed || d.Cin != d.Cout || d.Cout != d.Cout_pad || d.Cin > 128)) return true;
    const int nphase = d.transposed ? d.stride : 1, ntap = d.transposed ? d.J : d.k;
    const size_t bytes = bf3_pack(nullptr, nphase, ntap, d.Cin_pad, d.Cout_pad, nullptr);
    const float* dptr = nullptr;
    float* p = st.alloc((bytes + 3) / 4 + 1024, &dptr);     // + slack: unconditional prefetches run one step past the end
    if (!p) return false;
    const float* wh = st.host.data() + (d.w - st.dev);
    bf3_pack(wh, nphase, ntap, d.Cin_pad, d.Cout_pad, p, perm_k);
    if (perm_k) d.wb3p = dptr; else d.wb3 = dptr;
    // the two-term fp16 form of the same copy (conv math "f16x2"; same layout and size)
    const float* hptr = nullptr;
    const size_t hbytes = bf3_pack(nullptr, nphase, ntap, d.Cin_pad, d.Cout_pad, nullptr, false, 1);
    float* q = st.alloc((hbytes + 3) / 4 + 1024, &hptr);
    if (!q) return false;
    bf3_pack(wh, nphase, ntap, d.Cin_pad, d.Cout_pad, q, perm_k, 1, &d.h2_scale);
    if (perm_k) d.wh2p = hptr; else d.wh2 = hptr;
    return true;
}

// row-interleaved-phase copies of a polyphase transposed conv (ConvArgs::rowph): merged row rho = cout * stride + phase, one "phase" of J taps
bool pack_bf3_rowph(Store& st, DConv& d) {
    if (!d.transposed || !d.w || !d.wb3 || d.Cin != d.Cin_pad || d.Cout != d.Cout_pad || (d.stride != 2 && d.stride != 4 && d.stride != 8)) return true;
    const int s = d.stride, rows = d.Cout_pad * s;
    const float* wh = st.host.data() + (d.w - st.dev);        // [phase][J][Cin_pad][Cout_pad]
    std::vector<float> wr((size_t)d.J * d.Cin_pad * rows);
    for (int ph = 0; ph < s; ph++)
        for (int j = 0; j < d.J; j++)
            for (int ci = 0; ci < d.Cin_pad; ci++) {
                const float* src = wh + (((size_t)ph * d.J + j) * d.Cin_pad + ci) * d.Cout_pad;
                float* dst = wr.data() + ((size_t)j * d.Cin_pad + ci) * rows + ph;
                for (int co = 0; co < d.Cout_pad; co++) dst[(size_t)co * s] = src[co];
            }
    const size_t bytes = bf3_pack(nullptr, 1, d.J, d.Cin_pad, rows, nullptr);
    const float* dptr = nullptr;
    float* p = st.alloc((bytes + 3) / 4 + 1024, &dptr);
    if (!p) return false;
    bf3_pack(wr.data(), 1, d.J, d.Cin_pad, rows, p, false);
    const size_t hbytes = bf3_pack(nullptr, 1, d.J, d.Cin_pad, rows, nullptr, false, 1);
    const float* hptr = nullptr;
    float* q = st.alloc((hbytes + 3) / 4 + 1024, &hptr);
    if (!q) return false;
    float sc = 1.0f;
    bf3_pack(wr.data(), 1, d.J, d.Cin_pad, rows, q, false, 1, &sc);
    if (sc != d.h2_scale) return true;                        // (same weights, same largest magnitude: cannot differ; keep the phase-major copy if it does)
    const float* bptr = nullptr;
    float* b = st.alloc((size_t)rows, &bptr);
    if (!b) return false;
    if (d.bias) { const float* bh = st.host.data() + (d.bias - st.dev); for (int r = 0; r < rows; r++) b[r] = bh[r / s]; }
    d.wb3r = dptr; d.wh2r = hptr; d.bias_r = bptr;
    return true;
}

// perm_k two-term fp16 copy alone (conv_h2p.hip: the pre-split path of the wide ResBlock stages consumes activations in the k order the
// accumulator layout produces them in; both convs of a layer need it, at every width the path takes)
bool pack_h2p(Store& st, DConv& d) {
    if (d.wh2p || d.depthwise || d.transposed || !d.w || d.Cin != d.Cin_pad || d.Cout != d.Cout_pad || d.Cin != d.Cout || d.Cin % 128 != 0) return true;
    const float* hptr = nullptr;
    const size_t hbytes = bf3_pack(nullptr, 1, d.k, d.Cin_pad, d.Cout_pad, nullptr, true, 1);
    float* q = st.alloc((hbytes + 3) / 4 + 2048, &hptr);
    if (!q) return false;
    const float* wh = st.host.data() + (d.w - st.dev);
    float sc = 1.0f;
    bf3_pack(wh, 1, d.k, d.Cin_pad, d.Cout_pad, q, true, 1, &sc);
    d.wh2p = hptr; d.h2_scale = sc;
    return true;
}

// Operands of wn_flow.hip for one coupling (see DFlowFused).  post is linear, so m = post(sum_l skip_l) = sum_l (W_post W_skip_l) acts_l + const:
// the composite matrices are formed in double here, negated (the kernel accumulates -m and ADDS it to x1), and appended to each layer's
// res rows.  All sources are the already packed fp32 copies, so the coupling's channel reversal (folded into pre / post) is in.
bool pack_flow_fused(Store& st, DCoupling& c) {
    DFlowFused& f = c.ff;
    const DWn& w = c.wn;
    const int H = w.H, half = c.post.Cout, n = w.n;
    if (n < 1 || c.pre.Cin != half || c.pre.Cout != H || c.post.Cin != H || !c.pre.wh2 || c.pre.Cin_pad != half) return true;
    if (!flow_layer_shape_ok(H, half, w.in[0].k, w.in[0].dil, n)) return true;
    for (int l = 0; l < n; l++) {
        const DConv &gi = w.in[l], &rs = w.rs[l];
        if (!gi.gate_perm || gi.Cin != H || gi.Cout != 2 * H || gi.Cin_pad != H || gi.Cout_pad != 2 * H || !gi.wh2 || gi.k != w.in[0].k || gi.dil != 1 || gi.pad != (gi.k - 1) / 2) return true;
        if (rs.k != 1 || rs.Cin != H || rs.Cin_pad != H || rs.Cout != (l + 1 < n ? 2 * H : H)) return true;
    }
    f.G = flow_layer_groups(H); f.Cg = 32;
    auto host = [&](const float* dptr) { return st.host.data() + (dptr - st.dev); };
    {   // gate conv of layer 0 in the k order of the parked `pre` output
        const DConv& gi = w.in[0];
        const size_t bytes = bf3_pack(nullptr, 1, gi.k, gi.Cin_pad, gi.Cout_pad, nullptr, true, 1);
        const float* dptr = nullptr;
        float* p = st.alloc((bytes + 3) / 4 + 1024, &dptr);
        if (!p) return false;
        bf3_pack(host(gi.w), 1, gi.k, gi.Cin_pad, gi.Cout_pad, p, true, 1, &f.gate0_scale);
        f.gate0 = dptr;
    }
    const float* wpost = host(c.post.w);                       // [1][H][post.Cout_pad]
    const int pcp = c.post.Cout_pad;
    std::vector<double> mb(half, 0.0);
    for (int cch = 0; cch < half; cch++) mb[cch] = c.post.bias ? (double)host(c.post.bias)[cch] : 0.0;
    f.wc.assign(n, nullptr); f.sc.assign(n, 1.f); f.rows_c.assign(n, 0); f.rows_res.assign(n, 0); f.b_res.assign(n, nullptr);
    for (int l = 0; l < n; l++) {
        const DConv& rs = w.rs[l];
        const float* wrs = host(rs.w);                         // [1][H][rs.Cout_pad]: rows 0..H-1 res (layers before the last), then skip
        const float* brs = rs.bias ? host(rs.bias) : nullptr;
        const int rows_res = l + 1 < n ? H : 0, skip0 = l + 1 < n ? H : 0;
        const int rows_c = round_up(rows_res + half, 32);
        std::vector<float> m((size_t)H * rows_c, 0.f);         // [ci][row]
        for (int ci = 0; ci < H; ci++) {
            for (int r = 0; r < rows_res; r++) m[(size_t)ci * rows_c + r] = wrs[(size_t)ci * rs.Cout_pad + r];
            for (int cch = 0; cch < half; cch++) {
                double acc = 0.0;
                for (int j = 0; j < H; j++) acc += (double)wpost[(size_t)j * pcp + cch] * (double)wrs[(size_t)ci * rs.Cout_pad + skip0 + j];
                m[(size_t)ci * rows_c + rows_res + cch] = (float)(-acc);
            }
        }
        if (brs) for (int cch = 0; cch < half; cch++) { double acc = 0.0; for (int j = 0; j < H; j++) acc += (double)wpost[(size_t)j * pcp + cch] * (double)brs[skip0 + j]; mb[cch] += acc; }
        const size_t bytes = bf3_pack(nullptr, 1, 1, H, rows_c, nullptr, true, 1);
        const float* dptr = nullptr;
        float* p = st.alloc((bytes + 3) / 4 + 1024, &dptr);
        if (!p) return false;
        bf3_pack(m.data(), 1, 1, H, rows_c, p, true, 1, &f.sc[l]);
        f.wc[l] = dptr; f.rows_c[l] = rows_c; f.rows_res[l] = rows_res;
        if (rows_res && brs) {
            float* b = st.alloc(round_up(H, 4) + 4, &f.b_res[l]);
            if (!b) return false;
            memcpy(b, brs, sizeof(float) * H);
        }
    }
    {
        float* b = st.alloc(round_up(half, 4) + 4, &f.b_m);
        if (!b) return false;
        for (int cch = 0; cch < half; cch++) b[cch] = (float)(-mb[cch]);
    }
    f.ok = true;
    return true;
}

// extra copy of a square 1x1 conv in the fused column-block kernel's operand order (col_layer.hip)
bool pack_col(Store& st, const HConv& h, DConv& d, int out_rows = -1) {
    const int rows = out_rows > 0 ? out_rows : h.out_ch;
    if (h.k != 1 || d.depthwise || !col_layer_width_ok(h.in_ch) || rows < h.in_ch) return true;
    const int npass = (rows + h.in_ch - 1) / h.in_ch;
    if (npass > 4) return true;
    float* p = st.alloc((size_t)npass * h.in_ch * h.in_ch, &d.wc);
    if (!p) { d.wc = nullptr; return false; }
    col_proj_pack(h.w, h.in_ch, rows, p);
    d.bias_rows = d.bias;       // plain convs are packed in source row order
    return true;
}

// polyphase repack of a transposed conv: phase p holds taps k = p + j*stride
bool pack_convT(Store& st, const HConv& h, int stride, int pad, DConv& d) {
    d = DConv();
    d.transposed = 1; d.stride = stride; d.pad = pad; d.k = h.k; d.dil = 1;
    d.J = (h.k + stride - 1) / stride;
    d.Cin = h.in_ch; d.Cout = h.out_ch;
    d.Cin_pad = round_up(d.Cin, 16); d.Cout_pad = round_up(d.Cout, 32);
    d.macs_per_out = (double)d.Cin * d.Cout * h.k;   // per INPUT position
    float* w = st.alloc((size_t)stride * d.J * d.Cin_pad * d.Cout_pad, &d.w);
    if (!w) return false;
    for (int ph = 0; ph < stride; ph++)
        for (int j = 0; j < d.J; j++) {
            const int kk = ph + j * stride;
            if (kk >= h.k) continue;
            for (int ci = 0; ci < d.Cin; ci++)
                for (int co = 0; co < d.Cout; co++)
                    w[(((size_t)ph * d.J + j) * d.Cin_pad + ci) * d.Cout_pad + co] = h.w[((size_t)co * h.k + kk) * h.in_ch + ci];
        }
    if (h.has_bias == 1 && h.b) {
        float* b = st.alloc(d.Cout_pad, &d.bias);
        if (!b) return false;
        memcpy(b, h.b, sizeof(float) * d.Cout);
    }
    return true;
}

bool pack_ln(Store& st, const HLn& h, DLn& d) {
    d.C = h.size;
    float* g = st.alloc(h.size, &d.g); if (!g) return false;
    memcpy(g, h.g, sizeof(float) * h.size);
    float* b = st.alloc(h.size, &d.b); if (!b) return false;
    memcpy(b, h.b, sizeof(float) * h.size);
    return true;
}

bool copy_raw(Store& st, const float* src, size_t n, const float** dptr) {
    float* p = st.alloc(n, dptr); if (!p) return false;
    memcpy(p, src, sizeof(float) * n);
    return true;
}

// /root/reference/src/modules/DDSConv.cpp:29-59 (header pad/dil of the depthwise convs are overridden)
bool parse_pack_dds(Reader& r, Store& st, DDds& d) {
    d.n = r.geti(); const int k = r.geti();
    if (!r.ok || d.n < 0 || d.n > 64) return false;
    int dil = 1;
    d.sep.resize(d.n); d.pw.resize(d.n); d.n1.resize(d.n); d.n2.resize(d.n);
    for (int i = 0; i < d.n; i++) {
        HConv h = parse_conv(r); if (!r.ok) return false;
        h.pad = (int)floor((float)(k * dil - dil) / 2.0); h.dil = dil;
        PackOpts o; o.depthwise = true;
        if (!pack_conv(st, h, o, d.sep[i])) return false;
        dil *= k;
    }
    for (int i = 0; i < d.n; i++) { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), d.pw[i]) || !pack_col(st, h, d.pw[i])) return false; }
    for (int i = 0; i < d.n; i++) { HLn h = parse_ln(r); if (!r.ok || !pack_ln(st, h, d.n1[i])) return false; }
    for (int i = 0; i < d.n; i++) { HLn h = parse_ln(r); if (!r.ok || !pack_ln(st, h, d.n2[i])) return false; }
    return true;
}
void skip_dds(Reader& r) {
    int n = r.geti(); r.geti();
    for (int i = 0; i < 2 * n && r.ok; i++) parse_conv(r);
    for (int i = 0; i < 2 * n && r.ok; i++) parse_ln(r);
}

// /root/reference/src/modules/ConvFlow.cpp:37-41
bool parse_pack_convflow(Reader& r, Store& st, DConvFlow& c) {
    HConv pre = parse_conv(r); if (!r.ok || !pack_conv(st, pre, PackOpts(), c.pre)) return false;
    if (!parse_pack_dds(r, st, c.dds)) return false;
    HConv proj = parse_conv(r); if (!r.ok || !pack_conv(st, proj, PackOpts(), c.proj)) return false;
    c.filter = pre.out_ch;
    return proj.out_ch == 29;
}
void skip_convflow(Reader& r) { parse_conv(r); skip_dds(r); parse_conv(r); }

// /root/reference/src/modules/pqmf.cpp:39-102: cosine-modulated synthesis bank, float arithmetic as the reference
void build_pqmf(float* fir /*[63][4]*/) {
    static const float proto[63] = {
        8.36595339e-06f, 2.68017852e-05f, 5.05711124e-05f, 6.13482515e-05f, 2.75281598e-05f, -8.62839965e-05f,
        -2.99268467e-04f, -5.88389492e-04f, -8.67064627e-04f, -9.82905838e-04f, -7.47200209e-04f, 8.04087656e-19f,
        1.30001234e-03f, 2.98798828e-03f, 4.64603942e-03f, 5.63488600e-03f, 5.22586317e-03f, 2.82493436e-03f,
        -1.75650987e-03f, -8.06073440e-03f, -1.48622207e-02f, -2.02404650e-02f, -2.18780344e-02f, -1.75512321e-02f,
        -5.71474631e-03f, 1.39652689e-02f, 4.02848855e-02f, 7.05021626e-02f, 1.00706377e-01f, 1.26503321e-01f,
        1.43873012e-01f, 1.50000000e-01f, 1.43873012e-01f, 1.26503321e-01f, 1.00706377e-01f, 7.05021626e-02f,
        4.02848855e-02f, 1.39652689e-02f, -5.71474631e-03f, -1.75512321e-02f, -2.18780344e-02f, -2.02404650e-02f,
        -1.48622207e-02f, -8.06073440e-03f, -1.75650987e-03f, 2.82493436e-03f, 5.22586317e-03f, 5.63488600e-03f,
        4.64603942e-03f, 2.98798828e-03f, 1.30001234e-03f, 8.04087656e-19f, -7.47200209e-04f, -9.82905838e-04f,
        -8.67064627e-04f, -5.88389492e-04f, -2.99268467e-04f, -8.62839965e-05f, 2.75281598e-05f, 6.13482515e-05f,
        5.05711124e-05f, 2.68017852e-05f, 8.36595339e-06f};
    for (int b = 0; b < 4; b++)
        for (int tau = 0; tau < 63; tau++) {
            const float t1 = ((float)tau - (((float)62 - 1) / 2)) * (float)(M_PI / (2 * 4));
            const float ph = (float)((b & 1 ? -1.0f : 1.0f) * (float)(M_PI / 4));
            fir[tau * 4 + b] = proto[tau] * 2.0f * cosf(t1 * (float)(2 * b + 1) - ph);
        }
}

}  // namespace

#define FAIL(msg) do { m.error = (msg); return false; } while (0)

bool load_model(const float* blob, int64_t nfloats, Model& m) {
    if (!blob || nfloats < 8) FAIL("blob too small");
    Reader r{blob, nfloats};
    Store st;
    // repacking pads channel counts to 16/32 and transposed convs to whole phases: bound generously
    if (!st.init((size_t)nfloats)) FAIL("allocation of the weight store failed");
    m.dev_weights = st.dev; m.dev_vmm = st.vmm; m.dev_reserved = st.reserved;

    // header: /root/reference/src/models/SynthesizerTrn.cpp:103-106
    m.is_ms = r.geti(); m.lang = r.geti(); m.dur_type = r.geti(); m.dec_type = r.geti();
    if (m.dec_type < 0 || m.dec_type > 3) FAIL("SynthesizerTrn: Unknown decoder");
    if (m.dur_type < 0 || m.dur_type > 1) FAIL("SynthesizerTrn: Unknown duration predicator");

    // ---- TextEncoder: /root/reference/src/models/TextEncoder.cpp:32-44
    m.hidden = r.geti(); m.vocab = r.geti(); m.emb_size = r.geti();
    if (!r.ok || m.vocab <= 0 || m.emb_size <= 0 || m.hidden <= 0) FAIL("bad text-encoder header");
    { const float* e = r.take((int64_t)m.vocab * m.emb_size); if (!r.ok || !copy_raw(st, e, (size_t)m.vocab * m.emb_size, &m.emb)) FAIL("embedding"); }
    // attention_encoder: grouped by kind (/root/reference/src/modules/attention_encoder.cpp:30-55)
    m.n_layers = r.geti();
    if (!r.ok || m.n_layers < 0 || m.n_layers > 256) FAIL("bad layer count");
    m.mha.resize(m.n_layers); m.ln1.resize(m.n_layers); m.ln2.resize(m.n_layers); m.ffn.resize(m.n_layers);
    for (int i = 0; i < m.n_layers; i++) {   // /root/reference/src/modules/multi_head_attention.cpp:40-90
        DMha& a = m.mha[i];
        a.ch = r.geti(); r.geti(); const int nh = r.geti(); a.win = r.geti();
        if (!r.ok || nh != 2 || a.ch <= 0 || (a.ch & 1)) FAIL("multi_head_attention: the reference hard-codes 2 heads");
        a.kc = a.ch / 2;
        if (a.win != 0) {
            int px = r.geti(), py = r.geti();
            if (px != 2 * a.win + 1 || py != a.kc) FAIL("relative embedding shape");
            a.px = px;
            const float* rk = r.take((int64_t)px * py);
            px = r.geti(); py = r.geti();
            if (px != a.px || py != a.kc) FAIL("relative embedding shape");
            const float* rv = r.take((int64_t)px * py);
            if (!r.ok || !copy_raw(st, rk, (size_t)px * py, &a.relk) || !copy_raw(st, rv, (size_t)px * py, &a.relv)) FAIL("relative embeddings");
        }
        HConv q = parse_conv(r), k = parse_conv(r), v = parse_conv(r), o = parse_conv(r);
        if (!r.ok || q.k != 1 || k.k != 1 || v.k != 1 || q.in_ch != a.ch || q.out_ch != a.ch || k.out_ch != a.ch || v.out_ch != a.ch || o.k != 1)
            FAIL("attention projections must be 1x1");
        // fuse q, k, v into ONE conv with 3*ch output rows
        std::vector<float> wcat((size_t)3 * a.ch * a.ch), bcat((size_t)3 * a.ch, 0.f);
        const HConv* qkv[3] = {&q, &k, &v};
        bool anyb = false;
        for (int t = 0; t < 3; t++) {
            memcpy(wcat.data() + (size_t)t * a.ch * a.ch, qkv[t]->w, sizeof(float) * (size_t)a.ch * a.ch);
            if (qkv[t]->has_bias == 1) { memcpy(bcat.data() + (size_t)t * a.ch, qkv[t]->b, sizeof(float) * a.ch); anyb = true; }
        }
        HConv cat; cat.out_ch = 3 * a.ch; cat.in_ch = a.ch; cat.k = 1; cat.pad = 0; cat.dil = 1; cat.has_bias = anyb ? 1 : 0;
        cat.w = wcat.data(); cat.b = bcat.data();
        if (!pack_conv(st, cat, PackOpts(), a.qkv) || !pack_bf3(st, a.qkv) || !pack_conv(st, o, PackOpts(), a.o) || !pack_bf3(st, a.o) || !pack_col(st, o, a.o) || !pack_col(st, cat, a.qkv))
            FAIL("weight store overflow");
    }
    for (int i = 0; i < m.n_layers; i++) { HLn h = parse_ln(r); if (!r.ok || !pack_ln(st, h, m.ln1[i])) FAIL("ln1"); }
    for (int i = 0; i < m.n_layers; i++) {   // /root/reference/src/modules/ffn.cpp:27-30
        m.ffn[i].ksize = r.geti();
        HConv c1 = parse_conv(r), c2 = parse_conv(r);
        if (!r.ok || c1.k != m.ffn[i].ksize || c2.k != m.ffn[i].ksize || c1.pad != 0 || c2.pad != 0) FAIL("ffn convs");
        if (!pack_conv(st, c1, PackOpts(), m.ffn[i].c1) || !pack_bf3(st, m.ffn[i].c1) || !pack_conv(st, c2, PackOpts(), m.ffn[i].c2) || !pack_bf3(st, m.ffn[i].c2)) FAIL("ffn pack");
    }
    for (int i = 0; i < m.n_layers; i++) { HLn h = parse_ln(r); if (!r.ok || !pack_ln(st, h, m.ln2[i])) FAIL("ln2"); }
    {
        HConv p = parse_conv(r);
        if (!r.ok || (p.out_ch & 1) || p.k != 1) FAIL("encoder proj");
        m.inter = p.out_ch / 2;
        // `logs` (second half) is dead at noiseScale == 0 (/root/reference/src/models/SynthesizerTrn.cpp:357,383): keep m only
        PackOpts o; o.out_rows = m.inter;
        if (!pack_conv(st, p, o, m.proj) || !pack_col(st, p, m.proj, m.inter)) FAIL("proj pack");
    }

    // ---- decoder: Generator_hifigan.cpp:44-101, Generator_MS.cpp:51-127, Generator_Istft.cpp:49-113, Generator_MBB.cpp:51-106
    if (m.dec_type != 0) { m.subbands = r.geti(); m.nfft = r.geti(); m.hop = r.geti(); }
    m.n_up = r.geti();
    if (!r.ok || m.n_up < 0 || m.n_up > 32) FAIL("bad upsample count");
    m.up_rate.resize(m.n_up);
    for (int i = 0; i < m.n_up; i++) m.up_rate[i] = r.geti();
    m.up_init = r.geti();
    { int nk = r.geti(); if (!r.ok || nk < m.n_up || nk > 32) FAIL("bad upsample kernel list"); m.up_k.resize(nk); for (int i = 0; i < nk; i++) m.up_k[i] = r.geti(); }
    m.n_resk = r.geti();
    if (!r.ok || m.n_resk <= 0 || m.n_resk > 32) FAIL("bad resblock kernel list");
    for (int i = 0; i < m.n_resk; i++) r.geti();
    { int nd = r.geti(); if (!r.ok || nd < 0 || nd > 64) FAIL("bad dilation table"); r.take(3 * (int64_t)nd); }
    { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.conv_pre) || !pack_bf3(st, m.conv_pre)) FAIL("conv_pre"); }
    m.ups.resize(m.n_up);
    m.hop_total = 1;
    for (int i = 0; i < m.n_up; i++) {
        HConv h = parse_convT(r);
        const int u = m.up_rate[i], k = m.up_k[i];
        if (!r.ok || u <= 0 || k != h.k) FAIL("upsampler");
        const int pad = (int)floor((float)(k - u) / (2.0));   // header stride/padding overridden (Generator_hifigan.cpp:76-82)
        if (k - 2 * pad != u) FAIL("upsampler with (k - stride) odd is not length-preserving");
        if (!pack_convT(st, h, u, pad, m.ups[i]) || !pack_bf3(st, m.ups[i]) || !pack_bf3_rowph(st, m.ups[i])) FAIL("upsampler pack");
        m.hop_total *= u;
    }
    m.rb.resize((size_t)m.n_up * m.n_resk);
    for (auto& rb : m.rb) {   // /root/reference/src/modules/ResBlock1.cpp:27-38
        const int n = r.geti();
        if (!r.ok || n <= 0 || n > 32) FAIL("resblock");
        rb.c1.resize(n); rb.c2.resize(n);
        for (int i = 0; i < n; i++) { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), rb.c1[i]) || !pack_wino(st, h, rb.c1[i]) || !pack_bf3(st, rb.c1[i]) || !pack_h2p(st, rb.c1[i])) FAIL("resblock convs1"); }
        for (int i = 0; i < n; i++) { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), rb.c2[i]) || !pack_wino(st, h, rb.c2[i]) || !pack_bf3(st, rb.c2[i]) || !pack_bf3(st, rb.c2[i], true) || !pack_h2p(st, rb.c2[i])) FAIL("resblock convs2"); }
    }
    { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.conv_post)) FAIL("conv_post");
      // the MB-iSTFT / iSTFT heads (128 -> 72 / 18 channels, 7 taps) run on the split-operand kernels at batch (round 6); HiFi-GAN's one-channel output conv has its own FIR kernel
      if (m.dec_type != 0 && !pack_bf3(st, m.conv_post)) FAIL("conv_post pack"); }
    if (m.dec_type == 0 && m.is_ms == 1) { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.dec_cond)) FAIL("decoder cond"); }
    if (m.dec_type == 0) { if (m.conv_post.Cout != 1) FAIL("conv_post must have one output channel"); }
    else if (m.dec_type == 2) { if (m.conv_post.Cout != 18) FAIL("iSTFT head must emit 18 channels (iStft(16,4,16) is hard-coded)"); m.hop_total *= 4; }
    else {
        if (m.subbands != 4 || m.conv_post.Cout != 72) FAIL("MB-iSTFT head must emit 4 x 18 channels (pqmf(4)/iStft(16,4,16) are hard-coded)");
        m.hop_total *= 16;
        float* fir = st.alloc(63 * 4, &m.synth_fir);
        if (!fir) FAIL("fir");
        if (m.dec_type == 3) { build_pqmf(fir); m.fir_taps = 63; m.fir_pad = 31; }
        else {   // MS: learned multistream conv, weights [1][k][4]
            HConv h = parse_conv(r);
            if (!r.ok || h.out_ch != 1 || h.in_ch != 4 || h.k > 63 || h.dil != 1 || 2 * h.pad != h.k - 1) FAIL("multistream_conv_post shape");
            memcpy(fir, h.w, sizeof(float) * (size_t)h.k * 4);
            m.fir_taps = h.k; m.fir_pad = h.pad;
            // (upstream builds this conv without a bias; the reference's nn_conv1d honours one when the blob carries it: nn_conv1d.cpp:40-46)
            m.fir_bias = h.has_bias == 1 && h.b ? h.b[0] : 0.f;
        }
    }

    // ---- flow: ResidualCouplingBlock.cpp:29-39, ResidualCouplingLayer.cpp:28-30, WN.cpp:32-60
    m.n_flows = r.geti(); r.geti();
    if (!r.ok || m.n_flows < 0 || m.n_flows > 64) FAIL("flow header");
    if (m.inter & 1) FAIL("odd latent channel count");
    m.cp.resize(m.n_flows);
    for (int i = 0; i < m.n_flows; i++) {
        DCoupling& c = m.cp[i];
        // couplings run i = n-1 .. 0 with a channel reversal BEFORE each (ResidualCouplingBlock.cpp:64-67):
        // coupling i sees (n - i) reversals.  Odd => fold the reversal into its weights.
        c.flipped = ((m.n_flows - i) & 1) != 0;
        HConv pre = parse_conv(r);
        if (!r.ok || pre.k != 1 || pre.in_ch != m.inter / 2) FAIL("coupling pre");
        { PackOpts o; o.reverse_in = c.flipped; if (!pack_conv(st, pre, o, c.pre) || !pack_bf3(st, c.pre)) FAIL("coupling pre pack"); }
        DWn& w = c.wn;
        w.n = r.geti(); const int wk = r.geti();
        if (!r.ok || w.n <= 0 || w.n > 64) FAIL("WN header");
        w.in.resize(w.n); w.rs.resize(w.n);
        int dil = 1;   // dilation_rate == 1 (SynthesizerTrn.cpp:134): dilation = rate^(i+1) = 1
        for (int l = 0; l < w.n; l++) {
            HConv h = parse_conv(r);
            if (!r.ok || (h.out_ch & 1)) FAIL("WN in_layer");
            if (l == 0) w.H = h.in_ch;
            if (h.in_ch != w.H || h.out_ch != 2 * w.H || h.k != wk) FAIL("WN in_layer shape");
            h.pad = (wk * dil - dil) / 2; h.dil = dil;
            PackOpts o; o.gate = true; o.gate_H = w.H;
            if (!pack_conv(st, h, o, w.in[l]) || !pack_bf3(st, w.in[l])) FAIL("WN in_layer pack");
        }
        for (int l = 0; l < w.n; l++) {
            HConv h = parse_conv(r);
            if (!r.ok || h.k != 1 || h.in_ch != w.H || (h.out_ch != 2 * w.H && h.out_ch != w.H)) FAIL("WN res_skip");
            if ((l < w.n - 1) != (h.out_ch == 2 * w.H)) FAIL("WN res_skip shape");
            if (!pack_conv(st, h, PackOpts(), w.rs[l]) || !pack_bf3(st, w.rs[l])) FAIL("WN res_skip pack");
            w.rs[l].H = w.H;
        }
        w.has_cond = m.is_ms == 1;
        if (w.has_cond) {
            HConv h = parse_conv(r);
            if (!r.ok || h.out_ch != 2 * w.H * w.n) FAIL("WN cond_layer");
            PackOpts o; o.gate_blocks = true; o.gate_H = w.H;
            if (!pack_conv(st, h, o, w.cond)) FAIL("WN cond pack");
        }
        HConv post = parse_conv(r);
        if (!r.ok || post.k != 1 || post.out_ch != m.inter / 2 || post.in_ch != w.H || pre.out_ch != w.H) FAIL("coupling post");
        { PackOpts o; o.reverse_out = c.flipped; if (!pack_conv(st, post, o, c.post) || !pack_bf3(st, c.post)) FAIL("coupling post pack"); }
        if (!pack_flow_fused(st, c)) FAIL("coupling fused-layer pack");
    }

    // ---- duration predictor
    if (m.dur_type == 0) {   // /root/reference/src/models/StochasticDurationPredictor.cpp:41-70
        m.sdp_flows = r.geti();
        if (!r.ok || m.sdp_flows < 2 || m.sdp_flows > 64) FAIL("sdp flows");
        { const float* e = r.take(4); if (!r.ok) FAIL("sdp affine"); m.ea_m = e[0]; m.ea_logs = e[2]; }   // m[2], logs[2]; channel 0 is used
        m.cf.resize(m.sdp_flows);
        for (int i = 0; i < m.sdp_flows; i++) if (!parse_pack_convflow(r, st, m.cf[i])) FAIL("sdp ConvFlow");
        parse_conv(r); parse_conv(r); skip_dds(r); r.take(4);   // posterior side: loaded by the reference, never run
        for (int i = 0; i < 4; i++) skip_convflow(r);
        { HConv h = parse_conv(r); if (!r.ok || h.k != 1 || !pack_conv(st, h, PackOpts(), m.sdp_pre)) FAIL("sdp pre"); }
        { HConv h = parse_conv(r); if (!r.ok || h.k != 1 || !pack_conv(st, h, PackOpts(), m.sdp_proj)) FAIL("sdp proj"); }
        if (!parse_pack_dds(r, st, m.sdp_dds)) FAIL("sdp dds");
        if (m.is_ms == 1) { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.sdp_cond)) FAIL("sdp cond"); }
    } else {                 // /root/reference/src/models/FixDurationPredictor.cpp:33-44
        { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.fix_c1)) FAIL("fix conv_1"); }
        { HLn h = parse_ln(r); if (!r.ok || !pack_ln(st, h, m.fix_n1)) FAIL("fix norm_1"); }
        { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.fix_c2)) FAIL("fix conv_2"); }
        { HLn h = parse_ln(r); if (!r.ok || !pack_ln(st, h, m.fix_n2)) FAIL("fix norm_2"); }
        { HConv h = parse_conv(r); if (!r.ok || h.out_ch != 1 || !pack_conv(st, h, PackOpts(), m.fix_proj)) FAIL("fix proj"); }
        if (m.is_ms == 1) { HConv h = parse_conv(r); if (!r.ok || !pack_conv(st, h, PackOpts(), m.fix_cond)) FAIL("fix cond"); }
    }
    if (m.is_ms == 1) {      // /root/reference/src/models/SynthesizerTrn.cpp:155-163
        m.spk_num = r.geti(); m.gin = r.geti();
        if (!r.ok || m.spk_num <= 0 || m.gin <= 0) FAIL("speaker table header");
        const float* e = r.take((int64_t)m.spk_num * m.gin);
        if (!r.ok || !copy_raw(st, e, (size_t)m.spk_num * m.gin, &m.emb_g)) FAIL("speaker table");
    }
    if (!r.ok) FAIL("blob truncated");
    m.consumed = r.o;
    { const float* slack; if (!st.alloc(1024, &slack)) FAIL("weight store overflow"); }   // kernels may read one row tile past a tensor
    m.dev_floats = st.used;
    if (!st.commit()) FAIL("mapping device memory behind the weight store failed");
    m.dev_mapped = st.mapped; m.dev_handle = st.vmm ? (void*)st.handle : nullptr;
    if (hipMemcpy(st.dev, st.host.data(), st.used * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) FAIL("weight upload failed");
    return true;
}

void free_model(Model& m) {
    if (m.dev_weights) {
        if (m.dev_vmm) {
            if (m.dev_mapped) { (void)hipMemUnmap(m.dev_weights, m.dev_mapped); (void)hipMemRelease((hipMemGenericAllocationHandle_t)m.dev_handle); }
            (void)hipMemAddressFree(m.dev_weights, m.dev_reserved);
        } else (void)hipFree(m.dev_weights);
    }
    m.dev_weights = nullptr; m.dev_mapped = 0;
}

}  // namespace sts
