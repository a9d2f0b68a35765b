// model.hpp -- host-side parse of the SummerTTS .bin float stream into typed descriptors, and the
// device-resident, kernel-ready (repacked) form of every weight tensor.
//
// Grammar = the reference constructors (each parse_* cites its constructor).
// The reference copies every weight into Eigen matrices at construction
// (/root/reference/src/nn_op/nn_conv1d.cpp:39); we repack every conv ONCE into the layout the
// gfx950 kernels want ([tap][cin][cout], polyphase for transposed convs, tile-paired rows for the
// gated WaveNet convs, channel flips of the flow folded into the weight order) and upload the lot
// to HBM in a single copy.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace sts {

struct HConv {     // host view into the blob (w: [out][k][in])
    int out_ch = 0, in_ch = 0, k = 0, pad = 0, dil = 1, has_bias = 0, stride = 1;
    const float* w = nullptr;
    const float* b = nullptr;
};
struct HLn { int size = 0; const float* g = nullptr; const float* b = nullptr; };

// device conv: weights repacked, ready for ConvArgs
struct DConv {
    int Cin = 0, Cout = 0, Cin_pad = 0, Cout_pad = 0;
    int k = 0, pad = 0, dil = 1;
    int depthwise = 0;
    int transposed = 0, stride = 1, J = 1;   // polyphase: J taps per phase
    int gate_perm = 0, H = 0;
    const float* w = nullptr;
    const float* bias = nullptr;
    const float* wc = nullptr;      // col_layer / col_proj A-strip copy of a 1x1 conv (DDSConv pointwise convs, attention o-proj, q/k/v, encoder proj)
    const float* bias_rows = nullptr;   // bias in source row order next to wc (== bias where the conv's rows are not permuted)
    const float* wu = nullptr;      // Winograd-domain copy [seg][4][Cin_pad][Cout_pad] (narrow decoder ResBlock convs)
    const void* wb3 = nullptr;      // split-bf16 copy (conv_bf3.hip: three bf16 planes, fragment order) of the decoder trunk convs
    const void* wb3p = nullptr;     // the same with the k order of resblock_bf3_kernel's parked intermediate (second conv of a narrow ResBlock layer)
    const void* wh2 = nullptr;      // two-term fp16 copy in wb3's layout (conv_bf3.hip MATH 1), weights scaled by 1 / h2_scale (a power of two)
    const void* wh2p = nullptr;     // ... in wb3p's k order
    float h2_scale = 1.0f;
    // transposed convs with stride 2 / 4 / 8: the same two copies with row-interleaved phases (ConvArgs::rowph) and the bias per merged row
    const void* wb3r = nullptr; const void* wh2r = nullptr; const float* bias_r = nullptr;
    double macs_per_out = 0;   // true-tap MACs per output position (all output channels)
};
struct DLn { int C = 0; const float* g = nullptr; const float* b = nullptr; };

struct PackOpts {
    bool reverse_in = false;    // input channels in reversed order (flow flip folded in)
    bool reverse_out = false;   // output rows in reversed order
    bool gate = false;          // WN in_layer: rows become (tanh16, sigmoid16) groups inside every 32-row tile when H % 16 == 0
    int gate_H = 0;
    bool gate_blocks = false;   // cond_layer: apply the gate permutation inside every 2H-row block
    bool depthwise = false;
    int out_rows = -1;          // keep only the first out_rows rows (dead `logs` half of the encoder proj)
};

struct DMha { int ch = 0, kc = 0, win = 0, px = 0; const float* relk = nullptr; const float* relv = nullptr; DConv qkv, o; };
struct DFfn { int ksize = 1; DConv c1, c2; };
struct DResBlock { std::vector<DConv> c1, c2; };
struct DWn { int n = 0, H = 0; std::vector<DConv> in, rs; bool has_cond = false; DConv cond; };
// operands of the one-launch-per-layer flow kernel (wn_flow.hip; model.hip pack_flow_fused): the first layer's gate weights in the k order
// of a parked accumulator, and per layer the 1x1 conv on the gated channels with rows [res | -(W_post W_skip_l)] (two-term fp16, perm_k)
struct DFlowFused {
    bool ok = false; int G = 0, Cg = 32;
    const void* gate0 = nullptr; float gate0_scale = 1.f;
    std::vector<const void*> wc; std::vector<float> sc; std::vector<int> rows_c, rows_res;
    std::vector<const float*> b_res;          // res bias per layer (null on the last)
    const float* b_m = nullptr;               // -(b_post + W_post sum_l b_skip_l)
};
struct DCoupling { DConv pre, post; DWn wn; bool flipped = false; DFlowFused ff; };
struct DDds { int n = 0; std::vector<DConv> sep, pw; std::vector<DLn> n1, n2; };
struct DConvFlow { DConv pre, proj; DDds dds; int filter = 0; };

struct Model {
    int is_ms = 0, lang = 0, dur_type = 0, dec_type = 0, spk_num = 0, gin = 0;
    // text encoder
    int hidden = 0, vocab = 0, emb_size = 0, n_layers = 0, inter = 0;
    const float* emb = nullptr;
    std::vector<DMha> mha; std::vector<DLn> ln1, ln2; std::vector<DFfn> ffn; DConv proj;
    // decoder
    int subbands = 4, nfft = 16, hop = 4, n_up = 0, n_resk = 0, up_init = 0;
    std::vector<int> up_rate, up_k;
    DConv conv_pre, conv_post, dec_cond, ms_post;
    std::vector<DConv> ups; std::vector<DResBlock> rb;
    const float* synth_fir = nullptr; int fir_taps = 63, fir_pad = 31; float fir_bias = 0.f;
    int hop_total = 0;
    // flow
    int n_flows = 0; std::vector<DCoupling> cp;
    // duration predictors
    int sdp_flows = 0; float ea_m = 0.f, ea_logs = 0.f;
    std::vector<DConvFlow> cf; DConv sdp_pre, sdp_proj, sdp_cond; DDds sdp_dds;
    DConv fix_c1, fix_c2, fix_proj, fix_cond; DLn fix_n1, fix_n2;
    const float* emb_g = nullptr;
    // device storage
    float* dev_weights = nullptr; size_t dev_floats = 0;
    bool dev_vmm = false; size_t dev_reserved = 0, dev_mapped = 0; void* dev_handle = nullptr;   // virtual-memory form of the store (model.hip Store)
    int64_t consumed = 0;
    std::string error;
};

// Parses `blob`, repacks and uploads to the current HIP device.  Returns false and sets m.error on failure.
bool load_model(const float* blob, int64_t nfloats, Model& m);
void free_model(Model& m);

}  // namespace sts
