// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Drives the *real* SummerTTS acoustic path: this file is linked against the reference's own
// sources compiled in place from /root/reference (see oracle/Makefile; nothing is copied) and
// exposes it as a small C API so the Python tests / bench can use the reference Eigen CPU
// implementation as the parity oracle and as the timed CPU baseline.
//
// Why an ID-driven harness instead of SynthesizerTrn::infer(string): the reference's model
// blobs are absent (.MISSING_LARGE_BLOBS), and a weights-only synthetic blob has no text
// frontend sections, so infer(string) cannot run.  The harness therefore constructs the
// sub-models in the order of /root/reference/src/models/SynthesizerTrn.cpp:103-163 and replays
// the post-frontend pipeline of SynthesizerTrn.cpp:357-396 from caller-supplied phoneme ids.
#include <Eigen/Dense>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using Eigen::Map;
using Eigen::MatrixXf;

#include "DurationPredictor_base.h"
#include "FixDurationPredictor.h"
#include "Generator_Istft.h"
#include "Generator_MBB.h"
#include "Generator_MS.h"
#include "Generator_base.h"
#include "Generator_hifigan.h"
#include "ResidualCouplingBlock.h"
#include "StochasticDurationPredictor.h"
#include "TextEncoder.h"
#include "nn_clamp_min.h"

namespace {

struct Ref {
    int isMS = 0, lang = 0, durType = 0, decType = 0, spkNum = 0, gin = 0;
    int consumed = 0;
    TextEncoder* te = nullptr;
    Generator_base* dec = nullptr;
    ResidualCouplingBlock* flow = nullptr;
    DurationPredictor_base* dp = nullptr;
    MatrixXf emb_g;
    std::map<std::string, MatrixXf> taps;   // last call's stage dumps
    std::vector<int32_t> durations;
    std::vector<float> wave;
    double t_te = 0, t_dp = 0, t_flow = 0, t_dec = 0, t_total = 0;
};

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// SynthesizerTrn.cpp:304-321 (length regulator), restated for the harness with int durations.
MatrixXf expand_rows(const MatrixXf& x, const std::vector<int32_t>& len) {
    long total = 0;
    for (int32_t l : len) total += l;
    if (total < 1) total = 1;   // nn_clamp_min(sum, 1.0)
    MatrixXf ret = MatrixXf::Zero(total, x.cols());
    long r = 0;
    for (size_t i = 0; i < len.size(); i++)
        for (int32_t j = 0; j < len[i]; j++) ret.row(r++) = x.row(i);
    return ret;
}

}  // namespace

extern "C" {

void* ref_create(const float* blob_in, int64_t nfloats) {
    // the reference constructors take a non-const pointer and copy everything out of it
    float* blob = const_cast<float*>(blob_in);
    Ref* r = new Ref();
    int32_t off = 0;
    r->isMS = (int)blob[off++];
    r->lang = (int)blob[off++];
    r->durType = (int)blob[off++];
    r->decType = (int)blob[off++];
    r->te = new TextEncoder(blob, off);
    switch (r->decType) {
        case 0: r->dec = new Generator_hifiGan(blob, off, r->isMS); break;
        case 1: r->dec = new Generator_MS(blob, off, r->isMS); break;
        case 2: r->dec = new Generator_Istft(blob, off, r->isMS); break;
        case 3: r->dec = new Generator_MBB(blob, off, r->isMS); break;
        default: delete r->te; delete r; return nullptr;
    }
    r->flow = new ResidualCouplingBlock(blob, off, 1, r->isMS);
    if (r->durType == 0) r->dp = new StochasticDurationPredictor(blob, off, r->isMS);
    else if (r->durType == 1) r->dp = new FixDurationPredictor(blob, off, r->isMS);
    else return nullptr;
    if (r->isMS == 1) {
        r->spkNum = (int)blob[off++];
        r->gin = (int)blob[off++];
        r->emb_g = Map<MatrixXf>(blob + off, r->spkNum, r->gin);
        r->dp->setMSSpk(r->isMS, r->gin);
        off += r->spkNum * r->gin;
    } else {
        r->dp->setMSSpk(0, 0);
    }
    r->consumed = off;
    (void)nfloats;
    return r;
}

int64_t ref_consumed(void* h) { return ((Ref*)h)->consumed; }
int ref_speaker_num(void* h) { Ref* r = (Ref*)h; return r->spkNum == 0 ? 1 : r->spkNum; }

void ref_destroy(void* h) {
    Ref* r = (Ref*)h;
    if (!r) return;
    delete r->te; delete r->dp; delete r->flow; delete r->dec;
    delete r;
}

// Runs the post-frontend pipeline.  forced_dur (nullable, n entries) overrides ceil(exp(logw)*ls).
// keep_taps != 0 stores stage dumps retrievable with ref_tap().  Returns the sample count.
int64_t ref_infer_ids(void* h, const int32_t* ids, int32_t n, int32_t sid, float lengthScale,
                      const int32_t* forced_dur, int keep_taps) {
    Ref* r = (Ref*)h;
    r->taps.clear();
    double t0 = now_s();
    std::vector<int32_t> idv(ids, ids + n);
    MatrixXf m, logs;
    MatrixXf XX = r->te->forward(idv.data(), n, m, logs);
    double t1 = now_s();
    MatrixXf g;
    if (r->isMS == 1) {
        if (sid < 0 || sid >= r->spkNum) sid = 0;
        g = r->emb_g.row(sid);
    }
    MatrixXf logw = r->dp->forward(XX, g, 0.0f);
    MatrixXf w = logw.array().exp() * lengthScale;
    MatrixXf w_ceil = w.array().ceil();
    r->durations.resize(n);
    for (int i = 0; i < n; i++) r->durations[i] = forced_dur ? forced_dur[i] : (int32_t)w_ceil(i, 0);
    double t2 = now_s();
    MatrixXf z_p = expand_rows(m, r->durations);   // noiseScale == 0  =>  z_p == m_expand
    MatrixXf z = r->flow->forward(z_p, g);
    double t3 = now_s();
    MatrixXf o = r->dec->forward(z, g);
    double t4 = now_s();
    r->wave.assign(o.data(), o.data() + o.rows() * o.cols());
    r->t_te = t1 - t0; r->t_dp = t2 - t1; r->t_flow = t3 - t2; r->t_dec = t4 - t3; r->t_total = t4 - t0;
    if (keep_taps) {
        r->taps["x_enc"] = XX; r->taps["m"] = m; r->taps["logs"] = logs; r->taps["logw"] = logw;
        r->taps["z_p"] = z_p; r->taps["z"] = z;
    }
    return (int64_t)r->wave.size();
}

const float* ref_wave(void* h) { return ((Ref*)h)->wave.data(); }
const int32_t* ref_durations(void* h) { return ((Ref*)h)->durations.data(); }

// int16 quantisation exactly as SynthesizerTrn.cpp:389-396 (executed by the same compiler/ISA as
// the reference build, so out-of-range behaviour is the reference's).
void ref_pcm(void* h, int16_t* out) {
    Ref* r = (Ref*)h;
    for (size_t i = 0; i < r->wave.size(); i++) out[i] = (int16_t)(r->wave[i] * 32737);
}

// Stage dump: Eigen column-major [rows=time, cols=channels] == channel-major, time contiguous.
int ref_tap(void* h, const char* name, const float** data, int32_t* rows, int32_t* cols) {
    Ref* r = (Ref*)h;
    auto it = r->taps.find(name);
    if (it == r->taps.end()) return -1;
    *data = it->second.data(); *rows = (int32_t)it->second.rows(); *cols = (int32_t)it->second.cols();
    return 0;
}

void ref_times(void* h, double* out5) {
    Ref* r = (Ref*)h;
    out5[0] = r->t_te; out5[1] = r->t_dp; out5[2] = r->t_flow; out5[3] = r->t_dec; out5[4] = r->t_total;
}

}  // extern "C"
