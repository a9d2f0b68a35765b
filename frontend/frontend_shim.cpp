// frontend/frontend_shim.cpp -- C ABI over the REFERENCE's host text frontend.
//
// north_star keeps the text frontend (text normalisation -> word segmentation -> pinyin / IPA -> phoneme ids) on host C++:
// this file is linked against the reference's own frontend classes compiled in place from /root/reference (frontend/Makefile;
// nothing is copied) and exposes the two things libsummertts_hip.so needs from them:
//   * stsfe_create      walks the frontend sections that follow the acoustic sections of a model blob and constructs the
//                       frontend objects from them -- the section walk of /root/reference/src/models/SynthesizerTrn.cpp:165-297,
//                       including its alignment rule `off += off % 4` (:192-195, :264-267, :291-294; NOT a round-up);
//   * stsfe_text_to_ids the call sequence of SynthesizerTrn.cpp:327-355: Chinese = tag -> verbalize -> jieba Cut ->
//                       hanzi2phoneid::convert, English = EnglishText2Id::getIPAId.
// stsfe_scan_sections repeats the walk on sizes only (no object construction) so that the offset arithmetic can be tested
// without the WeTextProcessing FSTs / jieba dictionaries that only a real model blob carries.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <streambuf>
#include <string>
#include <vector>

#include "EnglishText2Id.h"
#include "cppjieba/Jieba.hpp"
#include "hanzi2phoneid.h"
#include "processor/processor.h"

namespace {

struct membuf : std::streambuf {
    membuf(char* b, char* e) { this->setg(b, b, e); }
};

enum { LANG_CHS = 0, LANG_ENG = 1 };      // SynthesizerTrn.cpp:59-64

struct Frontend {
    int lang = 0;
    wetext::Processor* tn = nullptr;
    cppjieba::Jieba* jieba = nullptr;
    hanzi2phoneid* hz2id = nullptr;
    EnglishText2Id* eng = nullptr;
    std::vector<std::string> words;
    int64_t end_float = 0;                 // float offset behind the last section consumed
};

// byte offset -> the reference's "aligned" float offset (SynthesizerTrn.cpp:192-197): off += off % 4, then / 4
int64_t ref_align(int64_t off_char) {
    if (off_char % 4 > 0) off_char = off_char + (off_char % 4);
    return off_char / 4;
}

// Walks the Chinese sections.  out[] = {tn_start, tn_tagger_bytes, tn_verbalizer_bytes, jieba_start, jieba sizes[5],
// poly_start, poly sizes[2], end}; every *_start is the float offset of the section's first payload byte; -1 = absent.
// Returns 0 or a negative code when a section would run past the blob.
int walk_chs(const float* blob, int64_t bytes, int64_t offset, int64_t* out) {
    for (int i = 0; i < 13; i++) out[i] = -1;
    int64_t offset_char = offset * 4;
    if (offset * 4 + 1 < bytes) {                                  // :181 text-normalisation FSTs
        if ((offset + 2) * 4 > bytes) return -2;
        const int64_t tagger = (int64_t)blob[offset], verb = (int64_t)blob[offset + 1];
        offset += 2;
        if (tagger < 0 || verb < 0 || offset * 4 + tagger + verb > bytes) return -2;
        out[0] = offset; out[1] = tagger; out[2] = verb;
        offset_char = offset * 4 + tagger + verb;
        offset = ref_align(offset_char);
    }
    if (offset_char + 1 < bytes) {                                 // :209 jieba dictionaries
        if ((offset + 5) * 4 > bytes) return -3;
        int64_t tot = 0;
        for (int i = 0; i < 5; i++) { out[4 + i] = (int64_t)blob[offset + i]; if (out[4 + i] < 0) return -3; tot += out[4 + i]; }
        offset += 5;
        if (offset * 4 + tot > bytes) return -3;
        out[3] = offset;
        offset_char = offset * 4 + tot;
        offset = ref_align(offset_char);
    }
    if (offset_char + 1 < bytes) {                                 // :272 polyphone word / pinyin lists
        if ((offset + 2) * 4 > bytes) return -4;
        const int64_t w = (int64_t)blob[offset], p = (int64_t)blob[offset + 1];
        offset += 2;
        if (w < 0 || p < 0 || offset * 4 + w + p > bytes) return -4;
        out[9] = offset; out[10] = w; out[11] = p;
        offset_char = offset * 4 + w + p;
        offset = ref_align(offset_char);
    }
    out[12] = offset;
    return 0;
}

}  // namespace

extern "C" {

int stsfe_scan_sections(const float* blob, int64_t blob_bytes, int64_t acoustic_floats, int32_t lang, int64_t* out13) {
    if (!blob || !out13 || acoustic_floats < 0 || acoustic_floats * 4 > blob_bytes) return -1;
    if (lang != LANG_CHS) { for (int i = 0; i < 13; i++) out13[i] = -1; return 0; }
    return walk_chs(blob, blob_bytes, acoustic_floats, out13);
}

// Returns a handle, or NULL when the blob carries no (usable) frontend sections behind float offset `acoustic_floats`.
void* stsfe_create(float* blob, int64_t blob_bytes, int64_t acoustic_floats, int32_t lang) {
    if (!blob || acoustic_floats < 0 || acoustic_floats * 4 > blob_bytes) return nullptr;
    Frontend* f = new Frontend();
    f->lang = lang;
    int64_t offset = acoustic_floats;
    if (lang == LANG_ENG) {                                        // SynthesizerTrn.cpp:169-176
        if ((uint64_t)blob_bytes > (uint64_t)(offset + 1) * sizeof(float)) {
            int32_t cur = 0;
            f->eng = new EnglishText2Id(blob + offset, cur);
            f->end_float = offset + cur;
        }
        if (!f->eng) { delete f; return nullptr; }
        return f;
    }
    if (lang != LANG_CHS) { delete f; return nullptr; }
    int64_t sec[13];
    if (walk_chs(blob, blob_bytes, offset, sec) != 0 || sec[3] < 0 || sec[9] < 0) { delete f; return nullptr; }
    char* base = (char*)blob;
    if (sec[0] >= 0) {                                             // :186-202
        char* p = base + sec[0] * 4;
        membuf bt(p, p + sec[1]), bv(p + sec[1], p + sec[1] + sec[2]);
        std::istream it(&bt), iv(&bv);
        f->tn = new wetext::Processor(it, iv);
    }
    {                                                              // :217-260
        char* p = base + sec[3] * 4;
        char* q[6];
        q[0] = p;
        for (int i = 0; i < 5; i++) q[i + 1] = q[i] + sec[4 + i];
        membuf b0(q[0], q[1]), b1(q[1], q[2]), b2(q[2], q[3]), b3(q[3], q[4]), b4(q[4], q[5]);
        std::istream i0(&b0), i1(&b1), i2(&b2), i3(&b3), i4(&b4);
        f->jieba = new cppjieba::Jieba(i0, i1, i2, i3, i4);
    }
    {                                                              // :277-287
        char* p = base + sec[9] * 4;
        membuf bw(p, p + sec[10]), bp(p + sec[10], p + sec[10] + sec[11]);
        std::istream iw(&bw), ip(&bp);
        f->hz2id = new hanzi2phoneid(iw, ip);
    }
    f->end_float = sec[12];
    return f;
}

int64_t stsfe_sections_end(void* h) { return h ? ((Frontend*)h)->end_float : -1; }

// text (UTF-8) -> phoneme ids, malloc()'d (release with stsfe_free).  Returns 0, or -1 on bad arguments.
int stsfe_text_to_ids(void* h, const char* utf8, int32_t** ids_out, int32_t* n_out) {
    Frontend* f = (Frontend*)h;
    if (!f || !utf8 || !ids_out || !n_out) return -1;
    *ids_out = nullptr; *n_out = 0;
    const std::string line(utf8);
    if (f->lang == LANG_CHS) {                                     // SynthesizerTrn.cpp:329-341
        std::string tn = line;
        if (f->tn) tn = f->tn->verbalize(f->tn->tag(line));
        f->jieba->Cut(tn, f->words, true);
        int32_t n = 0;
        int32_t* ids = f->hz2id->convert(tn, n, f->words);         // new int32_t[]
        int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
        if (!out) return -1;
        for (int32_t i = 0; i < n; i++) out[i] = ids[i];
        delete[] ids;
        *ids_out = out; *n_out = n;
        return 0;
    }
    std::vector<int> v = f->eng->getIPAId(line);                   // SynthesizerTrn.cpp:343-353 (the 0.83 factor stays with the caller)
    int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (v.empty() ? 1 : v.size()));
    if (!out) return -1;
    for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
    *ids_out = out; *n_out = (int32_t)v.size();
    return 0;
}

void stsfe_free(void* p) { free(p); }

void stsfe_destroy(void* h) {
    Frontend* f = (Frontend*)h;
    if (!f) return;
    delete f->tn; delete f->jieba; delete f->hz2id; delete f->eng;
    delete f;
}

}  // extern "C"
