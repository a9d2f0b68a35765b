// synthesizer_trn.hip -- the reference's C++ surface (include/SynthesizerTrn.h, utils.h, tts_logger.h,
// tts_file_io.h) implemented over the C ABI, so the reference's own caller (test/main.cpp:100-145)
// links against libsummertts_hip.so unchanged.
//
// Text frontend: TN -> jieba -> pinyin -> phoneme ids (/root/reference/src/models/SynthesizerTrn.cpp:327-355)
// stays host C++ -- the reference's own classes, loaded at run time from libsummertts_frontend.so when the model blob
// carries frontend sections.  Without them infer(string) accepts a line of whitespace/comma separated phoneme ids
// ("12 7 0 33 ...", the frontend's OUTPUT) and reports plain text as an error instead of guessing.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include <sys/stat.h>

#include <string>
#include <vector>

#include "../../include/SynthesizerTrn.h"
#include "../../include/summertts_hip.h"
#include "../../include/tts_file_io.h"
#include "../../include/tts_logger.h"
#include "../../include/utils.h"

void tts_log(TTS_LOG_CAT_t cat, const char* logStr) { (void)cat; printf("%s", logStr); }

int32_t tts_stat(char* filePath, TTS_STAT_t* ttsSt) {
    struct stat st;
    int32_t ret = stat(filePath, &st);
    if (ret != -1) ttsSt->size_ = (int32_t)st.st_size;
    return ret;
}
TTS_FILE_t* tts_fopen(char* filePath) {
    TTS_FILE_t* f = new TTS_FILE_t();
    f->fp_ = (void*)fopen(filePath, "rb");
    return f;
}
void tts_fclose(TTS_FILE_t* f) {
    if (!f) return;
    if (f->fp_) fclose((FILE*)f->fp_);
    delete f;
}
int32_t tts_fread(void* buf, int32_t size, TTS_FILE_t* f) { return (int32_t)fread(buf, size, 1, (FILE*)f->fp_); }

int ttsLoadModel(char* ttsModelName, float** ttsModel) {
    TTS_STAT_t st;
    if (-1 == tts_stat(ttsModelName, &st)) return -1;
    TTS_FILE_t* fp = tts_fopen(ttsModelName);
    if (!fp || !fp->fp_) {
        tts_log(TTS_LOG_ERROR, "TTS_SYNC: Fail to open am model file\n");
        tts_fclose(fp);
        return -1;
    }
    float* data = (float*)malloc(st.size_);
    tts_fread(data, st.size_, fp);
    tts_fclose(fp);
    *ttsModel = data;
    return st.size_;
}
void tts_free_data(void* data) { free(data); }

// ---- optional host text frontend -------------------------------------------------------------------
// frontend/_ref/libsummertts_frontend.so = the reference's own TN / jieba / hanzi2phoneid / EnglishText2Id classes compiled
// in place (frontend/Makefile) behind frontend/frontend_shim.cpp.  It is looked up at run time, never linked: a model blob
// that carries frontend sections behind its acoustic sections gets a frontend (text in, as SynthesizerTrn.cpp:327-355);
// a weights-only blob -- or a machine without the library -- keeps the phoneme-ids-as-text input.
struct FrontendApi {
    void* (*create)(float*, int64_t, int64_t, int32_t) = nullptr;
    int (*text_to_ids)(void*, const char*, int32_t**, int32_t*) = nullptr;
    void (*free_ids)(void*) = nullptr;
    void (*destroy)(void*) = nullptr;
    bool ok = false;
};

static FrontendApi& frontend_api() {
    static FrontendApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    std::vector<std::string> cand;
    if (const char* e = getenv("SUMMERTTS_FRONTEND_LIB")) cand.push_back(e);
    Dl_info info;
    if (dladdr((void*)&frontend_api, &info) && info.dli_fname) {      // next to / relative to libsummertts_hip.so
        std::string dir(info.dli_fname);
        const size_t sl = dir.find_last_of('/');
        dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
        cand.push_back(dir + "/libsummertts_frontend.so");
        cand.push_back(dir + "/../../frontend/_ref/libsummertts_frontend.so");
    }
    for (const std::string& path : cand) {
        void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) continue;
        api.create = (void* (*)(float*, int64_t, int64_t, int32_t))dlsym(h, "stsfe_create");
        api.text_to_ids = (int (*)(void*, const char*, int32_t**, int32_t*))dlsym(h, "stsfe_text_to_ids");
        api.free_ids = (void (*)(void*))dlsym(h, "stsfe_free");
        api.destroy = (void (*)(void*))dlsym(h, "stsfe_destroy");
        api.ok = api.create && api.text_to_ids && api.free_ids && api.destroy;
        if (api.ok) break;
    }
    return api;
}

struct SynPriv { sts_engine* eng = nullptr; void* fe = nullptr; };

SynthesizerTrn::SynthesizerTrn(float* modelData, int32_t modelSize) {
    SynPriv* p = new SynPriv();
    priv_ = p;
    int dev = 0;
    if (const char* s = getenv("SUMMERTTS_HIP_DEVICE")) dev = atoi(s);
    if (sts_create(modelData, modelSize, dev, &p->eng) != STS_OK) {
        tts_log(TTS_LOG_ERROR, "SynthesizerTrn: ");
        tts_log(TTS_LOG_ERROR, sts_last_error());
        tts_log(TTS_LOG_ERROR, "\n");
        return;
    }
    // frontend sections follow the acoustic sections (SynthesizerTrn.cpp:165-297); the size guards are the reference's
    sts_model_info info;
    sts_get_info(p->eng, &info);
    if ((int64_t)modelSize > (info.blob_floats_consumed + 1) * (int64_t)sizeof(float)) {
        FrontendApi& api = frontend_api();
        if (api.ok) p->fe = api.create(modelData, modelSize, info.blob_floats_consumed, info.lang_type);
        if (!p->fe)
            tts_log(TTS_LOG_ERROR, "SynthesizerTrn: the model carries text-frontend sections but libsummertts_frontend.so is not "
                                   "available (frontend/Makefile); infer() takes phoneme ids as text\n");
    }
}

int32_t SynthesizerTrn::getSpeakerNum() {
    SynPriv* p = (SynPriv*)priv_;
    return p && p->eng ? sts_speaker_num(p->eng) : 1;
}

int16_t* SynthesizerTrn::infer(const string& line, int32_t sid, float lengthScale, int32_t& dataLen) {
    SynPriv* p = (SynPriv*)priv_;
    dataLen = 0;
    if (!p || !p->eng) return NULL;
    std::vector<int32_t> ids;
    if (p->fe) {                                   // text -> ids on the host (SynthesizerTrn.cpp:327-353)
        int32_t* fid = NULL;
        int32_t n = 0;
        if (frontend_api().text_to_ids(p->fe, line.c_str(), &fid, &n) != 0) return NULL;
        ids.assign(fid, fid + n);
        frontend_api().free_ids(fid);
    } else {                                       // no frontend: the line holds the frontend's OUTPUT, "12 7 0 33 ..."
        const char* s = line.c_str();
        while (*s) {
            while (*s == ' ' || *s == '\t' || *s == ',' || *s == '\n' || *s == '\r') s++;
            if (!*s) break;
            char* end = NULL;
            long v = strtol(s, &end, 10);
            if (end == s) {
                tts_log(TTS_LOG_ERROR, "SynthesizerTrn::infer: no text frontend for this model (weights-only blob or "
                                       "libsummertts_frontend.so missing); pass phoneme ids\n");
                return NULL;
            }
            ids.push_back((int32_t)v);
            s = end;
        }
    }
    if (ids.empty()) return NULL;
    sts_model_info info;
    sts_get_info(p->eng, &info);
    if (info.lang_type == 1) lengthScale = lengthScale * 0.83;   // SynthesizerTrn.cpp:354
    int16_t* pcm = NULL;
    int32_t n = 0;
    if (sts_infer_ids(p->eng, ids.data(), (int32_t)ids.size(), sid, lengthScale, &pcm, &n) != STS_OK) {
        tts_log(TTS_LOG_ERROR, sts_last_error());
        tts_log(TTS_LOG_ERROR, "\n");
        return NULL;
    }
    dataLen = n;
    return pcm;
}

SynthesizerTrn::~SynthesizerTrn() {
    SynPriv* p = (SynPriv*)priv_;
    if (p) {
        if (p->fe) frontend_api().destroy(p->fe);
        sts_destroy(p->eng);
        delete p;
    }
}
