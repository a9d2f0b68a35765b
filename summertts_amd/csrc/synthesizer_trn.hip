// synthesizer_trn.hip -- the reference's C++ surface (include/SynthesizerTrn.h, utils.h, tts_logger.h,
// tts_file_io.h) implemented over the C ABI, so the reference's own caller (test/main.cpp:100-145)
// links against libsummertts_hip.so unchanged.
//
// Scope note: the text frontend (TN -> jieba -> pinyin -> phoneme ids,
// /root/reference/src/models/SynthesizerTrn.cpp:327-355) stays host C++ in the reference and is the next
// row to wire; until then infer(string) accepts a line of whitespace/comma separated phoneme ids
// ("12 7 0 33 ...", the frontend's OUTPUT) and reports plain text as an error instead of guessing.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

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

struct SynPriv { sts_engine* eng = nullptr; };

SynthesizerTrn::SynthesizerTrn(float* modelData, int32_t modelSize) {
    SynPriv* p = new SynPriv();
    priv_ = p;
    int dev = 0;
    if (const char* s = getenv("SUMMERTTS_HIP_DEVICE")) dev = atoi(s);
    if (sts_create(modelData, modelSize, dev, &p->eng) != STS_OK) {
        tts_log(TTS_LOG_ERROR, "SynthesizerTrn: ");
        tts_log(TTS_LOG_ERROR, sts_last_error());
        tts_log(TTS_LOG_ERROR, "\n");
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
    const char* s = line.c_str();
    while (*s) {
        while (*s == ' ' || *s == '\t' || *s == ',' || *s == '\n' || *s == '\r') s++;
        if (!*s) break;
        char* end = NULL;
        long v = strtol(s, &end, 10);
        if (end == s) {
            tts_log(TTS_LOG_ERROR, "SynthesizerTrn::infer: text frontend not wired in this build; pass phoneme ids\n");
            return NULL;
        }
        ids.push_back((int32_t)v);
        s = end;
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
    if (p) { sts_destroy(p->eng); delete p; }
}
