// SynthesizerTrn.h -- source-compatible with huakunyang/SummerTTS include/SynthesizerTrn.h:9-19:
// same class name, same four public members with the same signatures, one pimpl pointer as the only
// data member (so the object layout and the mangled symbols are those of the reference and
// test/main.cpp links against libsummertts_hip.so unchanged).  The acoustic model + vocoder behind
// infer() run on an AMD MI355X through the C ABI in summertts_hip.h.
#ifndef _TTS_SYNTHESIZER_H_
#define _TTS_SYNTHESIZER_H_

#include "stdint.h"
#include "string"

using namespace std;   // kept: the reference header does this and callers rely on it

class SynthesizerTrn
{
public:
    // modelData/modelSize as returned by ttsLoadModel (size in BYTES).  The blob is copied.
    SynthesizerTrn(float * modelData, int32_t modelSize);
    // Returns malloc()'d int16 PCM @16 kHz (free with tts_free_data); dataLen = sample count.
    int16_t * infer(const string & line, int32_t sid, float lengthScale, int32_t & dataLen);
    int32_t getSpeakerNum();
    ~SynthesizerTrn();

private:
    void * priv_;
};

#endif
