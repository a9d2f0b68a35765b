// tts_ids -- command-line demo in the shape of the reference's test/main.cpp:75-148, built only on the public
// headers of this repo (include/SynthesizerTrn.h, include/utils.h): read a text file, load a .bin model,
// call SynthesizerTrn::infer, write a 16 kHz mono s16 WAV (44-byte RIFF header as test/main.cpp:7-65).
// Until the text frontend is wired the text file holds phoneme ids ("12 7 0 33 ...").
//
//   g++ -O2 -I include tools/cli/tts_ids.cpp -L summertts_amd/lib -lsummertts_hip -Wl,-rpath,$PWD/summertts_amd/lib -o tts_ids
//   ./tts_ids ids.txt model.bin out.wav
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <string>

#include "SynthesizerTrn.h"
#include "utils.h"

static void put32(char* p, uint32_t v) { memcpy(p, &v, 4); }
static void put16(char* p, uint16_t v) { memcpy(p, &v, 2); }

// RIFF/WAVE header: PCM, 1 channel, 16 kHz, 16 bit
static void wav_header(char* h, uint32_t data_bytes) {
    memcpy(h, "RIFF", 4); put32(h + 4, 36 + data_bytes); memcpy(h + 8, "WAVEfmt ", 8);
    put32(h + 16, 16); put16(h + 20, 1); put16(h + 22, 1); put32(h + 24, 16000); put32(h + 28, 16000 * 2);
    put16(h + 32, 2); put16(h + 34, 16); memcpy(h + 36, "data", 4); put32(h + 40, data_bytes);
}

static int write_wav(const char* path, const int16_t* pcm, int32_t n) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    char h[44];
    wav_header(h, (uint32_t)n * 2);
    fwrite(h, 44, 1, f);
    fwrite(pcm, 2, (size_t)n, f);
    fclose(f);
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: %s <text-or-ids file> <model.bin> <out.wav>\n", argv[0]); return 1; }
    std::ifstream in(argv[1]);
    if (!in) { printf("Failed to open %s\n", argv[1]); return 0; }
    std::string line, sub;
    while (std::getline(in, sub)) {
        if (sub.size() >= 3 && (unsigned char)sub[0] == 0xEF && (unsigned char)sub[1] == 0xBB && (unsigned char)sub[2] == 0xBF) sub = sub.substr(3);
        line = line + sub + "  ";      // whole file = one utterance, as the reference demo does
    }
    float* dataW = NULL;
    int32_t modelSize = ttsLoadModel(argv[2], &dataW);
    if (modelSize < 0) { printf("Failed to load %s\n", argv[2]); return 1; }
    SynthesizerTrn* synthesizer = new SynthesizerTrn(dataW, modelSize);
    int32_t spkNum = synthesizer->getSpeakerNum();
    printf("Available speakers in the model are %d\n", spkNum);
    int rc = 0;
    if (spkNum > 20) {                 // multi-speaker demo loop of test/main.cpp:108-129
        for (int spkID = 10; spkID < 20; spkID++) {
            int32_t retLen = 0;
            int16_t* wavData = synthesizer->infer(line, spkID, 1.1, retLen);
            if (!wavData) { rc = 2; break; }
            char fileName[512];
            snprintf(fileName, sizeof(fileName), "%s_%d.wav", argv[3], spkID);
            write_wav(fileName, wavData, retLen);
            printf("%s generated\n", fileName);
            tts_free_data(wavData);
        }
    } else {
        int32_t retLen = 0;
        int16_t* wavData = synthesizer->infer(line, 0, 1.0, retLen);
        if (!wavData) rc = 2;
        else { write_wav(argv[3], wavData, retLen); printf("%s: %d samples (%.2f s)\n", argv[3], retLen, retLen / 16000.0); tts_free_data(wavData); }
    }
    delete synthesizer;
    tts_free_data(dataW);
    return rc;
}
