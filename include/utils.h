// utils.h -- same two C++ functions as huakunyang/SummerTTS include/utils.h:4-5.
#ifndef _TTS_UTILS_H_
#define _TTS_UTILS_H_

// Reads the whole model file into a malloc()'d buffer; returns its size in BYTES or -1.
int ttsLoadModel(char * ttsModelName, float **ttsModel);
void tts_free_data(void * data);

#endif
