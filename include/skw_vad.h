/*
 * skw_vad.h — C ABI of the Silero-VAD gate (libskw_vad.so): SURVEY.md §8a row W2.
 *
 * What a host binds in place of the `ort` session the reference drives
 * (/root/reference/plugins/native/whisper/src/vad.rs):
 *
 *   reference                                            this library
 *   ---------------------------------------------------  ---------------------------------------
 *   SileroVAD::new(model_path, 16000, threshold) :34-55   skw_vad_create(model_path, err, errlen)
 *   vad.process_chunk(&frame[..512]) -> prob     :67-120  skw_vad_process_chunk(vad, frame512, &prob)
 *   vad.reset()                                  :139-142 skw_vad_reset(vad)
 *   state [2,1,128] (outputs[1])                 :107-114 skw_vad_state(vad, out256)
 *   drop                                                  skw_vad_free(vad)
 *
 * The threshold comparison (`probability >= threshold`, vad.rs:131-134, lib.rs:421) stays with the caller, as in the reference's
 * `process` loop.  CPU code by design (a 128-unit LSTM stepped 31 times per audio-second; strictly sequential per stream); the
 * same header-only implementation (streamkit_amd/csrc/skw_silero.h) is compiled into libwhisper.so.
 */
#ifndef SKW_VAD_H
#define SKW_VAD_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct skw_vad skw_vad;
/* Reads a Silero VAD v5/v6 .onnx file (16 kHz branch).  NULL on failure, with the reference's message shape in err:
 * "Failed to load VAD model from '<path>': <reason>". */
skw_vad* skw_vad_create(const char* onnx_path, char* err, size_t errlen);
/* exactly 512 samples of 16 kHz mono f32; returns 0 and the speech probability, updates the carried state and 64-sample context */
int  skw_vad_process_chunk(skw_vad*, const float* frame512, float* probability);
void skw_vad_reset(skw_vad*);
void skw_vad_state(const skw_vad*, float* out256);   /* [2][1][128]: h then c */
void skw_vad_free(skw_vad*);
#ifdef __cplusplus
}
#endif
#endif
