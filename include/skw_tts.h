/*
 * skw_tts.h — C ABI of the MI355X speech synthesiser behind the Kokoro TTS node (libskw_tts.so).
 *
 * This is the drop-in boundary for what the reference reaches through its hand-written FFI to the sherpa-onnx C API
 * (/root/reference/plugins/native/kokoro/src/ffi.rs:119-137; linked in build.rs:10).  One entry point per call it makes:
 *
 *   reference (ffi.rs)                                            this library
 *   ---------------------------------------------------------    -------------------------------------------
 *   SherpaOnnxCreateOfflineTts(&SherpaOnnxOfflineTtsConfig)  :121  skw_tts_create(&skw_tts_config, err, errlen)
 *   SherpaOnnxOfflineTtsGenerate(tts, text, sid, speed)      :127  skw_tts_generate(tts, text, sid, speed)
 *     -> *const SherpaOnnxOfflineTtsGeneratedAudio {samples, n, sample_rate}  :21-25   -> const skw_tts_audio* (same three fields, same order)
 *   SherpaOnnxDestroyOfflineTtsGeneratedAudio(audio)         :134  skw_tts_destroy_audio(audio)
 *   SherpaOnnxDestroyOfflineTts(tts)                         :124  skw_tts_destroy(tts)
 *
 * The config carries the fields of SherpaOnnxOfflineTtsKokoroModelConfig the reference fills (kokoro_node.rs:797-806: model, voices,
 * tokens, lexicon, length_scale; data_dir / dict_dir name espeak-ng and jieba data this build does not use) plus the GPU to run on.
 * Plain pointers and sizes only.  There is no CPU fallback: without a gfx950 device skw_tts_create fails and says so.
 *
 * PARITY UNPINNED: the arithmetic the reference runs is Kokoro-82M's ONNX graph inside onnxruntime, neither of which exists in
 * /root/reference or offline.  What this library evaluates is the published Kokoro-82M architecture as recalled (include/skw_kokoro_net.h:
 * ALBERT text encoder -> BiLSTM / AdaLayerNorm duration predictor -> length regulation -> AdainResBlk1d F0 / energy curves -> acoustic text
 * encoder -> AdaIN decoder -> ISTFTNet generator with a harmonic-plus-noise source, n_fft 20, hop 5, 24 kHz), all widths read from the tensor
 * shapes, its weights read by PyTorch module NAME from the initializers of `model`, checked operator by operator against
 * oracle/skw_kokoro_oracle.cpp.  A sherpa-onnx export names its initializers by graph node: create() rejects such a file and says which name it missed.
 */
#ifndef SKW_TTS_H
#define SKW_TTS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct skw_tts skw_tts;

typedef struct {
    const char* model;        /* <model_dir>/model.onnx   (kokoro_node.rs:741) */
    const char* voices;       /* <model_dir>/voices.bin   f32 [n_speakers][510][256] style vectors, row = token count (kokoro_node.rs:742) */
    const char* tokens;       /* <model_dir>/tokens.txt   "symbol id" per line (kokoro_node.rs:743) */
    const char* lexicon;      /* comma-separated lexicon files "word ph ph ..." (kokoro_node.rs:766); NULL / missing files: characters map straight through tokens */
    float length_scale;       /* SherpaOnnxOfflineTtsKokoroModelConfig.length_scale, the reference passes 1.0 */
    int32_t gpu_device;       /* additive: the reference has `execution_provider` instead */
} skw_tts_config;

/* field for field SherpaOnnxOfflineTtsGeneratedAudio (ffi.rs:21-25) */
typedef struct { const float* samples; int32_t n; int32_t sample_rate; } skw_tts_audio;

skw_tts* skw_tts_create(const skw_tts_config* config, char* err, size_t errlen);      /* NULL on failure (message in err) */
void skw_tts_destroy(skw_tts*);
/* text: NUL-terminated UTF-8; sid: speaker row of voices.bin; speed: > 0, durations are divided by it.  NULL on failure (skw_tts_last_error).
 * The returned audio is owned by the caller until skw_tts_destroy_audio; calls on one engine are serialised internally. */
const skw_tts_audio* skw_tts_generate(skw_tts*, const char* text, int32_t sid, float speed);
void skw_tts_destroy_audio(const skw_tts_audio*);
const char* skw_tts_last_error(const skw_tts*);
int32_t skw_tts_num_speakers(const skw_tts*);
int32_t skw_tts_sample_rate(const skw_tts*);                                           /* 24000 */

/* ---- stage taps for the parity tests (tests/test_gpu_kokoro.py): the host-side text -> token ids step, and the last call's intermediates ---- */
int32_t skw_tts_tokenize(skw_tts*, const char* text, int32_t* ids, int32_t cap);        /* ids incl. the pad token at both ends; returns the count (<= cap) */
/* the same from token ids (incl. the pad id at both ends): a chosen number of tokens whatever the lexicon (tests, bench) */
const skw_tts_audio* skw_tts_generate_ids(skw_tts*, const int32_t* ids, int32_t n_ids, int32_t sid, float speed);
/* taps are copied to the host only after skw_tts_debug_enable(tts, 1) (off by default: a product call pays nothing for them) */
void skw_tts_debug_enable(skw_tts*, int on);
/* what: 0 durations [T] (as floats), 1 F0 curve [2 F], 2 energy curve [2 F], 3 decoder output [2 F][C], 4 log-magnitude + phase [120 F + 1][22], 5 ALBERT output [T][hid],
 * 6 bert_encoder output [T][d], 7 acoustic text encoder output [T][d], 8 the source's STFT magnitude + phase [120 F + 1][22];
 * returns the element count of the last call with taps on, copies min(count, cap) */
long skw_tts_debug_get(skw_tts*, int what, float* out, long cap);
/* which convolution kernel the next calls use: 0 automatic, 1 untiled, 2 LDS-tiled (the same f32 chain: results are bit-identical, which a test asserts) */
void skw_tts_debug_conv_mode(int mode);
/* which LSTM kernel: 0 automatic, 1 one workgroup per direction, 2 H / 32 workgroups per direction with the recurrent weights resident in LDS (bit-identical, asserted by a test) */
void skw_tts_debug_lstm_mode(int mode);
/* timing of the last generate (GPU events): milliseconds */
float skw_tts_last_ms(const skw_tts*);

#ifdef __cplusplus
}
#endif
#endif
