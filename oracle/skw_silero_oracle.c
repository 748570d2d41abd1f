/*
 * skw_silero_oracle.c — CPU restatement of one Silero-VAD step (TEST INFRASTRUCTURE; nothing under streamkit_amd/ links this).
 *
 * PARITY UNPINNED: the model file and onnxruntime are third-party and absent offline; this restates the
 * published v5 graph as recalled, with the call contract of /root/reference/plugins/native/whisper/src/vad.rs:67-120:
 *   input  [1, 576]  = 64 context samples + 512 new samples         (vad.rs:72-80)
 *   state  [2, 1, 128] = (h, c) of the LSTM cell, carried by the caller (vad.rs:87, 107-114)
 *   sr     16000                                                      (vad.rs:82)
 *   output probability [1, 1]                                         (vad.rs:100-104)
 * It is pinned by tests/test_cpu_silero.py against torch.nn.functional (conv1d, LSTMCell semantics) on seeded weights.
 * Weights arrive as plain arrays (tests/onnx_mini.py reads the file): the product has its own reader (skw_silero.h).
 */
#include <math.h>
#include <string.h>

typedef struct {
    const float* basis;              /* [258][256] */
    const float* cw[4]; const float* cb[4];
    const float* w_ih; const float* w_hh; const float* b_ih; const float* b_hh;   /* gate blocks i, f, g, o */
    const float* ow; float ob;
} skwo_silero_weights;

static float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

/* x576: context + frame; state: [2][128] updated in place; returns the speech probability */
float skwo_silero_step(const skwo_silero_weights* w, const float* x576, float* state) {
    static const int CI[4] = {129, 128, 64, 64}, CO[4] = {128, 64, 64, 128}, ST[4] = {1, 2, 2, 1};
    float xp[640]; float cur[129 * 4], nxt[129 * 4];
    memcpy(xp, x576, sizeof(float) * 576);
    for (int j = 0; j < 64; ++j) xp[576 + j] = xp[574 - j];                     /* F.pad(x, (0, 64), mode="reflect") */
    /* STFT: conv1d(x, basis, stride 128) -> [258][4]; magnitude of the 129 (re, im) pairs */
    for (int bin = 0; bin < 129; ++bin)
        for (int fr = 0; fr < 4; ++fr) {
            float re = 0.0f, im = 0.0f;
            for (int k = 0; k < 256; ++k) { re += w->basis[bin * 256 + k] * xp[128 * fr + k]; im += w->basis[(129 + bin) * 256 + k] * xp[128 * fr + k]; }
            cur[bin * 4 + fr] = sqrtf(re * re + im * im);
        }
    int T = 4;
    for (int l = 0; l < 4; ++l) {                                                 /* Conv1d(k = 3, padding = 1, stride ST[l]) + ReLU */
        const int To = (T + 2 * 1 - 3) / ST[l] + 1;
        for (int o = 0; o < CO[l]; ++o)
            for (int t = 0; t < To; ++t) {
                float s = w->cb[l][o];
                for (int c = 0; c < CI[l]; ++c)
                    for (int k = 0; k < 3; ++k) {
                        const int p = t * ST[l] + k - 1;
                        if (p < 0 || p >= T) continue;
                        s += w->cw[l][(o * CI[l] + c) * 3 + k] * cur[c * T + p];
                    }
                nxt[o * To + t] = s < 0.0f ? 0.0f : s;
            }
        memcpy(cur, nxt, sizeof(float) * CO[l] * To); T = To;
    }
    /* LSTMCell: gates = W_ih x + b_ih + W_hh h + b_hh; c' = sigmoid(f) c + sigmoid(i) tanh(g); h' = sigmoid(o) tanh(c') */
    float* h = state; float* c = state + 128; float gates[512], hn[128];
    for (int r = 0; r < 512; ++r) {
        float a = w->b_ih[r]; for (int k = 0; k < 128; ++k) a += w->w_ih[r * 128 + k] * cur[k];
        float b = w->b_hh[r]; for (int k = 0; k < 128; ++k) b += w->w_hh[r * 128 + k] * h[k];
        gates[r] = a + b;
    }
    for (int j = 0; j < 128; ++j) {
        const float cn = sigmoidf_(gates[128 + j]) * c[j] + sigmoidf_(gates[j]) * tanhf(gates[256 + j]);
        c[j] = cn; hn[j] = sigmoidf_(gates[384 + j]) * tanhf(cn);
    }
    memcpy(h, hn, sizeof hn);
    float acc = w->ob;                                                            /* ReLU -> Conv1d(128 -> 1, k = 1) -> Sigmoid */
    for (int j = 0; j < 128; ++j) acc += w->ow[j] * (h[j] < 0.0f ? 0.0f : h[j]);
    return sigmoidf_(acc);
}
