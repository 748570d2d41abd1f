// skw_resampler_core.h — the host side of the GPU resampler, shared by libresampler.so (kind `resampler`) and by libwhisper.so's
// `input_sample_rate` front end.  Restates AudioResamplerNode::run, /root/reference/crates/nodes/src/audio/filters/resampler.rs:148-743:
//   R1 resample loop (:384-514)  -> skw_resample_linear (HIP kernels, bit-exact with rubato FastFixedIn / Linear)
//   R2 pass-through / re-chunk   (:299-373)
//   R3 remainder + final frame   (:543-730)  -> finish()
//   R4 packetisation to output_frame_size; timing metadata does not cross the native ABI for audio packets
//      (sdks/plugin-sdk/native/src/conversions.rs:342-346), so only sample counts are observable downstream.
// A sink receives every block of output samples in order (interleaved f32 at the target rate).
#pragma once
#include "../../include/skw_engine.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

namespace skw {

struct ResamplerCore {
    using Sink = std::function<bool(const float*, size_t, std::string*)>;
    uint32_t target = 0; size_t chunk_frames = 960, out_frame = 960; int gpu_device = 0;
    bool init = false, needs = false; uint32_t rate = 0; uint16_t channels = 0;
    skw_dsp* dsp = nullptr; skw_resampler_state st{};
    std::vector<float> sample_buffer, output_buffer, scratch;
    // mode "polyphase" (additive): streaming form of skw_resample_polyphase, state on the device
    bool polyphase = false; int L = 1, M = 1, T = 32; skw_pp_stream* pp = nullptr;
    ResamplerCore() = default;
    ResamplerCore(const ResamplerCore&) = delete; ResamplerCore& operator=(const ResamplerCore&) = delete;
    ~ResamplerCore() { if (pp) skw_polyphase_stream_free(pp); if (dsp) skw_dsp_free(dsp); }

    bool drain(const Sink& sink, std::string* err) {
        const size_t fs = out_frame * channels; size_t off = 0;
        while (output_buffer.size() - off >= fs) { if (!sink(output_buffer.data() + off, fs, err)) return false; off += fs; }
        output_buffer.erase(output_buffer.begin(), output_buffer.begin() + off); return true;
    }
    // frames each of the next n_chunks chunks will produce: the index recurrence of rubato's FastFixedIn (the engine walks the same IEEE additions on long calls, skw_engine.hip
    // skw_resample_linear) replayed on the host from the state BEFORE the call — so that a batched GPU call can still be handed on chunk by chunk
    static void chunk_counts(const skw_resampler_state& s, int n_chunks, std::vector<int>* counts) {
        const double t_ratio = 1.0 / s.ratio, end_idx = (double)(s.chunk_frames - 9) - std::ceil(t_ratio); double x0 = s.last_index;
        counts->assign((size_t)n_chunks, 0);
        for (int c = 0; c < n_chunks; ++c) { double x = x0; int n = 0; while (x < end_idx) { x += t_ratio; ++n; } (*counts)[(size_t)c] = n; x0 = x - (double)s.chunk_frames; }
    }
    std::vector<int> last_counts;      // per chunk of the last run_chunks call
    bool run_chunks(skw_resampler_state* s, const float* in, int n_chunks, std::vector<float>* out, std::string* err) {
        const double ratio = s->ratio; const int cap = (int)((double)n_chunks * s->chunk_frames * ratio) + 64;
        chunk_counts(*s, n_chunks, &last_counts);
        out->resize((size_t)cap * channels); int n = 0;
        if (skw_resample_linear(dsp, s, in, n_chunks, out->data(), cap, &n) != 0) { *err = std::string("Resampling failed: ") + skw_dsp_last_error(dsp); return false; }
        long sum = 0; for (int c : last_counts) sum += c;
        if (sum != n) {
            *err = "Resampling failed: the host's replay of the index recurrence disagrees with the device (" + std::to_string(sum) + " vs " + std::to_string(n) + " frames)"; return false;
        }
        out->resize((size_t)n * channels); return true;
    }
    // streaming polyphase: every output whose filter support has arrived (all of them at end of stream); the input tail that later
    // outputs need stays in HBM (skw_polyphase_stream_*), a packet uploads only its own frames
    bool polyphase_step(const float* in, long n_frames, bool final_call, std::vector<float>* out, std::string* err) {
        out->clear();
        if (!pp) return true;
        const long cap = (n_frames + 2L * T + 8) * L / M + 64 + (final_call ? (long)T * L / M + 64 : 0);
        out->resize((size_t)cap * channels); long got = 0;
        if (skw_polyphase_stream_push(pp, in, n_frames, final_call ? 1 : 0, out->data(), cap, &got) != 0) { *err = std::string("Resampling failed: ") + skw_dsp_last_error(dsp); return false; }
        out->resize((size_t)got * channels); return true;
    }
    // scratch -> re-chunker, or straight out.  Without re-chunking the reference sends ONE packet per processed chunk (resampler.rs:471-510), also when a chunk produced no
    // frame; per_chunk: scratch holds the output of last_counts.size() chunks back to back (the linear mode); else one packet (the additive polyphase mode, which has no chunks)
    bool deliver(const Sink& sink, std::string* err, bool per_chunk = false) {
        if (out_frame > 0) { if (scratch.empty()) return true; output_buffer.insert(output_buffer.end(), scratch.begin(), scratch.end()); return drain(sink, err); }
        if (!per_chunk) return scratch.empty() ? true : sink(scratch.data(), scratch.size(), err);
        size_t off = 0;
        for (int n : last_counts) { if (!sink(scratch.data() + off, (size_t)n * channels, err)) return false; off += (size_t)n * channels; }
        return true;
    }

    // one input packet (resampler.rs:247-527).  false + *err on failure.
    bool push(const float* samples, size_t count, uint32_t in_rate, uint16_t in_channels, const Sink& sink, std::string* err) {
        if (!init) {
            init = true; needs = in_rate != target; rate = in_rate; channels = in_channels;
            if (needs) {
                if (in_channels < 1 || in_channels > 2) { *err = "Failed to create resampler: only mono and stereo are supported on the GPU path"; return false; }
                char eb[512] = {0}; dsp = skw_dsp_create(gpu_device, eb, sizeof eb);
                if (!dsp) { *err = std::string("Failed to create resampler: ") + eb; return false; }
                skw_resampler_init(&st, (double)target / (double)rate, (int)chunk_frames, channels);
                { long a = rate, b = target; while (b) { long t2 = a % b; a = b; b = t2; } L = (int)(target / a); M = (int)(rate / a); T = 32 * std::max(1, (M + L - 1) / L); }
                if (polyphase) { pp = skw_polyphase_stream_create(dsp, channels, (int)rate, (int)target);
                if (!pp) { *err = std::string("Failed to create resampler: ") + skw_dsp_last_error(dsp); return false; } }
            }
        }
        if (in_rate != rate || in_channels != channels) {
            char b[200]; snprintf(b, sizeof b, "Audio format changed mid-stream: expected %uHz/%uch, got %uHz/%uch", rate, channels, in_rate, in_channels); *err = b; return false;
        }
        if (!needs) {
            if (out_frame == 0) return sink(samples, count, err);
            output_buffer.insert(output_buffer.end(), samples, samples + count);
            return drain(sink, err);
        }
        if (polyphase) {
            if (!polyphase_step(samples, (long)(count / channels), false, &scratch, err)) return false;
            return deliver(sink, err);
        }
        sample_buffer.insert(sample_buffer.end(), samples, samples + count);
        const size_t cs = chunk_frames * channels; const int n_chunks = (int)(sample_buffer.size() / cs);
        if (n_chunks > 0) {
            if (!run_chunks(&st, sample_buffer.data(), n_chunks, &scratch, err)) return false;
            sample_buffer.erase(sample_buffer.begin(), sample_buffer.begin() + (size_t)n_chunks * cs);
            if (!deliver(sink, err, true)) return false;
        }
        return true;
    }
    // end of stream (resampler.rs:543-730): the remainder through a fresh resampler, then the final short frame
    bool finish(const Sink& sink, std::string* err) {
        if (needs && polyphase) {
            if (!polyphase_step(nullptr, 0, true, &scratch, err)) return false;
            if (!deliver(sink, err)) return false;
        }
        if (needs && !polyphase && !sample_buffer.empty()) {
            const size_t rem = sample_buffer.size() / channels;
            if (rem >= 1) {   // fresh resampler sized to the remainder: zero history, last_index = -4 (resampler.rs:564-570)
                skw_resampler_state t; skw_resampler_init(&t, (double)target / (double)rate, (int)rem, channels);
                if (!run_chunks(&t, sample_buffer.data(), 1, &scratch, err)) return false;
                if (!deliver(sink, err, true)) return false;
            }
            sample_buffer.clear();
        }
        if (!output_buffer.empty() && out_frame > 0) { if (!sink(output_buffer.data(), output_buffer.size(), err)) return false; output_buffer.clear(); }
        return true;
    }
};

}  // namespace skw
