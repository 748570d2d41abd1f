// skw_resampler_plugin.cpp — libresampler.so: native plugin `resampler` (registered as plugin::native::resampler), the
// GPU drop-in for the reference's built-in audio::resampler node where a pipeline feeds the Whisper node from a
// non-16 kHz source (Opus decode always yields 48 kHz, crates/nodes/src/audio/codecs/opus.rs:70,103).
// Restates AudioResamplerNode::run, /root/reference/crates/nodes/src/audio/filters/resampler.rs:148-743:
//   R1 resample loop (:384-514)  -> skw_resample_linear (HIP kernels k_resample_scan / k_resample_lerp, bit-exact with rubato Linear)
//   R2 pass-through / re-chunk   (:299-373)
//   R3 remainder + final frame   (:543-730)  -> flush()
//   R4 packetisation to output_frame_size; timing metadata does not cross the native ABI for audio packets
//      (sdks/plugin-sdk/native/src/conversions.rs:342-346), so only sample counts are observable downstream.
// One line of YAML swaps it in: `kind: plugin::native::resampler` with the same params as audio::resampler.
#include "../../include/streamkit_native_abi.h"
#include "../../include/skw_engine.h"
#include "skw_segmenter.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
CResult ok_result() { CResult r; r.success = true; r.error_message = nullptr; return r; }
CResult err_result(const std::string& m) { g_err = m; CResult r; r.success = false; r.error_message = g_err.c_str(); return r; }
CResult err_null() { CResult r; r.success = false; r.error_message = nullptr; return r; }

struct Resampler {
    uint32_t target = 0; size_t chunk_frames = 960, out_frame = 960; int gpu_device = 0;
    bool init = false, needs = false; uint32_t rate = 0; uint16_t channels = 0;
    skw_dsp* dsp = nullptr; skw_resampler_state st{};
    std::vector<float> sample_buffer, output_buffer, scratch;
    // mode "polyphase" (additive): streaming form of skw_resample_polyphase, state on the device
    bool polyphase = false; int L = 1, M = 1, T = 32; skw_pp_stream* pp = nullptr;
    CLogCallback log_cb = nullptr; void* log_ud = nullptr;
    ~Resampler() { if (pp) skw_polyphase_stream_free(pp); if (dsp) skw_dsp_free(dsp); }
};

bool emit(Resampler* r, COutputCallback cb, void* ud, const float* d, size_t n, std::string* err) {
    CAudioFrame fr; fr.sample_rate = r->target; fr.channels = r->channels; fr.samples = d; fr.sample_count = n;
    CPacket pk; pk.packet_type = SK_PACKET_RAW_AUDIO; pk.data = &fr; pk.len = sizeof(CAudioFrame);
    CResult res = cb("out", &pk, ud);
    if (res.success) return true; *err = res.error_message ? res.error_message : "Unknown error"; return false;
}
bool drain(Resampler* r, COutputCallback cb, void* ud, std::string* err) {
    const size_t fs = r->out_frame * r->channels; size_t off = 0;
    while (r->output_buffer.size() - off >= fs) { if (!emit(r, cb, ud, r->output_buffer.data() + off, fs, err)) return false; off += fs; }
    r->output_buffer.erase(r->output_buffer.begin(), r->output_buffer.begin() + off); return true;
}
bool run_chunks(Resampler* r, skw_resampler_state* st, const float* in, int n_chunks, std::vector<float>* out, std::string* err) {
    const double ratio = st->ratio; const int cap = (int)((double)n_chunks * st->chunk_frames * ratio) + 64;
    out->resize((size_t)cap * r->channels); int n = 0;
    if (skw_resample_linear(r->dsp, st, in, n_chunks, out->data(), cap, &n) != 0) { *err = std::string("Resampling failed: ") + skw_dsp_last_error(r->dsp); return false; }
    out->resize((size_t)n * r->channels); return true;
}

// streaming polyphase: every output whose filter support has arrived (all of them at end of stream); the input tail that later
// outputs need stays in HBM (skw_polyphase_stream_*), a packet uploads only its own frames
bool polyphase_step(Resampler* r, const float* in, long n_frames, bool final_call, std::vector<float>* out, std::string* err) {
    out->clear();
    if (!r->pp) return true;
    const long cap = (n_frames + 2L * r->T + 8) * r->L / r->M + 64 + (final_call ? (long)r->T * r->L / r->M + 64 : 0);
    out->resize((size_t)cap * r->channels); long got = 0;
    if (skw_polyphase_stream_push(r->pp, in, n_frames, final_call ? 1 : 0, out->data(), cap, &got) != 0) { *err = std::string("Resampling failed: ") + skw_dsp_last_error(r->dsp); return false; }
    out->resize((size_t)got * r->channels); return true;
}

const char* const kSchema =
    "{\"type\":\"object\",\"required\":[\"target_sample_rate\"],\"properties\":{"
    "\"target_sample_rate\":{\"type\":\"integer\",\"minimum\":1,\"description\":\"Target output sample rate in Hz\"},"
    "\"chunk_frames\":{\"type\":\"integer\",\"minimum\":1,\"default\":960,\"description\":\"Fixed chunk size for resampler\"},"
    "\"output_frame_size\":{\"type\":\"integer\",\"default\":960,\"description\":\"Output frame size (0 disables re-chunking); must be 0 or a valid Opus frame size\"},"
    "\"mode\":{\"type\":\"string\",\"default\":\"linear\",\"description\":\"(additive) linear = the built-in node's rubato interpolation, bit for bit; polyphase = Kaiser-windowed sinc, 32 taps per phase of the slower rate\"},"
    "\"gpu_device\":{\"type\":\"integer\",\"default\":0,\"minimum\":0,\"maximum\":7,\"description\":\"(additive) GPU that runs the kernels\"}}}";
const CAudioFormat kAnyF32 = {0, 0, SK_SAMPLE_F32};
const CPacketTypeInfo kIn[1] = {{SK_PACKET_RAW_AUDIO, &kAnyF32, nullptr}};
const CInputPin kInputs[1] = {{"in", kIn, 1}};
// the produced rate depends on params; pins are read once at load (static), so the wildcard format is declared (packet_meta.rs:97-109: 0 matches any)
const COutputPin kOutputs[1] = {{"out", {SK_PACKET_RAW_AUDIO, &kAnyF32, nullptr}}};
const char* const kCats[2] = {"audio", "filters"};
const CNodeMetadata kMeta = {"resampler", "Resamples raw audio to a target sample rate on the GPU (linear interpolation, bit-exact with the built-in audio::resampler).",
                             kInputs, 1, kOutputs, 1, kSchema, kCats, 2};

const CNodeMetadata* get_metadata() { return &kMeta; }

CPluginHandle create_instance(const char* params, CLogCallback log_cb, void* log_ud) {
    auto r = std::unique_ptr<Resampler>(new Resampler()); r->log_cb = log_cb; r->log_ud = log_ud;
    auto fail = [&](const std::string& m) -> CPluginHandle { if (log_cb) log_cb(SK_LOG_ERROR, "resampler_plugin_native", m.c_str(), log_ud); return nullptr; };
    if (!params || !*params) return fail("target_sample_rate is required");
    skw::JsonValue v; std::string perr;
    if (!skw::json_parse(params, &v, &perr) || v.type != skw::JsonValue::Object) return fail("Invalid config: " + perr);
    auto num = [&](const char* k, double def) { const skw::JsonValue* x = v.get(k); return (x && x->type == skw::JsonValue::Number) ? x->num : def; };
    const double t = num("target_sample_rate", 0), cf = num("chunk_frames", 960), of = num("output_frame_size", 960);
    if (t < 1) return fail("target_sample_rate must be greater than 0");
    if (cf < 1) return fail("chunk_frames must be greater than 0");
    if (of != 0) { const double okv[] = {120, 240, 480, 960, 1920, 2880}; bool okf = false; for (double x : okv) okf = okf || x == of; if (!okf) return fail("output_frame_size must be 0 (disabled) or a valid Opus frame size: [120, 240, 480, 960, 1920, 2880]"); }
    r->target = (uint32_t)t; r->chunk_frames = (size_t)cf; r->out_frame = (size_t)of; r->gpu_device = (int)num("gpu_device", 0);
    { const skw::JsonValue* x = v.get("mode"); if (x) { if (x->type != skw::JsonValue::String || (x->str != "linear" && x->str != "polyphase")) return fail("mode must be \"linear\" or \"polyphase\""); r->polyphase = x->str == "polyphase"; } }
    return (CPluginHandle)r.release();
}

CResult process_packet(CPluginHandle h, const char* pin, const CPacket* pk, COutputCallback cb, void* ud, CTelemetryCallback, void*) {
    if (!h || !pin || !pk) return err_null();
    Resampler* r = (Resampler*)h; std::string err;
    if (!pk->data) return err_result("Invalid packet: Null packet data pointer");
    if (pk->packet_type != SK_PACKET_RAW_AUDIO) {   // non-audio packets pass through unchanged (resampler.rs:529-538)
        CResult res = cb("out", pk, ud); if (!res.success) return err_result(res.error_message ? res.error_message : "Unknown error"); return ok_result();
    }
    const CAudioFrame* fr = (const CAudioFrame*)pk->data;
    if (!fr->samples) return err_result("Invalid packet: Null samples pointer in audio frame");
    if (!r->init) {
        r->init = true; r->needs = fr->sample_rate != r->target; r->rate = fr->sample_rate; r->channels = fr->channels;
        if (r->needs) {
            if (fr->channels < 1 || fr->channels > 2) return err_result("Failed to create resampler: only mono and stereo are supported on the GPU path");
            char eb[512] = {0}; r->dsp = skw_dsp_create(r->gpu_device, eb, sizeof eb);
            if (!r->dsp) return err_result(std::string("Failed to create resampler: ") + eb);
            skw_resampler_init(&r->st, (double)r->target / (double)r->rate, (int)r->chunk_frames, r->channels);
            { long a = r->rate, b = r->target; while (b) { long t2 = a % b; a = b; b = t2; } r->L = (int)(r->target / a); r->M = (int)(r->rate / a); r->T = 32 * std::max(1, (r->M + r->L - 1) / r->L); }
            if (r->polyphase) { r->pp = skw_polyphase_stream_create(r->dsp, r->channels, (int)r->rate, (int)r->target); if (!r->pp) return err_result(std::string("Failed to create resampler: ") + skw_dsp_last_error(r->dsp)); }
        }
    }
    if (fr->sample_rate != r->rate || fr->channels != r->channels) {
        char b[200]; snprintf(b, sizeof b, "Audio format changed mid-stream: expected %uHz/%uch, got %uHz/%uch", r->rate, r->channels, fr->sample_rate, fr->channels); return err_result(b);
    }
    if (!r->needs) {
        if (r->out_frame == 0) { if (!emit(r, cb, ud, fr->samples, fr->sample_count, &err)) return err_result(err); return ok_result(); }
        r->output_buffer.insert(r->output_buffer.end(), fr->samples, fr->samples + fr->sample_count);
        if (!drain(r, cb, ud, &err)) return err_result(err); return ok_result();
    }
    if (r->polyphase) {
        if (!polyphase_step(r, fr->samples, (long)(fr->sample_count / r->channels), false, &r->scratch, &err)) return err_result(err);
        if (r->scratch.empty()) return ok_result();
        if (r->out_frame > 0) { r->output_buffer.insert(r->output_buffer.end(), r->scratch.begin(), r->scratch.end()); if (!drain(r, cb, ud, &err)) return err_result(err); }
        else if (!emit(r, cb, ud, r->scratch.data(), r->scratch.size(), &err)) return err_result(err);
        return ok_result();
    }
    r->sample_buffer.insert(r->sample_buffer.end(), fr->samples, fr->samples + fr->sample_count);
    const size_t cs = r->chunk_frames * r->channels; const int n_chunks = (int)(r->sample_buffer.size() / cs);
    if (n_chunks > 0) {
        if (!run_chunks(r, &r->st, r->sample_buffer.data(), n_chunks, &r->scratch, &err)) return err_result(err);
        r->sample_buffer.erase(r->sample_buffer.begin(), r->sample_buffer.begin() + (size_t)n_chunks * cs);
        if (r->out_frame > 0) { r->output_buffer.insert(r->output_buffer.end(), r->scratch.begin(), r->scratch.end()); if (!drain(r, cb, ud, &err)) return err_result(err); }
        else if (!r->scratch.empty()) {
            // without re-chunking the reference emits one packet per processed chunk; chunk boundaries are recovered from the frame counts
            if (!emit(r, cb, ud, r->scratch.data(), r->scratch.size(), &err)) return err_result(err);
        }
    }
    return ok_result();
}

CResult update_params(CPluginHandle h, const char*) { if (!h) return err_result("Invalid handle (null)"); return ok_result(); }

CResult flush(CPluginHandle h, COutputCallback cb, void* ud, CTelemetryCallback, void*) {
    if (!h) return err_result("Invalid handle (null)");
    Resampler* r = (Resampler*)h; std::string err;
    if (r->needs && r->polyphase) {
        if (!polyphase_step(r, nullptr, 0, true, &r->scratch, &err)) return err_result(err);
        if (!r->scratch.empty()) {
            if (r->out_frame > 0) { r->output_buffer.insert(r->output_buffer.end(), r->scratch.begin(), r->scratch.end()); if (!drain(r, cb, ud, &err)) return err_result(err); }
            else if (!emit(r, cb, ud, r->scratch.data(), r->scratch.size(), &err)) return err_result(err);
        }
    }
    if (r->needs && !r->polyphase && !r->sample_buffer.empty()) {
        const size_t rem = r->sample_buffer.size() / r->channels;
        if (rem >= 1) {   // fresh resampler sized to the remainder: zero history, last_index = -4 (resampler.rs:564-570)
            skw_resampler_state t; skw_resampler_init(&t, (double)r->target / (double)r->rate, (int)rem, r->channels);
            if (!run_chunks(r, &t, r->sample_buffer.data(), 1, &r->scratch, &err)) return err_result(err);
            if (r->out_frame > 0) { r->output_buffer.insert(r->output_buffer.end(), r->scratch.begin(), r->scratch.end()); if (!drain(r, cb, ud, &err)) return err_result(err); }
            else if (!r->scratch.empty() && !emit(r, cb, ud, r->scratch.data(), r->scratch.size(), &err)) return err_result(err);
        }
        r->sample_buffer.clear();
    }
    if (!r->output_buffer.empty() && r->out_frame > 0) { if (!emit(r, cb, ud, r->output_buffer.data(), r->output_buffer.size(), &err)) return err_result(err); r->output_buffer.clear(); }
    return ok_result();
}
void destroy_instance(CPluginHandle h) { if (h) delete (Resampler*)h; }
const CNativePluginAPI kApi = {STREAMKIT_NATIVE_PLUGIN_API_VERSION, get_metadata, create_instance, process_packet, update_params, flush, destroy_instance};
}  // namespace

extern "C" const CNativePluginAPI* streamkit_native_plugin_api(void) { return &kApi; }
