// skw_resampler_plugin.cpp — libresampler.so: native plugin `resampler` (registered as plugin::native::resampler), the
// GPU drop-in for the reference's built-in audio::resampler node where a pipeline feeds the Whisper node from a
// non-16 kHz source (Opus decode always yields 48 kHz, crates/nodes/src/audio/codecs/opus.rs:70,103).
// The node's logic (R1-R4 of /root/reference/crates/nodes/src/audio/filters/resampler.rs:148-743) is skw_resampler_core.h; this file is the
// plugin shell around it: metadata, params, the six entry points, and the rule that no C++ exception leaves them.
// One line of YAML swaps it in: `kind: plugin::native::resampler` with the same params as audio::resampler.
#include "../../include/streamkit_native_abi.h"
#include "../../include/skw_engine.h"
#include "skw_segmenter.h"
#include "skw_resampler_core.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <exception>
#include <memory>
#include <string>
#include <vector>

namespace {
thread_local std::string g_err;
CResult ok_result() { CResult r; r.success = true; r.error_message = nullptr; return r; }
CResult err_result(const std::string& m) { g_err = m; CResult r; r.success = false; r.error_message = g_err.c_str(); return r; }
CResult err_null() { CResult r; r.success = false; r.error_message = nullptr; return r; }
// Nothing may unwind into the host's Rust frames (SURVEY.md section 8b "Errors"): every entry point body runs under this.
template <typename F> CResult guarded(F&& f) {
    try { return f(); }
    catch (const std::exception& e) { return err_result(std::string("resampler plugin: ") + e.what()); }
    catch (...) { return err_result("resampler plugin: unknown C++ exception"); }
}

struct Resampler { skw::ResamplerCore core; CLogCallback log_cb = nullptr; void* log_ud = nullptr; };

skw::ResamplerCore::Sink make_sink(Resampler* r, COutputCallback cb, void* ud) {
    return [r, cb, ud](const float* d, size_t n, std::string* err) {
        CAudioFrame fr; fr.sample_rate = r->core.target; fr.channels = r->core.channels; fr.samples = d; fr.sample_count = n;
        CPacket pk; pk.packet_type = SK_PACKET_RAW_AUDIO; pk.data = &fr; pk.len = sizeof(CAudioFrame);
        CResult res = cb("out", &pk, ud);
        if (res.success) return true; *err = res.error_message ? res.error_message : "Unknown error"; return false;
    };
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
    auto fail = [&](const std::string& m) -> CPluginHandle { if (log_cb) log_cb(SK_LOG_ERROR, "resampler_plugin_native", m.c_str(), log_ud); return nullptr; };
    try {
        auto r = std::unique_ptr<Resampler>(new Resampler()); r->log_cb = log_cb; r->log_ud = log_ud;
        if (!params || !*params) return fail("target_sample_rate is required");
        skw::JsonValue v; std::string perr;
        if (!skw::json_parse(params, &v, &perr) || v.type != skw::JsonValue::Object) return fail("Invalid config: " + perr);
        auto num = [&](const char* k, double def) { const skw::JsonValue* x = v.get(k); return (x && x->type == skw::JsonValue::Number) ? x->num : def; };
        const double t = num("target_sample_rate", 0), cf = num("chunk_frames", 960), of = num("output_frame_size", 960);
        if (t < 1) return fail("target_sample_rate must be greater than 0");
        if (cf < 1) return fail("chunk_frames must be greater than 0");
        if (of != 0) { const double okv[] = {120, 240, 480, 960, 1920, 2880};
        bool okf = false; for (double x : okv) okf = okf || x == of; if (!okf) return fail("output_frame_size must be 0 (disabled) or a valid Opus frame size: [120, 240, 480, 960, 1920, 2880]");
        }
        r->core.target = (uint32_t)t; r->core.chunk_frames = (size_t)cf; r->core.out_frame = (size_t)of; r->core.gpu_device = (int)num("gpu_device", 0);
        { const skw::JsonValue* x = v.get("mode"); if (x) { if (x->type != skw::JsonValue::String || (x->str != "linear"
            && x->str != "polyphase")) return fail("mode must be \"linear\" or \"polyphase\"");
        r->core.polyphase = x->str == "polyphase"; } }
        return (CPluginHandle)r.release();
    } catch (const std::exception& e) { return fail(std::string("resampler plugin: ") + e.what()); }
    catch (...) { return fail("resampler plugin: unknown C++ exception"); }
}

CResult process_packet(CPluginHandle h, const char* pin, const CPacket* pk, COutputCallback cb, void* ud, CTelemetryCallback, void*) {
    if (!h || !pin || !pk) return err_null();
    return guarded([&]() -> CResult {
        Resampler* r = (Resampler*)h; std::string err;
        if (!pk->data) return err_result("Invalid packet: Null packet data pointer");
        if (pk->packet_type != SK_PACKET_RAW_AUDIO) {   // non-audio packets pass through unchanged (resampler.rs:529-538)
            CResult res = cb("out", pk, ud); if (!res.success) return err_result(res.error_message ? res.error_message : "Unknown error"); return ok_result();
        }
        const CAudioFrame* fr = (const CAudioFrame*)pk->data;
        if (!fr->samples) return err_result("Invalid packet: Null samples pointer in audio frame");
        if (!r->core.push(fr->samples, fr->sample_count, fr->sample_rate, fr->channels, make_sink(r, cb, ud), &err)) return err_result(err);
        return ok_result();
    });
}

CResult update_params(CPluginHandle h, const char*) { if (!h) return err_result("Invalid handle (null)"); return ok_result(); }

CResult flush(CPluginHandle h, COutputCallback cb, void* ud, CTelemetryCallback, void*) {
    if (!h) return err_result("Invalid handle (null)");
    return guarded([&]() -> CResult {
        Resampler* r = (Resampler*)h; std::string err;
        if (!r->core.finish(make_sink(r, cb, ud), &err)) return err_result(err);
        return ok_result();
    });
}
void destroy_instance(CPluginHandle h) { try { if (h) delete (Resampler*)h; } catch (...) {} }
const CNativePluginAPI kApi = {STREAMKIT_NATIVE_PLUGIN_API_VERSION, get_metadata, create_instance, process_packet, update_params, flush, destroy_instance};
}  // namespace

extern "C" const CNativePluginAPI* streamkit_native_plugin_api(void) { return &kApi; }
