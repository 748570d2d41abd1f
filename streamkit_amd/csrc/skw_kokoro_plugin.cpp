// skw_kokoro_plugin.cpp — libkokoro.so: the StreamKit native plugin `kokoro` (registered by the host as plugin::native::kokoro) on top of
// the MI355X synthesiser (include/skw_tts.h).  Drop-in for the reference cdylib, restated function by function:
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:178-256  metadata (kind, description, pins, param schema, categories)
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:258-441  new: config, model_dir canonicalisation, engine cache keyed (dir, threads, provider)
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:444-492  process: Text | Binary -> sanitize -> '.' -> accumulate -> sentences -> generate
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:494-541  update_params (speaker_id, speed only), flush (the buffered rest is spoken)
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:560-652  generate_and_send: one 24 kHz mono f32 frame per sentence, tts.start / tts.done telemetry
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:734-829  create_tts_engine: the files a model_dir must hold, error strings
//   /root/reference/plugins/native/kokoro/src/config.rs               KokoroTtsConfig defaults
// The text front end is skw_kokoro_text.h; the synthesiser replaces the sherpa-onnx calls of ffi.rs:119-137.
// ABI: include/streamkit_native_abi.h.  Additive param: gpu_device.  `execution_provider` and `num_threads` keep their place in the config, the
// telemetry and the engine-cache key; this build always runs on the MI355X named by gpu_device.
#include "../../include/streamkit_native_abi.h"
#include "../../include/skw_tts.h"
#include "skw_kokoro_text.h"
#include <limits.h>
#include <stdlib.h>
#include <unistd.h>
#include <errno.h>
#include <sys/stat.h>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;      // conversions.rs:441-461: thread-local, borrowed until the next error on this thread
CResult ok_result() { CResult r; r.success = true; r.error_message = nullptr; return r; }
CResult err_result(const std::string& msg) { g_last_error = msg; for (auto& ch : g_last_error) if (ch == '\0') ch = ' '; CResult r;
r.success = false; r.error_message = g_last_error.c_str(); return r; }
CResult err_null() { CResult r; r.success = false; r.error_message = nullptr; return r; }
template <typename F> CResult guarded(F&& f) {      // nothing unwinds into the host's Rust frames
    try { return f(); }
    catch (const std::exception& e) { return err_result(std::string("Kokoro plugin: ") + e.what()); }
    catch (...) { return err_result("Kokoro plugin: unknown C++ exception"); }
}

// ------------------------------------------------------------------ config.rs
struct KokoroTtsConfig {
    std::string model_dir = "models/kokoro-multi-lang-v1_1";
    int32_t speaker_id = 50; float speed = 1.0f; int32_t num_threads = 4; size_t min_sentence_length = 10;
    std::string execution_provider = "cpu"; bool emit_telemetry = false; size_t telemetry_preview_chars = 80;
    int gpu_device = 0;      // additive
};
std::string default_execution_provider() { const char* e = getenv("KOKORO_EXECUTION_PROVIDER"); return e ? e : "cpu"; }      // config.rs:55-58

// serde_json::from_value::<KokoroTtsConfig>: model_dir is the one field without a default
bool parse_config(const skw::JsonValue& v, KokoroTtsConfig* c, std::string* err) {
    if (v.type != skw::JsonValue::Object) { *err = "Config parse error: invalid type: expected struct KokoroTtsConfig"; return false; }
    c->execution_provider = default_execution_provider();
    const skw::JsonValue* md = v.get("model_dir");
    if (!md) { *err = "Config parse error: missing field `model_dir`"; return false; }
    if (md->type != skw::JsonValue::String) { *err = "Config parse error: invalid type for `model_dir`, expected a string"; return false; }
    c->model_dir = md->str;
    auto integer = [&](const char* k, double lo, double hi, double* dst) {
        const skw::JsonValue* x = v.get(k); if (!x) return true;
        if (x->type != skw::JsonValue::Number || x->num != std::floor(x->num) || x->num < lo
            || x->num > hi) { *err = std::string("Config parse error: invalid value for `") + k + "`, expected an integer";
        return false; }
        *dst = x->num; return true;
    };
    double d;
    d = c->speaker_id; if (!integer("speaker_id", -2147483648.0, 2147483647.0, &d)) return false; c->speaker_id = (int32_t)d;
    d = c->num_threads; if (!integer("num_threads", -2147483648.0, 2147483647.0, &d)) return false; c->num_threads = (int32_t)d;
    d = (double)c->min_sentence_length; if (!integer("min_sentence_length", 0.0, 9007199254740992.0, &d)) return false; c->min_sentence_length = (size_t)d;
    d = (double)c->telemetry_preview_chars; if (!integer("telemetry_preview_chars", 0.0, 9007199254740992.0, &d)) return false; c->telemetry_preview_chars = (size_t)d;
    d = c->gpu_device; if (!integer("gpu_device", 0.0, 1024.0, &d)) return false; c->gpu_device = (int)d;
    if (const skw::JsonValue* x = v.get("speed")) { if (x->type != skw::JsonValue::Number) { *err = "Config parse error: invalid type for `speed`, expected a number";
    return false; } c->speed = (float)x->num; }
    if (const skw::JsonValue* x = v.get("execution_provider")) { if (x->type != skw::JsonValue::String) { *err = "Config parse error: invalid type for `execution_provider`, expected a string";
    return false; } c->execution_provider = x->str; }
    if (const skw::JsonValue* x = v.get("emit_telemetry")) { if (x->type != skw::JsonValue::Bool) { *err = "Config parse error: invalid type for `emit_telemetry`, expected a boolean";
    return false; } c->emit_telemetry = x->b; }
    return true;
}

// ------------------------------------------------------------------ engine cache (kokoro_node.rs:147-170): key (canonical model_dir, num_threads, provider) + the additive device
struct Engine { skw_tts* tts = nullptr; ~Engine() { if (tts) skw_tts_destroy(tts); } };
std::mutex g_cache_mu; std::map<std::string, std::shared_ptr<Engine>> g_cache;      // strong references, as the reference's cache holds Arcs for the life of the process

bool file_exists(const std::string& p) { struct stat sb; return stat(p.c_str(), &sb) == 0; }

// create_tts_engine (kokoro_node.rs:734-829)
std::shared_ptr<Engine> create_engine(const std::string& dir, const KokoroTtsConfig& cfg, std::string* err) {
    const std::string model = dir + "/model.onnx", voices = dir + "/voices.bin", tokens = dir + "/tokens.txt";
    const std::pair<const char*, const std::string*> need[3] = {{"model", &model}, {"voices", &voices}, {"tokens", &tokens}};
    for (auto& n : need) if (!file_exists(*n.second)) { *err = std::string(n.first) + " file not found: " + *n.second; return nullptr; }
    const std::string lexicon = dir + "/lexicon-us-en.txt," + dir + "/lexicon-zh.txt";      // kokoro_node.rs:746-747, 766
    skw_tts_config tc; tc.model = model.c_str(); tc.voices = voices.c_str(); tc.tokens = tokens.c_str(); tc.lexicon = lexicon.c_str(); tc.length_scale = 1.0f; tc.gpu_device = cfg.gpu_device;
    char ebuf[512] = {0};
    auto e = std::make_shared<Engine>();
    e->tts = skw_tts_create(&tc, ebuf, sizeof ebuf);
    if (!e->tts) { *err = std::string("Failed to create TTS engine") + (ebuf[0] ? std::string(": ") + ebuf : std::string()); return nullptr; }
    return e;
}

struct KokoroTtsNode {
    std::shared_ptr<Engine> engine; KokoroTtsConfig config; std::string text_buffer; skw::kokoro::SentenceSplitter splitter;
    CLogCallback log_cb = nullptr; void* log_ud = nullptr;
    void log(CLogLevel lv, const char* fmt, ...) { if (!log_cb) return; char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap); log_cb(lv, "kokoro::kokoro_node", buf, log_ud); }
};

// KokoroTtsNode::new (kokoro_node.rs:258-441)
KokoroTtsNode* node_new(const char* params, CLogCallback log_cb, void* log_ud, std::string* err) {
    auto n = std::unique_ptr<KokoroTtsNode>(new KokoroTtsNode()); n->log_cb = log_cb; n->log_ud = log_ud;
    if (params && *params) {
        skw::JsonValue v; std::string perr;
        if (!skw::json_parse(params, &v, &perr)) { *err = "Invalid params JSON: " + perr; return nullptr; }      // the SDK returns NULL here without a message (sdk lib.rs:703-706)
        if (!parse_config(v, &n->config, err)) return nullptr;
    }                                                                                                              // params None -> KokoroTtsConfig::default()
    // model_dir: made absolute against the working directory, then canonicalised (kokoro_node.rs:287-306)
    char real[PATH_MAX];
    if (!realpath(n->config.model_dir.c_str(), real)) {
        std::string shown = n->config.model_dir;
        if (!shown.empty() && shown[0] != '/') { char cwd[PATH_MAX]; if (getcwd(cwd, sizeof cwd)) shown = std::string(cwd) + "/" + shown; }
        *err = "Failed to canonicalize model dir '" + shown + "': " + strerror(errno); return nullptr;
    }
    const std::string dir = real;
    // normalize_execution_provider (kokoro_node.rs:90-112) guards the cache key against sherpa-onnx silently falling back to CPU; here every provider
    // name runs the same HIP engine, so the requested name is kept as the reference keeps it for "cpu" and for unknown names
    char key[64]; snprintf(key, sizeof key, "|%d|%d|", n->config.num_threads, n->config.gpu_device);
    const std::string cache_key = dir + key + n->config.execution_provider;
    {
        std::lock_guard<std::mutex> l(g_cache_mu);
        auto it = g_cache.find(cache_key);
        if (it != g_cache.end()) { n->engine = it->second; n->log(SK_LOG_INFO, "CACHE HIT: Reusing cached TTS engine (%s)", dir.c_str()); }
        else {
            n->log(SK_LOG_WARN, "CACHE MISS: Creating new TTS engine (%s)", dir.c_str());
            n->engine = create_engine(dir, n->config, err);
            if (!n->engine) return nullptr;
            g_cache[cache_key] = n->engine;
        }
    }
    n->splitter = skw::kokoro::SentenceSplitter(n->config.min_sentence_length);
    return n.release();
}

struct Out { COutputCallback out_cb; void* out_ud; CTelemetryCallback tel_cb; void* tel_ud;
             void telemetry(const char* ev, const std::string& json) const { if (tel_cb) (void)tel_cb(ev, (const uint8_t*)json.data(), json.size(), nullptr, tel_ud); } };

// serde_json::json!({...}) serialises its keys in alphabetical order (serde_json's default map is a BTreeMap; the crate does not enable preserve_order)
std::string telemetry_json(const KokoroTtsNode* n, const std::string& text, bool done, size_t samples, long long duration_ms, long long latency_ms) {
    std::string prev; const bool has_prev = skw::kokoro::text_preview(text, n->config.telemetry_preview_chars, &prev);
    std::string j = "{";
    if (done) j += "\"audio_duration_ms\":" + std::to_string(duration_ms) + ",\"audio_samples\":" + std::to_string(samples) + ",";
    j += "\"execution_provider\":" + skw::json_quote(n->config.execution_provider) + ",";
    if (done) j += "\"latency_ms\":" + std::to_string(latency_ms) + ",";
    j += "\"speaker_id\":" + std::to_string(n->config.speaker_id) + ",\"speed\":" + skw::json_f32(n->config.speed) + ",\"text_length\":" + std::to_string(text.size()) +
         ",\"text_preview\":" + (has_prev ? skw::json_quote(prev) : std::string("null")) + "}";
    return j;
}

// generate_and_send (kokoro_node.rs:560-652)
bool generate_and_send(KokoroTtsNode* n, const std::string& text, const Out& out, std::string* err) {
    const auto start = std::chrono::steady_clock::now();
    if (n->config.emit_telemetry) out.telemetry("tts.start", telemetry_json(n, text, false, 0, 0, 0));
    if (text.find('\0') != std::string::npos) { *err = "Invalid text: nul byte found in provided data"; return false; }      // CString::new
    const skw_tts_audio* audio = skw_tts_generate(n->engine->tts, text.c_str(), n->config.speaker_id, n->config.speed);
    if (!audio) { n->log(SK_LOG_ERROR, "TTS generation returned null pointer: %s", skw_tts_last_error(n->engine->tts)); *err = "TTS generation failed"; return false; }
    if (!audio->samples || audio->n <= 0) { skw_tts_destroy_audio(audio); *err = "TTS generated empty audio"; return false; }
    const size_t sample_count = (size_t)audio->n;
    CAudioFrame fr; fr.sample_rate = 24000; fr.channels = 1; fr.samples = audio->samples; fr.sample_count = sample_count;      // AudioFrame::new(24000, 1, samples)
    CPacket pk; pk.packet_type = SK_PACKET_RAW_AUDIO; pk.data = &fr; pk.len = sizeof(CAudioFrame);
    CResult r = out.out_cb("out", &pk, out.out_ud);
    if (!r.success) { *err = std::string("Failed to send audio: ") + (r.error_message ? r.error_message : "Unknown error"); skw_tts_destroy_audio(audio); return false; }
    if (n->config.emit_telemetry) {
        const long long latency = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - start).count();
        const unsigned long long dur = ((unsigned long long)sample_count * 1000ull + 12000ull) / 24000ull;      // rounded to the nearest millisecond
        out.telemetry("tts.done", telemetry_json(n, text, true, sample_count, (long long)dur, latency));
    }
    skw_tts_destroy_audio(audio);
    return true;
}

// ------------------------------------------------------------------ metadata (kokoro_node.rs:178-256)
const char* const kDescription =
    "High-quality text-to-speech synthesis using the Kokoro TTS model. Supports 103 voices across Chinese and English with streaming output. "
    "Outputs 24kHz mono audio for real-time playback or further processing.";
const char* const kSchema =
    "{\"type\":\"object\",\"properties\":{"
    "\"model_dir\":{\"type\":\"string\",\"description\":\"Path to Kokoro model directory\",\"default\":\"./models/kokoro-multi-lang-v1_1\"},"
    "\"speaker_id\":{\"type\":\"integer\",\"description\":\"Voice ID (0-102 for v1.1)\",\"default\":50,\"minimum\":0,\"maximum\":102},"
    "\"speed\":{\"type\":\"number\",\"description\":\"Speech speed multiplier\",\"default\":1.0,\"minimum\":0.5,\"maximum\":2.0},"
    "\"num_threads\":{\"type\":\"integer\",\"description\":\"CPU threads for inference\",\"default\":4,\"minimum\":1,\"maximum\":16},"
    "\"min_sentence_length\":{\"type\":\"integer\",\"description\":\"Minimum chars before TTS generation\",\"default\":10,\"minimum\":1},"
    "\"execution_provider\":{\"type\":\"string\",\"description\":\"ONNX Runtime execution provider (requires libsherpa-onnx built with GPU support)\",\"default\":\"cpu\",\"enum\":[\"cpu\",\"cuda\",\"tensorrt\"]},"
    "\"emit_telemetry\":{\"type\":\"boolean\",\"description\":\"Emit out-of-band telemetry events (tts.start/tts.done) to the session telemetry bus\",\"default\":false},"
    "\"telemetry_preview_chars\":{\"type\":\"integer\",\"description\":\"Maximum characters of text preview to include in telemetry events (0 = omit preview)\",\"default\":80,\"minimum\":0,\"maximum\":1000},"
    "\"gpu_device\":{\"type\":\"integer\",\"description\":\"(additive) MI355X that runs the synthesiser\",\"default\":0,\"minimum\":0,\"maximum\":7}"
    "},\"required\":[\"model_dir\"]}";
const CPacketTypeInfo kInTypes[1] = {{SK_PACKET_TEXT, nullptr, nullptr}};
const CInputPin kInputs[1] = {{"in", kInTypes, 1}};
const CAudioFormat kOutFormat = {24000, 1, SK_SAMPLE_F32};
const COutputPin kOutputs[1] = {{"out", {SK_PACKET_RAW_AUDIO, &kOutFormat, nullptr}}};
const char* const kCategories[2] = {"audio", "tts"};
const CNodeMetadata kMetadata = {"kokoro", kDescription, kInputs, 1, kOutputs, 1, kSchema, kCategories, 2};

// ------------------------------------------------------------------ the six entry points (sdk lib.rs:462-854)
const CNodeMetadata* plugin_get_metadata() { return &kMetadata; }

CPluginHandle plugin_create_instance(const char* params, CLogCallback log_cb, void* log_ud) {
    auto fail = [&](const std::string& m) -> CPluginHandle { if (log_cb) log_cb(SK_LOG_ERROR, "kokoro::kokoro_node", m.c_str(), log_ud); return nullptr; };
    try { std::string err; KokoroTtsNode* n = node_new(params, log_cb, log_ud, &err); if (!n) return fail(err); return (CPluginHandle)n; }
    catch (const std::exception& e) { return fail(std::string("Kokoro plugin: ") + e.what()); }
    catch (...) { return fail("Kokoro plugin: unknown C++ exception"); }
}

CResult plugin_process_packet(CPluginHandle handle, const char* input_pin, const CPacket* packet, COutputCallback out_cb, void* out_ud, CTelemetryCallback tel_cb, void* tel_ud) {
    if (!handle || !input_pin || !packet) return err_null();
    return guarded([&]() -> CResult {
        KokoroTtsNode* self = (KokoroTtsNode*)handle;
        std::string text;
        if (packet->packet_type == SK_PACKET_TEXT) {            // packet_from_c: NUL-terminated UTF-8 (conversions.rs:349-354)
            if (!packet->data) return err_result("Invalid packet: Null packet data pointer");
            text = (const char*)packet->data;
            if (!skw::utf8_valid(text)) return err_result("Invalid packet: Invalid UTF-8 in text packet");
        } else if (packet->packet_type == SK_PACKET_BINARY) {   // kokoro_node.rs:449-452
            if (!packet->data && packet->len) return err_result("Invalid packet: Null packet data pointer");
            text.assign((const char*)packet->data, packet->len);
            if (!skw::utf8_valid(text)) return err_result("Failed to decode binary data as UTF-8: invalid utf-8 sequence");
        } else return err_result("Only accepts Text or Binary packets");
        std::string sanitized = skw::kokoro::sanitize_text(text);
        if (sanitized.empty()) return ok_result();
        if (!skw::kokoro::ends_with_final_punct(sanitized)) sanitized.push_back('.');
        self->text_buffer += sanitized;
        Out out{out_cb, out_ud, tel_cb, tel_ud}; std::string sentence, err;
        while (self->splitter.extract_sentence(&self->text_buffer, &sentence)) if (!generate_and_send(self, sentence, out, &err)) return err_result(err);
        return ok_result();
    });
}

// kokoro_node.rs:494-506: the whole config is parsed again (so model_dir is required here too); only speaker_id and speed are taken over
CResult plugin_update_params(CPluginHandle handle, const char* params) {
    if (!handle) return err_result("Invalid handle (null)");
    return guarded([&]() -> CResult {
        KokoroTtsNode* self = (KokoroTtsNode*)handle;
        if (!params || !*params) return ok_result();
        skw::JsonValue v; std::string perr;
        if (!skw::json_parse(params, &v, &perr)) return err_result("Invalid params JSON: " + perr);
        KokoroTtsConfig nc; std::string err;
        if (!parse_config(v, &nc, &err)) return err_result(err);
        self->config.speaker_id = nc.speaker_id; self->config.speed = nc.speed;
        return ok_result();
    });
}

// kokoro_node.rs:508-533: whatever is still buffered is spoken as it stands
CResult plugin_flush(CPluginHandle handle, COutputCallback out_cb, void* out_ud, CTelemetryCallback tel_cb, void* tel_ud) {
    if (!handle) return err_result("Invalid handle (null)");
    return guarded([&]() -> CResult {
        KokoroTtsNode* self = (KokoroTtsNode*)handle;
        if (self->text_buffer.empty()) return ok_result();
        Out out{out_cb, out_ud, tel_cb, tel_ud}; std::string err;
        const std::string text = self->text_buffer;
        if (!generate_and_send(self, text, out, &err)) return err_result(err);
        self->text_buffer.clear();
        return ok_result();
    });
}

void plugin_destroy_instance(CPluginHandle handle) { try { if (handle) delete (KokoroTtsNode*)handle; } catch (...) {} }

const CNativePluginAPI kApi = {STREAMKIT_NATIVE_PLUGIN_API_VERSION, plugin_get_metadata, plugin_create_instance, plugin_process_packet, plugin_update_params, plugin_flush, plugin_destroy_instance};
}  // namespace

extern "C" const CNativePluginAPI* streamkit_native_plugin_api(void) { return &kApi; }
