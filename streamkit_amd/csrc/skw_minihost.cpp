// skw_minihost.cpp — libskw_minihost.so: a C++ stand-in for the parts of the StreamKit host that sit on either side of
// the Whisper plugin, so the drop-in boundary can be exercised without the Rust engine (no cargo in this image):
//   * LoadedNativePlugin::load   /root/reference/crates/plugin-native/src/lib.rs:50-103   (dlopen, symbol, version, metadata)
//   * NativeNodeWrapper          /root/reference/crates/plugin-native/src/wrapper.rs:159-191, 314-465, 505-636
//       create_instance -> N x process_packet (each on a fresh OS thread, like tokio::task::spawn_blocking; never
//       concurrent for one instance) -> flush when the input closes -> destroy_instance; output / telemetry / log shims
//       copy everything inside the callback because the pointers are only valid during the call.
//   * audio::resampler node      /root/reference/crates/nodes/src/audio/filters/resampler.rs:148-743 (R1-R4)
//   * core::json_serialize       /root/reference/crates/nodes/src/core/json_serialize.rs:85-107 (mh_json_serialize: NDJSON of externally tagged packets)
// Plain C API so pytest can drive it through ctypes.
#include "../../include/streamkit_native_abi.h"
#include "skw_segmenter.h"
#include "skw_kokoro_text.h"
#include <dlfcn.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>

namespace {
std::string jq(const std::string& s) {
    std::string o = "\""; char buf[8];
    for (unsigned char c : s) { switch (c) { case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break;
    case '\r': o += "\\r"; break; case '\t': o += "\\t"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break;
        default: if (c < 0x20) { snprintf(buf, sizeof buf, "\\u%04x", c); o += buf; } else o += (char)c; } }
    return o + "\"";
}
}  // namespace

struct mh_plugin { void* lib = nullptr; const CNativePluginAPI* api = nullptr; std::string kind, meta_json; };
struct mh_output { std::string pin; int packet_type; std::string payload; uint32_t sample_rate = 0; uint16_t channels = 0; };
struct mh_telemetry { std::string event_type, json; };
struct mh_node {
    mh_plugin* plugin = nullptr; CPluginHandle handle = nullptr; std::vector<mh_output> outputs; std::vector<mh_telemetry> telemetry; std::vector<std::string> logs;
    std::string last_error, cb_error; bool failed = false;
};

extern "C" {

// LoadedNativePlugin::load
mh_plugin* mh_load(const char* path, char* err, size_t errlen) {
    void* lib = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { snprintf(err, errlen, "Failed to load library: %s", dlerror()); return nullptr; }
    typedef const CNativePluginAPI* (*getter_t)(void);
    getter_t get = (getter_t)dlsym(lib, STREAMKIT_PLUGIN_API_SYMBOL);
    if (!get) { snprintf(err, errlen, "Failed to find symbol %s", STREAMKIT_PLUGIN_API_SYMBOL); dlclose(lib); return nullptr; }
    const CNativePluginAPI* api = get();
    if (!api) { snprintf(err, errlen, "Plugin returned null API pointer"); dlclose(lib); return nullptr; }
    if (api->version != STREAMKIT_NATIVE_PLUGIN_API_VERSION) { snprintf(err, errlen, "Plugin API version mismatch: expected %u, got %u", STREAMKIT_NATIVE_PLUGIN_API_VERSION, api->version);
    dlclose(lib); return nullptr; }
    const CNodeMetadata* md = api->get_metadata();
    if (!md || !md->kind) { snprintf(err, errlen, "Plugin returned null metadata"); dlclose(lib); return nullptr; }
    mh_plugin* p = new mh_plugin(); p->lib = lib; p->api = api; p->kind = md->kind;
    if (p->kind.find("::") != std::string::npos) { snprintf(err, errlen, "plugin kind must not contain '::'"); delete p; dlclose(lib); return nullptr; }   // plugin-native lib.rs:307-333
    std::string j = "{\"kind\":" + jq(md->kind) + ",\"registered_as\":" + jq("plugin::native::" + p->kind);
    j += ",\"description\":" + (md->description ? jq(md->description) : std::string("null")) + ",\"inputs\":[";
    for (size_t i = 0; i < md->inputs_count; ++i) {
        if (i) j += ","; j += "{\"name\":" + jq(md->inputs[i].name) + ",\"accepts\":[";
        for (size_t k = 0; k < md->inputs[i].accepts_types_count; ++k) {
            const CPacketTypeInfo& t = md->inputs[i].accepts_types[k]; if (k) j += ",";
            j += "{\"type\":" + std::to_string((int)t.type_discriminant);
            if (t.audio_format) {
                j += ",\"sample_rate\":" + std::to_string(t.audio_format->sample_rate) + ",\"channels\":" + std::to_string(t.audio_format->channels);
                j += ",\"sample_format\":" + std::to_string((int)t.audio_format->sample_format);
            }
            j += "}";
        }
        j += "]}";
    }
    j += "],\"outputs\":[";
    for (size_t i = 0; i < md->outputs_count; ++i) { if (i) j += ",";
    j += "{\"name\":" + jq(md->outputs[i].name) + ",\"type\":" + std::to_string((int)md->outputs[i].produces_type.type_discriminant) + "}"; }
    j += "],\"categories\":[";
    for (size_t i = 0; i < md->categories_count; ++i) { if (i) j += ","; j += jq(md->categories[i]); }
    j += "],\"param_schema\":" + std::string(md->param_schema ? md->param_schema : "null") + "}";
    p->meta_json = j;
    return p;
}
const char* mh_metadata_json(mh_plugin* p) { return p->meta_json.c_str(); }
void mh_unload(mh_plugin* p) { if (!p) return; if (p->lib) dlclose(p->lib); delete p; }

static void log_shim(CLogLevel level, const char* target, const char* message, void* ud) {
    mh_node* n = (mh_node*)ud; if (!n) return;
    n->logs.push_back(std::to_string((int)level) + " " + (target ? target : "unknown") + ": " + (message ? message : ""));
}
// output_callback_shim (wrapper.rs:522-557): copy inside the callback
static CResult out_shim(const char* pin, const CPacket* pk, void* ud) {
    CResult r; r.error_message = nullptr;
    if (!pin || !pk || !ud) { r.success = false; return r; }
    mh_node* n = (mh_node*)ud;
    if (!pk->data) { n->cb_error = "Failed to convert packet: Null packet data pointer"; r.success = false; return r; }
    mh_output o; o.pin = pin; o.packet_type = (int)pk->packet_type;
    if (pk->packet_type == SK_PACKET_TRANSCRIPTION || pk->packet_type == SK_PACKET_BINARY) o.payload.assign((const char*)pk->data, pk->len);
    else if (pk->packet_type == SK_PACKET_TEXT) o.payload = (const char*)pk->data;
    else if (pk->packet_type == SK_PACKET_RAW_AUDIO) { const CAudioFrame* f = (const CAudioFrame*)pk->data;
    o.payload.assign((const char*)f->samples, f->sample_count * sizeof(float)); o.sample_rate = f->sample_rate; o.channels = f->channels; }
    else { n->cb_error = "Failed to convert packet: Unsupported packet type"; r.success = false; return r; }
    n->outputs.push_back(std::move(o)); r.success = true; return r;
}
// telemetry_callback_shim (wrapper.rs:563-636): best effort, always success
static CResult tel_shim(const char* event_type, const uint8_t* data, size_t len, const CPacketMetadata*, void* ud) {
    CResult r; r.success = true; r.error_message = nullptr; if (!event_type || !ud) return r;
    mh_node* n = (mh_node*)ud; n->telemetry.push_back(mh_telemetry{event_type, std::string((const char*)data, data ? len : 0)}); return r;
}

// NativeNodeWrapper::new (wrapper.rs:159-191)
mh_node* mh_create_node(mh_plugin* p, const char* params_json, char* err, size_t errlen) {
    mh_node* n = new mh_node(); n->plugin = p;
    n->handle = p->api->create_instance(params_json, log_shim, n);
    if (!n->handle) { std::string extra; for (auto& l : n->logs) extra += " | " + l; snprintf(err, errlen, "Plugin failed to create instance%s", extra.c_str()); delete n; return nullptr; }
    return n;
}
// one input packet (wrapper.rs:398-465): the FFI call runs on its own OS thread, the caller waits for it
int mh_process_audio(mh_node* n, const float* samples, size_t count, uint32_t sample_rate, uint16_t channels) {
    if (n->failed) return -2;
    CResult res; res.success = true; res.error_message = nullptr; std::string emsg;
    std::thread t([&] {
        CAudioFrame fr; fr.sample_rate = sample_rate; fr.channels = channels; fr.samples = samples; fr.sample_count = count;
        CPacket pk; pk.packet_type = SK_PACKET_RAW_AUDIO; pk.data = &fr; pk.len = sizeof(CAudioFrame);
        n->cb_error.clear();
        res = n->plugin->api->process_packet(n->handle, "in", &pk, out_shim, n, tel_shim, n);
        if (!res.success) emsg = res.error_message ? res.error_message : "Unknown plugin error";   // copied immediately (wrapper.rs:438-447)
        else if (!n->cb_error.empty()) emsg = n->cb_error;
    });
    t.join();
    if (!emsg.empty()) { n->last_error = emsg; n->failed = true; return -1; }   // node enters Failed (wrapper.rs:468-483)
    return 0;
}
int mh_process_text(mh_node* n, const char* text) {
    CPacket pk; pk.packet_type = SK_PACKET_TEXT; pk.data = text; pk.len = strlen(text) + 1;
    CResult res = n->plugin->api->process_packet(n->handle, "in", &pk, out_shim, n, tel_shim, n);
    if (!res.success) { n->last_error = res.error_message ? res.error_message : "Unknown plugin error"; return -1; } return 0;
}
int mh_process_binary(mh_node* n, const void* data, size_t len) {
    CPacket pk; pk.packet_type = SK_PACKET_BINARY; pk.data = data; pk.len = len;
    CResult res = n->plugin->api->process_packet(n->handle, "in", &pk, out_shim, n, tel_shim, n);
    if (!res.success) { n->last_error = res.error_message ? res.error_message : "Unknown plugin error"; return -1; } return 0;
}
int mh_output_audio_format(mh_node* n, size_t i, uint32_t* rate, uint16_t* channels) { if (i >= n->outputs.size()) return -1;
*rate = n->outputs[i].sample_rate; *channels = n->outputs[i].channels; return 0; }
// The Transcription -> Text step of the voice-agent pipelines (samples/pipelines/dynamic/voice-agent-openai.yaml:86-95, a core::script node): output i of
// `src` (a Transcription packet) becomes a Text packet into `dst`; an empty / missing text produces nothing (returns 1).  skw_kokoro_text.h.
int mh_forward_transcription_as_text(mh_node* src, size_t i, mh_node* dst) {
    if (i >= src->outputs.size() || src->outputs[i].packet_type != SK_PACKET_TRANSCRIPTION) { dst->last_error = "not a Transcription packet"; return -1; }
    std::string text; if (!skw::kokoro::transcription_to_text(src->outputs[i].payload, &text)) return 1;
    return mh_process_text(dst, text.c_str());
}
int mh_process_null(mh_node* n) { CResult res = n->plugin->api->process_packet(n->handle, nullptr, nullptr, out_shim, n, tel_shim, n);
if (!res.success) { n->last_error = res.error_message ? res.error_message : "(null message)"; return -1; } return 0; }
int mh_update_params(mh_node* n, const char* json) {
    CResult res; std::string emsg; std::thread t([&] { res = n->plugin->api->update_params(n->handle, json);
        if (!res.success) emsg = res.error_message ? res.error_message : "Failed to update parameters"; });
    t.join();
    if (!emsg.empty()) { n->last_error = emsg; return -1; } return 0;   // only logged by the host (wrapper.rs:297-299)
}
int mh_flush(mh_node* n) {
    CResult res; std::string emsg; std::thread t([&] { res = n->plugin->api->flush(n->handle, out_shim, n, tel_shim, n);
        if (!res.success) emsg = res.error_message ? res.error_message : "Plugin flush failed"; });
    t.join();
    if (!emsg.empty()) { n->last_error = emsg; return -1; } return 0;
}
// Oneshot batch driver (BASELINE config 2): one pipeline task (thread) per node feeds its clip in `packet`-sample RawAudio packets and
// flushes when the input closes, all nodes concurrently, as N http requests would.  process_packet runs on the node's own thread
// (tokio's spawn_blocking hands calls to pooled threads; what matters is one call at a time per instance).  Returns 0 when every
// call succeeded; wall_ms = first packet in -> last flush returned.
int mh_run_oneshot(mh_node** nodes, int n_nodes, const float* const* pcm, const size_t* n_samples, size_t packet, double* wall_ms) {
    std::vector<std::thread> th; std::vector<int> rc(n_nodes, 0);
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < n_nodes; ++i) th.emplace_back([&, i] {
        mh_node* n = nodes[i];
        for (size_t off = 0; off < n_samples[i] && rc[i] == 0; off += packet) {
            const size_t cnt = std::min(packet, n_samples[i] - off);
            CAudioFrame fr; fr.sample_rate = 16000; fr.channels = 1; fr.samples = pcm[i] + off; fr.sample_count = cnt;
            CPacket pk; pk.packet_type = SK_PACKET_RAW_AUDIO; pk.data = &fr; pk.len = sizeof(CAudioFrame);
            n->cb_error.clear();
            CResult res = n->plugin->api->process_packet(n->handle, "in", &pk, out_shim, n, tel_shim, n);
            if (!res.success) { n->last_error = res.error_message ? res.error_message : "Unknown plugin error"; n->failed = true; rc[i] = -1; }
            else if (!n->cb_error.empty()) { n->last_error = n->cb_error; n->failed = true; rc[i] = -1; }
        }
        if (rc[i] == 0) { CResult res = n->plugin->api->flush(n->handle, out_shim, n, tel_shim, n);
        if (!res.success) { n->last_error = res.error_message ? res.error_message : "Plugin flush failed"; rc[i] = -1; } }
    });
    for (auto& t : th) t.join();
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (wall_ms) *wall_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    for (int r : rc) if (r) return -1;
    return 0;
}
// Dynamic-session driver (BASELINE config 4): every node receives its stream in `packet`-sample packets paced at real time (packet k is
// due at t0 + k * pace_us); the duration of each process_packet call that produced output is that segment's latency (segment end ->
// transcript: the plugin emits from inside the call that closed the segment).  lat_ms: [n_nodes][max_lat], n_lat: [n_nodes].
int mh_run_paced(mh_node** nodes, int n_nodes, const float* const* pcm, const size_t* n_samples, size_t packet, long pace_us, double* lat_ms, int max_lat, int* n_lat, double* wall_ms) {
    std::vector<std::thread> th; std::vector<int> rc(n_nodes, 0);
    struct timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    auto now_us = [&]() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (t.tv_sec - t0.tv_sec) * 1000000L + (t.tv_nsec - t0.tv_nsec) / 1000L; };
    for (int i = 0; i < n_nodes; ++i) th.emplace_back([&, i] {
        mh_node* n = nodes[i]; n_lat[i] = 0; long k = 0;
        for (size_t off = 0; off < n_samples[i] && rc[i] == 0; off += packet, ++k) {
            const long due = k * pace_us, t = now_us();
            if (due > t) { struct timespec sl; sl.tv_sec = (due - t) / 1000000L; sl.tv_nsec = ((due - t) % 1000000L) * 1000L; nanosleep(&sl, nullptr); }
            const size_t cnt = std::min(packet, n_samples[i] - off), before = n->outputs.size();
            CAudioFrame fr; fr.sample_rate = 16000; fr.channels = 1; fr.samples = pcm[i] + off; fr.sample_count = cnt;
            CPacket pk; pk.packet_type = SK_PACKET_RAW_AUDIO; pk.data = &fr; pk.len = sizeof(CAudioFrame);
            n->cb_error.clear();
            const long a = now_us();
            CResult res = n->plugin->api->process_packet(n->handle, "in", &pk, out_shim, n, tel_shim, n);
            const long b = now_us();
            if (!res.success) { n->last_error = res.error_message ? res.error_message : "Unknown plugin error"; n->failed = true; rc[i] = -1; }
            if (n->outputs.size() > before && n_lat[i] < max_lat) lat_ms[(size_t)i * max_lat + n_lat[i]++] = (b - a) * 1e-3;
        }
    });
    for (auto& t : th) t.join();
    if (wall_ms) *wall_ms = now_us() * 1e-3;
    for (int r : rc) if (r) return -1;
    return 0;
}
size_t mh_output_count(mh_node* n) { return n->outputs.size(); }
const char* mh_output_pin(mh_node* n, size_t i) { return n->outputs[i].pin.c_str(); }
int mh_output_type(mh_node* n, size_t i) { return n->outputs[i].packet_type; }
const char* mh_output_payload(mh_node* n, size_t i, size_t* len) { if (len) *len = n->outputs[i].payload.size(); return n->outputs[i].payload.data(); }
size_t mh_telemetry_count(mh_node* n) { return n->telemetry.size(); }
const char* mh_telemetry_type(mh_node* n, size_t i) { return n->telemetry[i].event_type.c_str(); }
const char* mh_telemetry_json(mh_node* n, size_t i) { return n->telemetry[i].json.c_str(); }
size_t mh_log_count(mh_node* n) { return n->logs.size(); }
const char* mh_log(mh_node* n, size_t i) { return n->logs[i].c_str(); }
const char* mh_last_error(mh_node* n) { return n->last_error.c_str(); }
void mh_destroy_node(mh_node* n) { if (!n) return; if (n->handle) n->plugin->api->destroy_instance(n->handle); delete n; }

// ------------------------------------------------------------------ audio::resampler (resampler.rs:148-743), mono/stereo interleaved f32
// rubato::FastFixedIn<f32> with PolynomialDegree::Linear, restated (rubato 0.16.2 asynchro_fast.rs: 2*8-sample history,
// last_index starts at -4, two-point interpolation (1-frac)*y0 + frac*y1, fixed ratio).
struct FastFixedInLinear {
    int ch, chunk; double last_index, ratio; std::vector<std::vector<float>> buf;
    FastFixedInLinear(double r, int chunk_frames, int channels) : ch(channels), chunk(chunk_frames), last_index(-4.0), ratio(r), buf(channels, std::vector<float>(chunk_frames + 16, 0.0f)) {}
    void process(const std::vector<std::vector<float>>& in, std::vector<std::vector<float>>& out) {
        for (int c = 0; c < ch; ++c) { std::copy(buf[c].begin() + chunk, buf[c].begin() + chunk + 16, buf[c].begin());
        std::copy(in[c].begin(), in[c].begin() + chunk, buf[c].begin() + 16); out[c].clear(); }
        double idx = last_index; const double t_ratio = 1.0 / ratio; const double end_idx = (double)(chunk - 9) - std::ceil(t_ratio);
        while (idx < end_idx) {
            idx += t_ratio; const double fl = std::floor(idx); const long start = (long)fl; const float frac = (float)(idx - fl);
            for (int c = 0; c < ch; ++c) { const float* b = buf[c].data() + (start + 16); out[c].push_back((1.0f - frac) * b[0] + frac * b[1]); }
        }
        last_index = idx - (double)chunk;
    }
};
struct mh_rs_packet { std::vector<float> samples; uint64_t timestamp_us; bool has_ts; uint64_t duration_us; uint64_t sequence; };
struct mh_resampler {
    uint32_t target; size_t chunk_frames, out_frame; bool init = false, needs = false; uint32_t rate = 0; uint16_t channels = 0;
    FastFixedInLinear* rs = nullptr; std::vector<float> sample_buffer, output_buffer; uint64_t seq = 0; bool has_ts = false; uint64_t ts = 0;
    std::vector<mh_rs_packet> out; std::string err;
    ~mh_resampler() { delete rs; }
    static uint64_t dur_us(uint32_t rate, size_t frames) { if (!rate) return 0; return ((uint64_t)frames * 1000000ull) / rate; }   // resampler.rs:108-116
    // next_metadata :286-297
    void emit(const float* d, size_t n) { mh_rs_packet p; p.samples.assign(d, d + n); p.duration_us = dur_us(target, n / channels); p.has_ts = has_ts;
    p.timestamp_us = ts; p.sequence = seq++; if (has_ts) ts += p.duration_us; out.push_back(std::move(p)); }
    void drain_output_frames() { const size_t fs = out_frame * channels;
    size_t off = 0; while (output_buffer.size() - off >= fs) { emit(output_buffer.data() + off, fs);
    off += fs; } output_buffer.erase(output_buffer.begin(), output_buffer.begin() + off); }
};
mh_resampler* mh_resampler_new(uint32_t target_rate, size_t chunk_frames, size_t output_frame_size, char* err, size_t errlen) {
    if (target_rate == 0) { snprintf(err, errlen, "target_sample_rate must be greater than 0"); return nullptr; }
    if (chunk_frames == 0) { snprintf(err, errlen, "chunk_frames must be greater than 0"); return nullptr; }
    if (output_frame_size != 0) { const size_t ok[] = {120, 240, 480, 960, 1920, 2880}; bool f = false; for (size_t v : ok) f = f || v == output_frame_size;
    if (!f) { snprintf(err, errlen, "output_frame_size must be 0 (disabled) or a valid Opus frame size: [120, 240, 480, 960, 1920, 2880]"); return nullptr; } }
    mh_resampler* r = new mh_resampler(); r->target = target_rate; r->chunk_frames = chunk_frames; r->out_frame = output_frame_size; return r;
}
int mh_resampler_push(mh_resampler* r, const float* samples, size_t count, uint32_t rate, uint16_t channels, int has_ts, uint64_t ts_us) {
    if (!r->init) { r->init = true; r->needs = rate != r->target; r->rate = rate; r->channels = channels; if (has_ts) { r->has_ts = true; r->ts = ts_us; }
        if (r->needs) r->rs = new FastFixedInLinear((double)r->target / (double)rate, (int)r->chunk_frames, channels); }
    if (rate != r->rate || channels != r->channels) { char b[160];
    snprintf(b, sizeof b, "Audio format changed mid-stream: expected %uHz/%uch, got %uHz/%uch", r->rate, r->channels, rate, channels); r->err = b; return -1; }
    if (!r->needs) {   // R2: pass-through / re-chunk (resampler.rs:299-373)
        if (r->out_frame == 0) { r->emit(samples, count); return 0; }   // forwarded unchanged (metadata of the input packet in the reference)
        r->output_buffer.insert(r->output_buffer.end(), samples, samples + count); r->drain_output_frames(); return 0;
    }
    r->sample_buffer.insert(r->sample_buffer.end(), samples, samples + count);   // R1 (resampler.rs:375-527)
    const size_t cs = r->chunk_frames * channels; size_t off = 0;
    std::vector<std::vector<float>> pin(channels, std::vector<float>(r->chunk_frames)), pout(channels);
    while (r->sample_buffer.size() - off >= cs) {
        for (size_t f = 0; f < r->chunk_frames; ++f) for (int c = 0; c < channels; ++c) pin[c][f] = r->sample_buffer[off + f * channels + c];
        r->rs->process(pin, pout);
        const size_t of = pout[0].size(); std::vector<float> inter(of * channels);
        for (size_t f = 0; f < of; ++f) for (int c = 0; c < channels; ++c) inter[f * channels + c] = pout[c][f];
        if (r->out_frame > 0) { r->output_buffer.insert(r->output_buffer.end(), inter.begin(), inter.end()); r->drain_output_frames(); } else r->emit(inter.data(), inter.size());
        off += cs;
    }
    r->sample_buffer.erase(r->sample_buffer.begin(), r->sample_buffer.begin() + off);
    return 0;
}
// input closed: R3 remainder with a fresh resampler sized to the remainder, then the final short frame (resampler.rs:543-730)
int mh_resampler_finish(mh_resampler* r) {
    if (r->needs && !r->sample_buffer.empty()) {
        const size_t rem = r->sample_buffer.size() / r->channels;
        if (rem > 0) {
            FastFixedInLinear t((double)r->target / (double)r->rate, (int)rem, r->channels);
            std::vector<std::vector<float>> pin(r->channels, std::vector<float>(rem)), pout(r->channels);
            for (size_t f = 0; f < rem; ++f) for (int c = 0; c < r->channels; ++c) pin[c][f] = r->sample_buffer[f * r->channels + c];
            t.process(pin, pout);
            const size_t of = pout[0].size(); std::vector<float> inter(of * r->channels);
            for (size_t f = 0; f < of; ++f) for (int c = 0; c < r->channels; ++c) inter[f * r->channels + c] = pout[c][f];
            if (r->out_frame > 0) { r->output_buffer.insert(r->output_buffer.end(), inter.begin(), inter.end()); r->drain_output_frames(); } else r->emit(inter.data(), inter.size());
        }
        r->sample_buffer.clear();
    }
    if (!r->output_buffer.empty() && r->out_frame > 0) { r->emit(r->output_buffer.data(), r->output_buffer.size()); r->output_buffer.clear(); }
    return 0;
}
size_t mh_resampler_out_count(mh_resampler* r) { return r->out.size(); }
const float* mh_resampler_out(mh_resampler* r, size_t i, size_t* n, uint64_t* ts, int* has_ts, uint64_t* dur, uint64_t* seq) {
    const mh_rs_packet& p = r->out[i]; *n = p.samples.size(); *ts = p.timestamp_us; *has_ts = p.has_ts; *dur = p.duration_us; *seq = p.sequence; return p.samples.data();
}
void mh_resampler_clear(mh_resampler* r) { r->out.clear(); }
const char* mh_resampler_error(mh_resampler* r) { return r->err.c_str(); }
void mh_resampler_free(mh_resampler* r) { delete r; }

// ------------------------------------------------------------------ segmenter test hook: drives skw::Segmenter (the class the plugin uses)
// with a scripted per-frame probability; cuts[i] = {start_ms, end_ms, n_samples, reason, silence_ms or -1, frame index}
struct ScriptVad : skw::Vad { const float* p; int i = 0; float process_chunk(const float*) override { return p[i++]; } };
int mh_segment_sim(const float* prob, int n_frames, float threshold, uint64_t min_silence_ms, float max_secs, long long* cuts, int max_cuts) {
    skw::Segmenter seg; seg.configure(threshold, min_silence_ms, max_secs); ScriptVad vad; vad.p = prob; int n_cuts = 0; std::string err;
    std::vector<float> frame(512, 0.25f);
    for (int f = 0; f < n_frames; ++f) {
        seg.push(frame.data(), 512, vad, [](const skw::SpeechStart&) {}, [&](const skw::SegmentCut& c) {
            if (n_cuts < max_cuts) { long long* o = cuts + 6 * n_cuts; o[0] = (long long)c.start_time_ms;
            o[1] = (long long)c.end_time_ms; o[2] = (long long)c.samples.size();
            o[3] = strcmp(c.reason, "silence") == 0 ? 1 : 0; o[4] = c.has_silence_duration ? (long long)c.silence_duration_ms : -1; o[5] = f; }
            n_cuts++; return true; }, &err);
    }
    return n_cuts;
}
// ------------------------------------------------------------------ core::json_serialize (json_serialize.rs:85-107)
// serde_json::to_vec / to_vec_pretty of the externally tagged `Packet` enum (crates/core/src/types.rs:92-113), one per output packet,
// '\n' appended when newline_delimited.  The host has DESERIALISED the plugin's Transcription payload into TranscriptionData
// (conversions.rs:356-361) before this node sees it, so the payload is parsed into the typed fields here and written again in
// declaration order: text, segments[{text, start_time_ms, end_time_ms, confidence}], language, metadata (types.rs:150-175).
static void js_indent(std::string& o, bool pretty, int depth) { if (pretty) { o += '\n'; o.append((size_t)depth * 2, ' '); } }
static std::string js_u64(const skw::JsonValue* v) { char b[32]; snprintf(b, sizeof b, "%llu", (unsigned long long)(v ? v->num : 0)); return b; }
static bool js_transcription(const std::string& payload, bool pretty, int depth, std::string* out, std::string* err) {
    skw::JsonValue v; if (!skw::json_parse(payload.c_str(), &v, err) || v.type != skw::JsonValue::Object) { if (err->empty()) *err = "expected an object"; return false; }
    const skw::JsonValue* text = v.get("text"); const skw::JsonValue* segs = v.get("segments"); const skw::JsonValue* lang = v.get("language"); const skw::JsonValue* meta = v.get("metadata");
    if (!text || text->type != skw::JsonValue::String || !segs || segs->type != skw::JsonValue::Array) { *err = "missing field `text` / `segments`"; return false; }
    if (meta && meta->type != skw::JsonValue::Null) { *err = "metadata is not produced on this path"; return false; }
    const char* colon = pretty ? ": " : ":";
    std::string& o = *out; o += '{';
    js_indent(o, pretty, depth + 1); o += "\"text\""; o += colon; o += skw::json_quote(text->str); o += ',';
    js_indent(o, pretty, depth + 1); o += "\"segments\""; o += colon; o += '[';
    for (size_t i = 0; i < segs->arr.size(); ++i) {
        const skw::JsonValue& sg = segs->arr[i]; const skw::JsonValue* st = sg.get("text"); const skw::JsonValue* conf = sg.get("confidence");
        if (sg.type != skw::JsonValue::Object || !st || st->type != skw::JsonValue::String) { *err = "bad segment"; return false; }
        if (i) o += ',';
        js_indent(o, pretty, depth + 2); o += '{';
        js_indent(o, pretty, depth + 3); o += "\"text\""; o += colon; o += skw::json_quote(st->str); o += ',';
        js_indent(o, pretty, depth + 3); o += "\"start_time_ms\""; o += colon; o += js_u64(sg.get("start_time_ms")); o += ',';
        js_indent(o, pretty, depth + 3); o += "\"end_time_ms\""; o += colon; o += js_u64(sg.get("end_time_ms")); o += ',';
        js_indent(o, pretty, depth + 3); o += "\"confidence\""; o += colon; o += (conf && conf->type == skw::JsonValue::Number) ? skw::json_f32((float)conf->num) : std::string("null");
        js_indent(o, pretty, depth + 2); o += '}';
    }
    if (!segs->arr.empty()) js_indent(o, pretty, depth + 1);
    o += "],";
    js_indent(o, pretty, depth + 1); o += "\"language\""; o += colon; o += (lang && lang->type == skw::JsonValue::String) ? skw::json_quote(lang->str) : std::string("null"); o += ',';
    js_indent(o, pretty, depth + 1); o += "\"metadata\""; o += colon; o += "null";
    js_indent(o, pretty, depth); o += '}';
    return true;
}
// every output packet of the node through JsonSerialize{pretty, newline_delimited}; returns the concatenated Binary payloads (content type application/json)
const char* mh_json_serialize(mh_node* n, int pretty, int newline_delimited, size_t* len) {
    static thread_local std::string r; r.clear(); std::string err;
    for (const mh_output& o : n->outputs) {
        const char* colon = pretty ? ": " : ":";
        if (o.packet_type == SK_PACKET_TRANSCRIPTION) {
            r += '{'; js_indent(r, pretty != 0, 1); r += "\"Transcription\""; r += colon;
            if (!js_transcription(o.payload, pretty != 0, 1, &r, &err)) { n->last_error = "Failed to serialize packet to JSON: " + err; if (len) *len = 0; return nullptr; }
            js_indent(r, pretty != 0, 0); r += '}';
        } else if (o.packet_type == SK_PACKET_TEXT) {
            std::string t = o.payload; if (!t.empty() && t.back() == '\0') t.pop_back();
            r += '{'; js_indent(r, pretty != 0, 1); r += "\"Text\""; r += colon; r += skw::json_quote(t); js_indent(r, pretty != 0, 0); r += '}';
        } else { n->last_error = "Failed to serialize packet to JSON: packet type not produced on this path"; if (len) *len = 0; return nullptr; }
        if (newline_delimited) r += '\n';
    }
    if (len) *len = r.size();
    return r.data();
}
// ---- the Kokoro node's text front end (skw_kokoro_text.h), exposed so that tests/test_cpu_kokoro.py can run the reference's own splitter vectors without a GPU
const char* mh_kokoro_sanitize(const char* s) { static thread_local std::string r; r = skw::kokoro::sanitize_text(s); return r.c_str(); }
// one extract_sentence call on *buffer (in/out, NUL-terminated, capacity cap): returns 1 and the sentence, or 0
int mh_kokoro_extract_sentence(char* buffer, size_t cap, size_t min_length, char* sentence, size_t sentence_cap) {
    std::string b = buffer, s; const bool got = skw::kokoro::SentenceSplitter(min_length).extract_sentence(&b, &s);
    snprintf(buffer, cap, "%s", b.c_str()); if (got) snprintf(sentence, sentence_cap, "%s", s.c_str()); return got ? 1 : 0;
}
int mh_kokoro_flush(char* buffer, size_t cap, char* out, size_t out_cap) { std::string b = buffer, s;
const bool got = skw::kokoro::SentenceSplitter::flush(&b, &s); snprintf(buffer, cap, "%s", b.c_str()); if (got) snprintf(out, out_cap, "%s", s.c_str());
return got ? 1 : 0; }
const char* mh_kokoro_preview(const char* s, size_t max_chars) { static thread_local std::string r; if (!skw::kokoro::text_preview(s, max_chars, &r)) return nullptr; return r.c_str(); }
const char* mh_json_quote(const char* s) { static thread_local std::string r; r = skw::json_quote(s); return r.c_str(); }
const char* mh_json_f32(float f) { static thread_local std::string r; r = skw::json_f32(f); return r.c_str(); }
const char* mh_utf8_trim(const char* s) { static thread_local std::string r; r = skw::utf8_trim(s); return r.c_str(); }
int mh_utf8_valid(const char* s, size_t n) { return skw::utf8_valid(std::string(s, n)) ? 1 : 0; }

}  // extern "C"
