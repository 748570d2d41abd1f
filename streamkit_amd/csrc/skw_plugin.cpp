// skw_plugin.cpp — libwhisper.so: the StreamKit native plugin `whisper` (registered by the host as
// plugin::native::whisper) on top of the MI355X engine.  Drop-in for the reference cdylib:
//   /root/reference/plugins/native/whisper/src/lib.rs        (this file restates it function by function)
//   /root/reference/sdks/plugin-sdk/native/src/lib.rs:426-856 (the six extern "C" entry points the macro generates)
// ABI: include/streamkit_native_abi.h.
//
// Params beyond the reference's are ADDITIVE (unknown keys are ignored by the reference's serde config, lib.rs:66-104):
// vad_mode, batch_window_ms, max_batch, flush_tail, precision, gpu_device: "auto", and input_sample_rate / input_resample_mode (the
// audio::resampler node's arithmetic run inside this plugin, skw_resampler_core.h: a 48 kHz Opus source then needs no node in between).
// One BEHAVIOURAL difference remains and is not additive: with the default vad_mode "auto" the Silero model at `vad_model_path`
// gates what Whisper sees exactly as in the reference (skw_silero.h) only when that file exists; when it does not, the
// reference fails ("Failed to initialize VAD: ...") while this build logs a warning and falls back to an RMS energy gate
// (p = rms / (rms + 0.01), so vad_threshold then means a level, not a speech probability).  `vad_mode: "silero"` restores the
// reference's hard failure.  INTEGRATION.md section D lists this with the other observable differences.
#include "../../include/streamkit_native_abi.h"
#include "../../include/skw_engine.h"
#include "skw_segmenter.h"
#include "skw_silero.h"
#include "skw_resampler_core.h"
#include <sys/stat.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------ error strings (conversions.rs:441-461: thread-local, borrowed)
thread_local std::string g_last_error;
CResult ok_result() { CResult r; r.success = true; r.error_message = nullptr; return r; }
CResult err_result(const std::string& msg) {
    g_last_error = msg; for (auto& ch : g_last_error) if (ch == '\0') ch = ' ';
    CResult r; r.success = false; r.error_message = g_last_error.c_str(); return r;
}
CResult err_null() { CResult r; r.success = false; r.error_message = nullptr; return r; }
// Nothing may unwind into the host's Rust frames (SURVEY.md section 8b "Errors": panics / C++ exceptions must not cross): every entry point
// body runs under this — std::bad_alloc / length_error from the buffers, std::system_error from the worker thread, std::future_error ...
template <typename F> CResult guarded(const char* what, F&& f) {
    try { return f(); }
    catch (const std::exception& e) { return err_result(std::string(what) + ": " + e.what()); }
    catch (...) { return err_result(std::string(what) + ": unknown C++ exception"); }
}

// ------------------------------------------------------------------ configuration (lib.rs:25-157)
struct WhisperConfig {
    std::string model_path = "models/ggml-base.en-q5_1.bin";
    std::string language = "en";
    std::string vad_model_path = "models/silero_vad.onnx";
    float vad_threshold = 0.5f;
    uint64_t min_silence_duration_ms = 700;
    float max_segment_duration_secs = 30.0f;
    uint64_t n_threads = 0;
    bool use_gpu = false; int gpu_device = 0; bool gpu_device_auto = false;
    bool suppress_blank = true, suppress_non_speech_tokens = true;
    bool emit_vad_events = false;
    // additive
    std::string vad_mode = "auto";   // auto | silero | energy | always
    int batch_window_ms = 2; int max_batch = 64; bool flush_tail = false;
    std::string precision = "exact"; // exact | f16_mfma  (include/skw_engine.h, SKW_PRECISION_*)
    uint32_t input_sample_rate = 16000; std::string input_resample_mode = "linear";   // linear (the audio::resampler node's rubato arithmetic, bit for bit) | polyphase
};

bool parse_config(const char* json, WhisperConfig* cfg, std::string* err) {
    if (!json || !*json) return true;
    skw::JsonValue v; std::string perr;
    if (!skw::json_parse(json, &v, &perr)) { *err = "Invalid config: " + perr; return false; }
    if (v.type == skw::JsonValue::Null) return true;
    if (v.type != skw::JsonValue::Object) { *err = "Invalid config: expected a JSON object"; return false; }
    auto str = [&](const char* k, std::string* dst) { const skw::JsonValue* x = v.get(k);
    if (!x) return true; if (x->type != skw::JsonValue::String) { *err = std::string("Invalid config: invalid type for `") + k + "`, expected a string";
    return false; } *dst = x->str; return true; };
    auto num = [&](const char* k, double* dst) { const skw::JsonValue* x = v.get(k);
    if (!x) return true; if (x->type != skw::JsonValue::Number) { *err = std::string("Invalid config: invalid type for `") + k + "`, expected a number";
    return false; } *dst = x->num; return true; };
    auto boo = [&](const char* k, bool* dst) { const skw::JsonValue* x = v.get(k);
    if (!x) return true; if (x->type != skw::JsonValue::Bool) { *err = std::string("Invalid config: invalid type for `") + k + "`, expected a boolean";
    return false; } *dst = x->b; return true; };
    double d;
    if (!str("model_path", &cfg->model_path) || !str("language", &cfg->language) || !str("vad_model_path", &cfg->vad_model_path) || !str("vad_mode", &cfg->vad_mode)
        || !str("precision", &cfg->precision)) return false;
    if (cfg->precision != "exact" && cfg->precision != "f16_mfma") { *err = "Invalid config: precision must be \"exact\" or \"f16_mfma\""; return false; }
    if (!str("input_resample_mode", &cfg->input_resample_mode)) return false;
    if (cfg->input_resample_mode != "linear" && cfg->input_resample_mode != "polyphase") { *err = "Invalid config: input_resample_mode must be \"linear\" or \"polyphase\""; return false; }
    d = cfg->input_sample_rate; if (!num("input_sample_rate", &d)) return false;
    if (d < 1000 || d > 768000 || d != std::floor(d)) { *err = "Invalid config: input_sample_rate must be an integer between 1000 and 768000";
    return false; } cfg->input_sample_rate = (uint32_t)d;
    d = cfg->vad_threshold; if (!num("vad_threshold", &d)) return false; cfg->vad_threshold = (float)d;
    d = (double)cfg->min_silence_duration_ms; if (!num("min_silence_duration_ms", &d)) return false;
    if (d < 0 || d != std::floor(d)) { *err = "Invalid config: min_silence_duration_ms must be a non-negative integer";
    return false; } cfg->min_silence_duration_ms = (uint64_t)d;
    d = cfg->max_segment_duration_secs; if (!num("max_segment_duration_secs", &d)) return false; cfg->max_segment_duration_secs = (float)d;
    d = (double)cfg->n_threads; if (!num("n_threads", &d)) return false;
    if (d < 0 || d != std::floor(d)) { *err = "Invalid config: n_threads must be a non-negative integer"; return false; } cfg->n_threads = (uint64_t)d;
    {   // gpu_device: an integer as in the reference (lib.rs:31-33), or (additive) the string "auto": instances are dealt round-robin over the visible GPUs
        const skw::JsonValue* x = v.get("gpu_device");
        if (x && x->type == skw::JsonValue::String) { if (x->str != "auto") { *err = "Invalid config: gpu_device must be an integer or \"auto\""; return false; } cfg->gpu_device_auto = true; }
        else { d = cfg->gpu_device; if (!num("gpu_device", &d)) return false; if (d != std::floor(d)) { *err = "Invalid config: gpu_device must be an integer";
        return false; } cfg->gpu_device = (int)d; }
    }
    d = cfg->batch_window_ms; if (!num("batch_window_ms", &d)) return false; cfg->batch_window_ms = (int)d;
    d = cfg->max_batch; if (!num("max_batch", &d)) return false; cfg->max_batch = std::max(1, (int)d);
    if (!boo("use_gpu", &cfg->use_gpu) || !boo("suppress_blank", &cfg->suppress_blank) || !boo("suppress_non_speech_tokens", &cfg->suppress_non_speech_tokens) ||
        !boo("emit_vad_events", &cfg->emit_vad_events) || !boo("flush_tail", &cfg->flush_tail)) return false;
    return true;
}

// ------------------------------------------------------------------ shared engine: model cache + batch scheduler
// One per (model_path, use_gpu, gpu_device) — the key of the reference's WHISPER_CONTEXT_CACHE (lib.rs:175-180, 330).
// Instances submit finished speech segments; a worker thread forms batches and runs skw_full_batch.
// A segment's samples in page-locked memory, so that the engine's H2D copy of the batch is asynchronous DMA (64 x 30 s: ~2.3 ms) and not the driver's staged copy out of pageable
// memory (~6 ms, all of it in front of the first kernel).  Buffers are recycled through a process-wide pool (page-locking is not free: one allocation per first use of a slot).
struct PinnedPool {
    struct Buf { float* p; size_t cap; };
    std::mutex mu; std::vector<Buf> free_; size_t pooled_bytes = 0;
    static PinnedPool& get() { static PinnedPool* pool = new PinnedPool(); return *pool; }      // (leaked on purpose: instances may outlive static destruction order)
    Buf acquire(size_t n) {
        {   std::lock_guard<std::mutex> l(mu); int best = -1;
            for (int i = 0; i < (int)free_.size(); ++i) if (free_[i].cap >= n && (best < 0 || free_[i].cap < free_[best].cap)) best = i;
            if (best >= 0) { Buf b = free_[best]; free_.erase(free_.begin() + best); pooled_bytes -= b.cap * sizeof(float); return b; } }
        const size_t cap = std::max<size_t>((n + 65535) & ~(size_t)65535, 491520);      // at least a 30.72 s segment: the common size comes back from the pool
        return Buf{(float*)skw_host_alloc(cap * sizeof(float)), cap};
    }
    void release(Buf b) {
        if (!b.p) return;
        {   std::lock_guard<std::mutex> l(mu);
            if (pooled_bytes + b.cap * sizeof(float) <= ((size_t)512 << 20)) { free_.push_back(b); pooled_bytes += b.cap * sizeof(float); return; } }
        skw_host_free(b.p);
    }
};
struct Job {
    std::vector<float> pcm_pageable; PinnedPool::Buf pinned{nullptr, 0}; size_t n = 0;
    skw_full_params params; std::promise<int> done; skw_result result{}; std::string error;
    uint32_t* rng = nullptr;      // the owning instance's std::mt19937 stream (WhisperPlugin::rng): continued by this segment's sampled passes, if it needs any
    void set_samples(const std::vector<float>& v) {
        n = v.size(); pinned = PinnedPool::get().acquire(n);
        if (pinned.p) memcpy(pinned.p, v.data(), n * sizeof(float)); else pcm_pageable = v;      // (no page-locked memory to be had: the ordinary copy path)
    }
    const float* data() const { return pinned.p ? pinned.p : pcm_pageable.data(); }
    ~Job() { PinnedPool::get().release(pinned); }
};
struct SharedEngine {
    skw_model* model = nullptr;
    skw_ctx* ctx = nullptr;
    int max_batch = 0;          // rows the workspace was created for
    int max_samples = 0;        // samples per row the workspace was created for
    int batch_limit = 64;       // scheduler: largest batch formed (the max_batch param of the instance created most recently; under mu)
    int window_ms = 2;          // scheduler: how long a batch waits for more jobs (batch_window_ms of the instance created most recently; under mu)
    int precision = SKW_PRECISION_EXACT;
    static const int kMaxSamples = 16000 * 121;   // schema maximum of max_segment_duration_secs (120 s) + one VAD frame of slack: no segment is longer
    static int samples_for(float max_segment_secs) { const double s = std::min(120.0, std::max(1.0, (double)max_segment_secs)); return std::min(kMaxSamples, (int)std::ceil(s * 16000.0) + 1024); }
    // The workspace is sized for the longest segment an instance of this engine can cut (its max_segment_duration_secs) and the largest batch one asked
    // for, not for the schema's maxima: 31 s by default instead of 121 (0.2 GB instead of 0.8 at 64 rows).  An instance that allows longer segments or
    // larger batches grows it, between batches (the engine outlives its instances: lib.rs:170-180).
    // Grows to max(configured floor, what was held, what is needed), rounded up (whole seconds, multiples of 8 rows) so that a run of slightly longer segments does not
    // re-create it batch after batch.  When the larger workspace cannot be had, the previous size — or the floor — is restored and only the oversize batch fails.
    int floor_samples = 0, floor_batch = 0;      // what the instances of this engine were configured for (get_engine); the workspace is never smaller while one is alive
    std::mutex ws_mu;                            // the workspace is in use (a batch is running) or being re-created / released
    int live_instances = 0;                      // under ws_mu: plugin instances holding this engine; the last one to go releases the workspace (the MODEL stays, lib.rs:170-180)
    bool ensure_workspace(int need_samples, int need_batch, std::string* err) {
        if (ctx && need_samples <= max_samples && need_batch <= max_batch) return true;
        const int was_s = max_samples, was_b = max_batch;
        const int ns = std::min(kMaxSamples, (std::max(std::max(need_samples, max_samples), floor_samples) + 15999) / 16000 * 16000);
        const int nb = (std::max(std::max(need_batch, max_batch), floor_batch) + 7) & ~7;
        char ebuf[512] = {0};
        if (ctx) { skw_ctx_free(ctx); ctx = nullptr; max_samples = 0; max_batch = 0; }      // release first: two workspaces need not fit side by side
        skw_ctx* nc = skw_ctx_create(model, nb, ns, ebuf, sizeof ebuf);
        if (!nc) {
            *err = std::string("Failed to create Whisper state: ") + ebuf;
            const int rs = std::max(was_s, floor_samples), rb = std::max(was_b, floor_batch);
            if (rs > 0 && rb > 0 && (rs < ns || rb < nb)) {      // back to what there was: later batches of the usual size must not pay for this one
                char e2[512] = {0};
                if (skw_ctx* oc = skw_ctx_create(model, rb, rs, e2, sizeof e2)) { skw_ctx_set_precision(oc, precision); ctx = oc; max_samples = rs; max_batch = rb; }
            }
            return false;
        }
        skw_ctx_set_precision(nc, precision);
        ctx = nc; max_samples = ns; max_batch = nb;
        return true;
    }
    void instance_added() { std::lock_guard<std::mutex> l(ws_mu); ++live_instances; }
    // The reference keeps the CONTEXT (weights) for the life of the process and drops each instance's WhisperState with the instance (lib.rs:160-180, 376-380).  Here the
    // batch workspace plays the state's part for all instances at once: when the last one goes — the prewarm node, or every session after a quiet spell — its device memory
    // (cross / self K-V caches for max_batch rows: 6.6 GB at 64 rows of Whisper-small) is returned; the next batch re-creates it (~3-20 ms, logged by get_engine).
    void instance_gone() {
        std::lock_guard<std::mutex> l(ws_mu);
        if (--live_instances <= 0) { live_instances = 0; if (ctx) { skw_ctx_free(ctx); ctx = nullptr; max_samples = 0; max_batch = 0; } }
    }
    std::mutex mu; std::condition_variable cv; std::deque<std::shared_ptr<Job>> queue; bool stop = false; std::thread worker;
    ~SharedEngine() {
        { std::lock_guard<std::mutex> l(mu); stop = true; } cv.notify_all(); if (worker.joinable()) worker.join();
        if (ctx) skw_ctx_free(ctx); if (model) skw_model_free(model);
    }
    void run() {
        for (;;) {
            std::vector<std::shared_ptr<Job>> batch;
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return stop || !queue.empty(); });
                if (stop && queue.empty()) return;
                // batch formation: give concurrent instances a short window to join
                auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(window_ms);
                while ((int)queue.size() < batch_limit && !stop) { if (cv.wait_until(l, deadline) == std::cv_status::timeout) break; }
                while (!queue.empty() && (int)batch.size() < batch_limit) {
                    // one skw_full_batch call shares its params: group by the params of the first job
                    if (!batch.empty() && memcmp(&batch[0]->params, &queue.front()->params, sizeof(skw_full_params)) != 0) break;
                    batch.push_back(queue.front()); queue.pop_front();
                }
            }
            const int n = (int)batch.size();
            int rc = -1; std::string why;
            try {
                std::vector<const float*> ptrs(n); std::vector<int32_t> ns(n); std::vector<skw_result> res(n); int need = 0;
                for (int i = 0; i < n; ++i) { ptrs[i] = batch[i]->data(); ns[i] = (int32_t)batch[i]->n; need = std::max(need, (int)ns[i]); }
                std::lock_guard<std::mutex> wl(ws_mu);
                if (ensure_workspace(need, n, &why)) {
                    std::vector<uint32_t*> rngs(n); for (int i = 0; i < n; ++i) rngs[i] = batch[i]->rng;
                    rc = skw_full_batch_rng(ctx, &batch[0]->params, ptrs.data(), ns.data(), n, 0, rngs.data(), res.data());
                    if (rc == 0) for (int i = 0; i < n; ++i) batch[i]->result = res[i]; else why = skw_ctx_last_error(ctx);
                }
            } catch (const std::exception& e) { rc = -1; why = e.what(); }     // the worker thread must not die with waiters blocked on it
            for (int i = 0; i < n; ++i) { if (rc != 0) batch[i]->error = why; batch[i]->done.set_value(rc); }
        }
    }
};
// WHISPER_CONTEXT_CACHE (lib.rs:170-180): process-global, STRONG references — an entry lives until the library is unloaded or the process exits, so a
// prewarmed model (apps/skit/src/plugins.rs:265-306: "Node is dropped immediately, but ... model loading via Arc persist") and every Oneshot request after
// the first find it loaded.  What is cached here is the whole SharedEngine: the weights with their device images, the batch workspace (the reference's
// per-instance WhisperState has no counterpart: instances share one batched context) and the scheduler thread.
// The map is a function-local static constructed after the HIP runtime is up (skw_device_count makes the first HIP call), so that at process exit it is
// destroyed — workers joined, device memory released — before the runtime's own teardown; a dlclose of the plugin runs the same destructor.
struct EngineCache {
    std::mutex mu;
    std::map<std::string, std::shared_ptr<SharedEngine>> map;
};
EngineCache& engine_cache() { (void)skw_device_count(); static EngineCache c; return c; }
std::atomic<int> g_model_loads{0}, g_cache_hits{0};

struct WhisperPlugin;
std::shared_ptr<SharedEngine> get_engine(const WhisperConfig& cfg, WhisperPlugin* who, std::string* err);
// gpu_device: "auto" (additive): instance k of the process runs on GPU k mod n_gpus — one SharedEngine (model copy + batch scheduler)
// per device, which is how the reference's docs spread pipelines over GPUs by hand (docs/.../deployment/gpu.md:64-66)
std::atomic<unsigned> g_auto_rr{0};
void resolve_auto_device(WhisperConfig* cfg) {
    if (!cfg->gpu_device_auto) return;
    const int n = skw_device_count();
    cfg->gpu_device = n > 0 ? (int)(g_auto_rr.fetch_add(1) % (unsigned)n) : 0;
}

// ------------------------------------------------------------------ the plugin instance (lib.rs:199-221)
struct WhisperPlugin {
    WhisperConfig config; std::shared_ptr<SharedEngine> engine; skw::Segmenter seg; std::unique_ptr<skw::Vad> vad;
    // the engine this instance holds, counted (SharedEngine::live_instances): the last holder to let go returns the batch workspace
    // whisper.cpp keeps one std::mt19937 per whisper_state (seeded with 0 at whisper_init_state) for the temperature ladder's draws and lets it run on across calls; the
    // reference creates one state per instance (lib.rs:377-379) and a new one when update_params swaps the context (lib.rs:520-535).  This is that generator: every segment of
    // this instance continues it (skw_full_batch_rng), whichever batch the segment lands in.
    uint32_t rng[SKW_RNG_STATE_WORDS];
    void hold(std::shared_ptr<SharedEngine> e) { if (e) e->instance_added(); if (engine) engine->instance_gone(); engine = std::move(e); skw_rng_state_init(rng); }
    ~WhisperPlugin() { if (engine) engine->instance_gone(); }
    std::unique_ptr<skw::ResamplerCore> front;      // input_sample_rate != 16000: the audio::resampler node's arithmetic on the GPU, feeding the segmenter
    CLogCallback log_cb = nullptr; void* log_ud = nullptr;
    void log(CLogLevel lv, const char* fmt, ...) {
        if (!log_cb) return; char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
        log_cb(lv, "whisper_plugin_native", buf, log_ud);
    }
};

std::shared_ptr<SharedEngine> get_engine(const WhisperConfig& cfg, WhisperPlugin* who, std::string* err) {
    char keybuf[96];
    snprintf(keybuf, sizeof keybuf, "|%d|%d|%s", cfg.use_gpu ? 1 : 0, cfg.gpu_device, cfg.precision.c_str());   // the reference's key (path, use_gpu, gpu_device) + the additive precision
    const std::string key = cfg.model_path + keybuf;
    EngineCache& cache = engine_cache();
    std::lock_guard<std::mutex> l(cache.mu);
    auto it = cache.map.find(key);
    if (it != cache.map.end()) {
        g_cache_hits.fetch_add(1);
        // additive scheduler params: the most recent instance's
        { std::lock_guard<std::mutex> le(it->second->mu); it->second->batch_limit = cfg.max_batch; it->second->window_ms = cfg.batch_window_ms; }
        { std::lock_guard<std::mutex> lw(it->second->ws_mu);      // the floor follows the largest configuration seen; the workspace itself grows (or comes back) with the next batch
          it->second->floor_samples = std::max(it->second->floor_samples, SharedEngine::samples_for(cfg.max_segment_duration_secs));
          it->second->floor_batch = std::max(it->second->floor_batch, cfg.max_batch);
          // an engine whose last instance had gone gets its workspace back here, at instance creation — where the reference creates a WhisperState (lib.rs:377-379) — not inside the first batch
          if (!it->second->ctx && !it->second->ensure_workspace(it->second->floor_samples, it->second->floor_batch, err)) return nullptr; }
        if (who) who->log(SK_LOG_INFO, "CACHE HIT: Reusing cached Whisper context (model_path=%s, gpu_device=%d, precision=%s)", cfg.model_path.c_str(), cfg.gpu_device, cfg.precision.c_str());
        return it->second;
    }
    if (who) who->log(SK_LOG_INFO, "CACHE MISS: Loading Whisper model (model_path=%s, gpu_device=%d, precision=%s)", cfg.model_path.c_str(), cfg.gpu_device, cfg.precision.c_str());
    char ebuf[512] = {0};
    auto eng = std::make_shared<SharedEngine>();
    const auto t0 = std::chrono::steady_clock::now();
    eng->model = skw_model_load(cfg.model_path.c_str(), cfg.gpu_device, ebuf, sizeof ebuf);
    if (!eng->model) { *err = ebuf[0] ? ebuf : ("Failed to load Whisper model from '" + cfg.model_path + "'"); return nullptr; }
    g_model_loads.fetch_add(1);
    const auto t1 = std::chrono::steady_clock::now();
    eng->batch_limit = cfg.max_batch;
    eng->window_ms = cfg.batch_window_ms;
    eng->precision = cfg.precision == "f16_mfma" ? SKW_PRECISION_F16_MFMA : SKW_PRECISION_EXACT;
    eng->floor_samples = SharedEngine::samples_for(cfg.max_segment_duration_secs); eng->floor_batch = cfg.max_batch;
    if (!eng->ensure_workspace(eng->floor_samples, eng->floor_batch, err)) return nullptr;
    const auto t2 = std::chrono::steady_clock::now();
    eng->worker = std::thread([e = eng.get()] { e->run(); });
    cache.map[key] = eng;
    if (who) who->log(SK_LOG_INFO, "Whisper model loaded and cached (model_load_ms=%.1f, ctx_create_ms=%.1f, max_batch=%d)",
                      std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), cfg.max_batch);
    return eng;
}

// Silero gate behind the segmenter's Vad interface; the weights of one file are shared by every instance that names it and, like the Whisper
// context, stay cached for the life of the process
struct SileroGate : skw::Vad {
    skw::SileroVad v;
    explicit SileroGate(std::shared_ptr<const skw::SileroWeights> w) : v(std::move(w)) {}
    float process_chunk(const float* f) override { return v.process_chunk(f); }
    void reset() override { v.reset(); }
};
std::mutex g_vad_mu;
std::map<std::string, std::shared_ptr<const skw::SileroWeights>> g_vad_cache;
std::shared_ptr<const skw::SileroWeights> get_silero(const std::string& path, std::string* err) {
    std::lock_guard<std::mutex> l(g_vad_mu);
    auto it = g_vad_cache.find(path);
    if (it != g_vad_cache.end()) return it->second;
    auto w = std::make_shared<skw::SileroWeights>();
    try { if (!skw::SileroVad::load_weights(path, w.get(), err)) return nullptr; }
    catch (const std::exception& e) { *err = "Failed to load VAD model from '" + path + "': " + e.what(); return nullptr; }     // a malformed file must not unwind across the C ABI
    g_vad_cache[path] = w;
    return w;
}

// SileroVAD::new at lib.rs:382-383 ("Failed to initialize VAD: {e}") / :557-560 ("Failed to reload VAD: {e}", added by the caller)
std::unique_ptr<skw::Vad> make_vad(const WhisperConfig& cfg, WhisperPlugin* p, std::string* err) {
    std::string mode = cfg.vad_mode;
    if (mode == "auto") {
        struct stat sb;
        if (stat(cfg.vad_model_path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) {
            // The file is there: run it — unless this build cannot read it (the graph is restated from the published v5 layout and has only ever
            // seen files of tools/make_synth_silero.py's making; the reference names v6, vad.rs:5).  In auto mode that is a warning and the energy
            // gate, not a node that fails to start; `vad_mode: "silero"` keeps the reference's hard failure.
            std::string e; auto w = get_silero(cfg.vad_model_path, &e);
            if (w) { if (p) p->log(SK_LOG_INFO, "Silero VAD: %s", w->bound.c_str()); return std::unique_ptr<skw::Vad>(new SileroGate(w)); }
            if (p) p->log(SK_LOG_WARN, "%s; using vad_mode=energy (set vad_mode to \"silero\" to make this an error)", e.c_str());
            mode = "energy";
        } else {
            // the reference fails here; this build keeps running on an energy gate and says so (header comment)
            if (p) p->log(SK_LOG_WARN, "Silero VAD model '%s' not found; using vad_mode=energy (set vad_mode to \"silero\" to make this an error)", cfg.vad_model_path.c_str());
            mode = "energy";
        }
    }
    if (mode == "always") return std::unique_ptr<skw::Vad>(new skw::AlwaysSpeechVad());
    if (mode == "energy") return std::unique_ptr<skw::Vad>(new skw::EnergyVad());
    if (mode == "silero") {
        std::string e; auto w = get_silero(cfg.vad_model_path, &e);
        if (!w) { *err = "Failed to initialize VAD: " + e; return nullptr; }
        if (p) p->log(SK_LOG_INFO, "Silero VAD: %s", w->bound.c_str());
        return std::unique_ptr<skw::Vad>(new SileroGate(w));
    }
    *err = "Failed to initialize VAD: unknown vad_mode '" + mode + "'"; return nullptr;
}

struct Emit {
    COutputCallback out_cb; void* out_ud; CTelemetryCallback tel_cb; void* tel_ud;
    // OutputSender::send (sdk lib.rs:200-223)
    bool send_transcription(const std::string& json, std::string* err) const {
        CPacket pk; pk.packet_type = SK_PACKET_TRANSCRIPTION; pk.data = json.data(); pk.len = json.size();
        CResult r = out_cb("out", &pk, out_ud);
        if (r.success) return true;
        *err = r.error_message ? std::string(r.error_message) : std::string("Unknown error"); return false;
    }
    // OutputSender::emit_telemetry (sdk lib.rs:237-283): best effort, errors ignored by the caller (lib.rs:438 `let _ =`)
    void telemetry(const char* event, const std::string& json) const { if (tel_cb) (void)tel_cb(event, (const uint8_t*)json.data(), json.size(), nullptr, tel_ud); }
};

// transcribe_and_emit (lib.rs:582-702)
bool transcribe_and_emit(WhisperPlugin* self, const Emit& em, const skw::SegmentCut& cut, std::string* err) {
    if (cut.samples.empty()) return true;
    if (self->config.emit_vad_events && !cut.segment_id.empty()) {
        std::string j = "{\"duration_ms\":" + std::to_string(cut.end_time_ms >= cut.start_time_ms ? cut.end_time_ms - cut.start_time_ms : 0) +
                        ",\"end_time_ms\":" + std::to_string(cut.end_time_ms) + ",\"reason\":" + skw::json_quote(cut.reason) + ",\"segment_id\":" + skw::json_quote(cut.segment_id) +
                        ",\"silence_duration_ms\":" + (cut.has_silence_duration ? std::to_string(cut.silence_duration_ms) : std::string("null")) +
                        ",\"start_time_ms\":" + std::to_string(cut.start_time_ms) + "}";
        em.telemetry("vad.speech_end", j);
    }
    auto job = std::make_shared<Job>();
    job->set_samples(cut.samples);
    skw_full_default_params(&job->params);
    int lang = self->config.language == "auto" ? -1 : skw_model_lang_id(self->config.language.c_str());     // "auto": whisper.cpp detects it from the first window
    if (lang < 0 && self->config.language != "auto") { *err = "Whisper inference failed: unknown language '" + self->config.language + "'"; return false; }
    job->params.lang_id = lang; job->params.translate = 0;
    job->params.suppress_blank = self->config.suppress_blank ? 1 : 0; job->params.suppress_nst = self->config.suppress_non_speech_tokens ? 1 : 0;
    job->params.n_threads = (int32_t)self->config.n_threads;
    job->rng = self->rng;
    if ((int)job->n > SharedEngine::kMaxSamples) { *err = "Whisper inference failed: segment longer than the engine workspace"; return false; }
    std::future<int> fut = job->done.get_future();
    { std::lock_guard<std::mutex> l(self->engine->mu); self->engine->queue.push_back(job); }
    self->engine->cv.notify_all();
    int rc = fut.get();
    if (rc != 0) { *err = "Whisper inference failed: " + job->error; return false; }
    // segments with absolute timestamps (lib.rs:648-677)
    struct Seg { std::string text; uint64_t t0, t1; };
    std::vector<Seg> segs;
    const skw_result& R = job->result;
    for (int i = 0; i < R.n_segments; ++i) {
        std::string text(R.text + R.segments[i].text_off, (size_t)R.segments[i].text_len);
        if (!skw::utf8_valid(text)) { self->log(SK_LOG_WARN, "Failed to get segment text: invalid UTF-8"); continue; }
        std::string trimmed = skw::utf8_trim(text);
        if (trimmed.empty()) continue;
        const uint64_t rel0 = (uint64_t)(R.segments[i].t0 * 10), rel1 = (uint64_t)(R.segments[i].t1 * 10);   // `as u64` of an i64
        segs.push_back(Seg{trimmed, cut.start_time_ms + rel0, cut.start_time_ms + rel1});
    }
    if (segs.empty()) self->log(SK_LOG_WARN, "Whisper inference produced no segments");
    else {
        std::string full; for (size_t i = 0; i < segs.size(); ++i) { if (i) full += " "; full += segs[i].text; }
        // serde_json::to_vec(TranscriptionData) (crates/core/src/types.rs:150-175): field order of the structs
        std::string j = "{\"text\":" + skw::json_quote(full) + ",\"segments\":[";
        for (size_t i = 0; i < segs.size(); ++i) {
            if (i) j += ",";
            j += "{\"text\":" + skw::json_quote(segs[i].text) + ",\"start_time_ms\":" + std::to_string(segs[i].t0) + ",\"end_time_ms\":" + std::to_string(segs[i].t1) + ",\"confidence\":null}";
        }
        j += "],\"language\":" + skw::json_quote(self->config.language) + ",\"metadata\":null}";
        skw_result_free(&job->result);
        if (!em.send_transcription(j, err)) return false;
        return true;
    }
    skw_result_free(&job->result);
    return true;
}

void emit_speech_start(WhisperPlugin* self, const Emit& em, const skw::SpeechStart& s) {
    if (!self->config.emit_vad_events) return;
    std::string j = "{\"segment_id\":" + skw::json_quote(s.segment_id) + ",\"speech_probability\":" + skw::json_f32(s.probability) +
                    ",\"start_time_ms\":" + std::to_string(s.start_time_ms) + ",\"threshold\":" + skw::json_f32(self->config.vad_threshold) + "}";
    em.telemetry("vad.speech_start", j);
}

// ------------------------------------------------------------------ metadata (lib.rs:224-320)
const char* const kDescription =
    "Real-time speech-to-text transcription using OpenAI's Whisper model. Features VAD-based segmentation for natural speech boundaries, "
    "GPU acceleration support, and streaming output. Requires 16kHz mono audio input.";
const char* const kSchema =
    "{\"type\":\"object\",\"properties\":{"
    "\"model_path\":{\"type\":\"string\",\"description\":\"Path to Whisper GGML model file (relative to repo root). IMPORTANT: Input audio must be 16kHz mono f32.\",\"default\":\"models/ggml-base.en-q5_1.bin\"},"
    "\"language\":{\"type\":\"string\",\"description\":\"Language code (e.g., 'en', 'es', 'fr')\",\"default\":\"en\"},"
    "\"vad_model_path\":{\"type\":\"string\",\"description\":\"Path to Silero VAD ONNX model file\",\"default\":\"models/silero_vad.onnx\"},"
    "\"vad_threshold\":{\"type\":\"number\",\"description\":\"VAD speech probability threshold (0.0-1.0)\",\"default\":0.5,\"minimum\":0.0,\"maximum\":1.0},"
    "\"min_silence_duration_ms\":{\"type\":\"integer\",\"description\":\"Minimum silence duration before transcription (milliseconds)\",\"default\":700,\"minimum\":100,\"maximum\":5000},"
    "\"max_segment_duration_secs\":{\"type\":\"number\",\"description\":\"Maximum segment duration before forced transcription (seconds)\",\"default\":30.0,\"minimum\":5.0,\"maximum\":120.0},"
    "\"n_threads\":{\"type\":\"integer\",\"description\":\"Number of threads for decoding (0 = auto: min(4, num_cores), 8-12 recommended for modern CPUs)\",\"default\":0,\"minimum\":0,\"maximum\":32},"
    "\"use_gpu\":{\"type\":\"boolean\",\"description\":\"Enable GPU acceleration (this build always runs on the MI355X selected by gpu_device)\",\"default\":false},"
    "\"gpu_device\":{\"type\":[\"integer\",\"string\"],\"description\":\"GPU device ID to use (0 = first GPU, 1 = second GPU, etc.); (additive) \\\"auto\\\" deals instances round-robin over the visible GPUs\",\"default\":0,\"minimum\":0,\"maximum\":7},"
    "\"suppress_blank\":{\"type\":\"boolean\",\"description\":\"Suppress blank/silent audio segments\",\"default\":true},"
    "\"suppress_non_speech_tokens\":{\"type\":\"boolean\",\"description\":\"Suppress non-speech tokens like [BLANK_AUDIO], [MUSIC], [APPLAUSE], etc.\",\"default\":true},"
    "\"emit_vad_events\":{\"type\":\"boolean\",\"description\":\"Emit VAD speech start/end out-of-band to the telemetry bus (does not flow through graph pins).\",\"default\":false},"
    "\"vad_mode\":{\"type\":\"string\",\"description\":\"(additive) auto (Silero when vad_model_path exists, else an energy gate) | silero | energy | always\",\"default\":\"auto\"},"
    "\"precision\":{\"type\":\"string\",\"description\":\"(additive) exact (f32-chain contractions, bit-reproducible; block-quantised model files run ggml's q8 arithmetic) | f16_mfma (f16 matrix cores; quantised files as their f16 twin)\",\"default\":\"exact\"},"
    "\"batch_window_ms\":{\"type\":\"integer\",\"description\":\"(additive) how long the per-GPU scheduler waits for concurrent instances before launching a batch\",\"default\":2},"
    "\"max_batch\":{\"type\":\"integer\",\"description\":\"(additive) largest number of segments transcribed in one GPU batch\",\"default\":64},"
    "\"flush_tail\":{\"type\":\"boolean\",\"description\":\"(additive) transcribe buffered speech when the input stream ends (the reference drops it)\",\"default\":false},"
    "\"input_sample_rate\":{\"type\":\"integer\",\"description\":\"(additive) sample rate of the mono f32 packets fed to this node; anything but 16000 is resampled to 16 kHz on the GPU with the audio::resampler node's arithmetic (chunk_frames 960) before VAD segmentation\",\"default\":16000,\"minimum\":1000,\"maximum\":768000},"
    "\"input_resample_mode\":{\"type\":\"string\",\"description\":\"(additive) linear (rubato FastFixedIn/Linear, bit for bit what audio::resampler gives) | polyphase (Kaiser-windowed sinc)\",\"default\":\"linear\"}"
    "}}";

const CAudioFormat kInFormat = {16000, 1, SK_SAMPLE_F32};
// the reference's one accepted type first; (additive) mono f32 at any rate (0 = wildcard, packet_meta.rs:97-109) so that a graph with
// `input_sample_rate: 48000` passes the host's connection check — the rate is validated per packet either way (lib.rs:183-197)
const CAudioFormat kInAnyRate = {0, 1, SK_SAMPLE_F32};
const CPacketTypeInfo kInTypes[2] = {{SK_PACKET_RAW_AUDIO, &kInFormat, nullptr}, {SK_PACKET_RAW_AUDIO, &kInAnyRate, nullptr}};
const CInputPin kInputs[1] = {{"in", kInTypes, 2}};
const COutputPin kOutputs[1] = {{"out", {SK_PACKET_TRANSCRIPTION, nullptr, nullptr}}};
const char* const kCategories[3] = {"ml", "speech", "transcription"};
const CNodeMetadata kMetadata = {"whisper", kDescription, kInputs, 1, kOutputs, 1, kSchema, kCategories, 3};

// ------------------------------------------------------------------ the six entry points (sdk lib.rs:462-854)
const CNodeMetadata* plugin_get_metadata() { return &kMetadata; }

CPluginHandle create_instance_impl(const char* params, CLogCallback log_cb, void* log_ud) {
    auto p = std::unique_ptr<WhisperPlugin>(new WhisperPlugin()); p->log_cb = log_cb; p->log_ud = log_ud;
    std::string err;
    if (!parse_config(params, &p->config, &err)) { p->log(SK_LOG_ERROR, "%s", err.c_str()); return nullptr; }
    resolve_auto_device(&p->config);
    p->hold(get_engine(p->config, p.get(), &err));
    if (!p->engine) { p->log(SK_LOG_ERROR, "%s", err.c_str()); return nullptr; }
    p->vad = make_vad(p->config, p.get(), &err);
    if (!p->vad) { p->log(SK_LOG_ERROR, "%s", err.c_str()); return nullptr; }
    p->seg.configure(p->config.vad_threshold, p->config.min_silence_duration_ms, p->config.max_segment_duration_secs);
    if (p->config.input_sample_rate != 16000) {
        p->front.reset(new skw::ResamplerCore());
        p->front->target = 16000; p->front->chunk_frames = 960; p->front->out_frame = 0;
        p->front->gpu_device = p->config.gpu_device; p->front->polyphase = p->config.input_resample_mode == "polyphase";
    }
    return (CPluginHandle)p.release();
}
CPluginHandle plugin_create_instance(const char* params, CLogCallback log_cb, void* log_ud) {
    auto fail = [&](const std::string& m) -> CPluginHandle { if (log_cb) log_cb(SK_LOG_ERROR, "whisper_plugin_native", m.c_str(), log_ud); return nullptr; };
    try { return create_instance_impl(params, log_cb, log_ud); }          // NULL = "Plugin failed to create instance" at the host (wrapper.rs:184-188)
    catch (const std::exception& e) { return fail(std::string("Failed to create Whisper plugin instance: ") + e.what()); }
    catch (...) { return fail("Failed to create Whisper plugin instance: unknown C++ exception"); }
}

// 16 kHz mono samples into the VAD segmenter (lib.rs:411-493)
bool feed_segmenter(WhisperPlugin* self, const Emit& em, const float* samples, size_t n, std::string* err) {
    bool failed = false;
    self->seg.push(samples, n, *self->vad,
                   [&](const skw::SpeechStart& s) { emit_speech_start(self, em, s); },
                   [&](const skw::SegmentCut& cut) { if (!transcribe_and_emit(self, em, cut, err)) { failed = true; return false; } return true; }, err);
    return !failed && err->empty();
}

CResult plugin_process_packet(CPluginHandle handle, const char* input_pin, const CPacket* packet, COutputCallback out_cb, void* out_ud,
                              CTelemetryCallback tel_cb, void* tel_ud) {
    if (!handle || !input_pin || !packet) return err_null();
    return guarded("Whisper plugin", [&]() -> CResult {
        WhisperPlugin* self = (WhisperPlugin*)handle;
        if (!packet->data) return err_result("Invalid packet: Null packet data pointer");
        if (packet->packet_type != SK_PACKET_RAW_AUDIO) {
            if (packet->packet_type == SK_PACKET_TEXT || packet->packet_type == SK_PACKET_TRANSCRIPTION || packet->packet_type == SK_PACKET_CUSTOM || packet->packet_type == SK_PACKET_BINARY)
                return err_result("Whisper plugin only accepts audio packets");
            return err_result("Invalid packet: Unsupported packet type");
        }
        const CAudioFrame* fr = (const CAudioFrame*)packet->data;
        if (!fr->samples) return err_result("Invalid packet: Null samples pointer in audio frame");
        // validate_audio_format (lib.rs:183-197); with the additive input_sample_rate the expected rate is the configured one
        const uint32_t want = self->config.input_sample_rate;
        if (fr->sample_rate != want) {
            if (want == 16000) return err_result("Whisper requires 16kHz audio, got " + std::to_string(fr->sample_rate) + "Hz. Please add an audio_resample node upstream.");
            return err_result("Whisper plugin is configured for " + std::to_string(want) + "Hz input (input_sample_rate), got " + std::to_string(fr->sample_rate) + "Hz.");
        }
        if (fr->channels != 1) return err_result("Whisper requires mono audio, got " + std::to_string(fr->channels) + " channels. Please add an audio_resample node upstream.");
        Emit em{out_cb, out_ud, tel_cb, tel_ud};
        std::string err;
        if (self->front) {
            if (!self->front->push(fr->samples, fr->sample_count, fr->sample_rate, fr->channels,
                                   [&](const float* d, size_t n, std::string* e) { return feed_segmenter(self, em, d, n, e); }, &err)) return err_result(err);
        } else if (!feed_segmenter(self, em, fr->samples, fr->sample_count, &err)) return err_result(err);
        return ok_result();
    });
}

CResult plugin_update_params(CPluginHandle handle, const char* params) {
    if (!handle) return err_result("Invalid handle (null)");
    return guarded("Whisper plugin", [&]() -> CResult {
        WhisperPlugin* self = (WhisperPlugin*)handle;
        if (!params || !*params) return ok_result();
        WhisperConfig nc;   // serde deserialises a fresh config from the new JSON (defaults for missing keys), lib.rs:498-499
        std::string err;
        if (!parse_config(params, &nc, &err)) return err_result(err);
        if (nc.gpu_device_auto) nc.gpu_device = self->config.gpu_device;        // an "auto" instance stays on the device it was dealt
        if (nc.model_path != self->config.model_path || nc.precision != self->config.precision || nc.use_gpu != self->config.use_gpu || nc.gpu_device != self->config.gpu_device) {
            auto eng = get_engine(nc, self, &err);
            if (!eng) return err_result("Failed to reload Whisper model: " + err);
            self->hold(eng);
        }
        if (nc.vad_model_path != self->config.vad_model_path || nc.vad_threshold != self->config.vad_threshold || nc.vad_mode != self->config.vad_mode) {
            auto v = make_vad(nc, self, &err);
            if (!v) { const std::string pre = "Failed to initialize VAD: "; return err_result("Failed to reload VAD: " + (err.compare(0, pre.size(), pre) == 0 ? err.substr(pre.size()) : err)); }
            self->vad = std::move(v);
        }
        if (nc.min_silence_duration_ms != self->config.min_silence_duration_ms) self->seg.set_min_silence_ms(nc.min_silence_duration_ms);
        self->seg.set_threshold(nc.vad_threshold); self->seg.set_max_duration_secs(nc.max_segment_duration_secs);
        nc.input_sample_rate = self->config.input_sample_rate; nc.input_resample_mode = self->config.input_resample_mode;   // the front end's rate is fixed at creation (a stream does not change rate)
        self->config = nc;
        return ok_result();
    });
}

CResult plugin_flush(CPluginHandle handle, COutputCallback out_cb, void* out_ud, CTelemetryCallback tel_cb, void* tel_ud) {
    if (!handle) return err_result("Invalid handle (null)");
    return guarded("Whisper plugin", [&]() -> CResult {
        WhisperPlugin* self = (WhisperPlugin*)handle;
        Emit em{out_cb, out_ud, tel_cb, tel_ud}; std::string err;
        // the resampler front end's remainder belongs to the stream whether or not the tail is then transcribed (resampler.rs:543-730 runs on input close)
        if (self->front && !self->front->finish([&](const float* d, size_t n, std::string* e) { return feed_segmenter(self, em, d, n, e); }, &err)) return err_result(err);
        if (!self->config.flush_tail) return ok_result();   // trait default (sdk lib.rs:324-326): buffered tail is dropped
        skw::SegmentCut cut;
        if (self->seg.take_tail(&cut) && !transcribe_and_emit(self, em, cut, &err)) return err_result(err);
        return ok_result();
    });
}

void plugin_destroy_instance(CPluginHandle handle) { try { if (handle) delete (WhisperPlugin*)handle; } catch (...) {} }

const CNativePluginAPI kApi = {STREAMKIT_NATIVE_PLUGIN_API_VERSION, plugin_get_metadata, plugin_create_instance, plugin_process_packet,
                               plugin_update_params, plugin_flush, plugin_destroy_instance};
}  // namespace

extern "C" const CNativePluginAPI* streamkit_native_plugin_api(void) { return &kApi; }
// additive, for tests and bench.py: how often this library loaded a model file / found one cached since it was loaded itself (W5)
extern "C" void skw_whisper_plugin_cache_stats(int* model_loads, int* cache_hits) {
    if (model_loads) *model_loads = g_model_loads.load();
    if (cache_hits) *cache_hits = g_cache_hits.load();
}
// additive, for tests: cached engines (models resident), how many of them hold a batch workspace right now, and live plugin instances over all of them
extern "C" void skw_whisper_plugin_workspace_stats(int* engines, int* workspaces, int* instances) {
    EngineCache& cache = engine_cache();
    std::lock_guard<std::mutex> l(cache.mu);
    int e = 0, w = 0, n = 0;
    for (auto& kv : cache.map) { std::lock_guard<std::mutex> lw(kv.second->ws_mu); ++e; w += kv.second->ctx != nullptr; n += kv.second->live_instances; }
    if (engines) *engines = e; if (workspaces) *workspaces = w; if (instances) *instances = n;
}
