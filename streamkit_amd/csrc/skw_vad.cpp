// libskw_vad.so — the C ABI of include/skw_vad.h over the header-only Silero implementation (skw_silero.h).
#include "../../include/skw_vad.h"
#include "skw_silero.h"

struct skw_vad { std::shared_ptr<const skw::SileroWeights> w; skw::SileroVad v; explicit skw_vad(std::shared_ptr<const skw::SileroWeights> ww) : w(ww), v(ww) {} };

extern "C" skw_vad* skw_vad_create(const char* path, char* err, size_t errlen) {
    try {
        auto w = std::make_shared<skw::SileroWeights>(); std::string e;
        if (!path || !skw::SileroVad::load_weights(path, w.get(), &e)) { if (err && errlen) snprintf(err, errlen, "%s", path ? e.c_str() : "Failed to load VAD model from '': no path");
        return nullptr; }
        return new skw_vad(w);
    } catch (const std::exception& ex) { if (err && errlen) snprintf(err, errlen, "Failed to load VAD model from '%s': %s", path ? path : "", ex.what()); return nullptr; }
}
extern "C" int skw_vad_process_chunk(skw_vad* v, const float* frame512, float* probability) { if (!v || !frame512 || !probability) return -1; *probability = v->v.process_chunk(frame512); return 0; }
extern "C" void skw_vad_reset(skw_vad* v) { if (v) v->v.reset(); }
extern "C" void skw_vad_state(const skw_vad* v, float* out256) { memcpy(out256, v->v.state_h(), sizeof(float) * 128); memcpy(out256 + 128, v->v.state_c(), sizeof(float) * 128); }
extern "C" void skw_vad_free(skw_vad* v) { delete v; }
