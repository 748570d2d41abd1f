// skw_segmenter.h — host-side logic of the Whisper node that is NOT arithmetic on the GPU:
// the 512-sample VAD framing + speech/silence state machine (W1-W3), JSON helpers, UTF-8 helpers.
// Restates /root/reference/plugins/native/whisper/src/lib.rs:404-494 (process) and :582-612 (segment hand-off).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <vector>

namespace skw {

// ------------------------------------------------------------------ JSON (params in, serde_json-shaped text out)
struct JsonValue {
    enum Type { Null, Bool, Number, String, Array, Object } type = Null;
    bool b = false; double num = 0; std::string str; std::vector<JsonValue> arr; std::vector<std::pair<std::string, JsonValue>> obj;
    const JsonValue* get(const char* k) const { const JsonValue* r = nullptr; for (auto& kv : obj) if (kv.first == k) r = &kv.second; return r; } // last duplicate wins, as serde_json
};
namespace detail {
struct P { const char* s; const char* e; std::string* err; };
inline void ws(P& p) { while (p.s < p.e && (*p.s == ' ' || *p.s == '\t' || *p.s == '\n' || *p.s == '\r')) ++p.s; }
inline bool fail(P& p, const char* m) { if (p.err->empty()) *p.err = m; return false; }
inline void put_utf8(std::string& o, unsigned cp) {
    if (cp < 0x80) o += (char)cp; else if (cp < 0x800) { o += (char)(0xC0 | (cp >> 6)); o += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { o += (char)(0xE0 | (cp >> 12)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
    else { o += (char)(0xF0 | (cp >> 18)); o += (char)(0x80 | ((cp >> 12) & 0x3F)); o += (char)(0x80 | ((cp >> 6) & 0x3F)); o += (char)(0x80 | (cp & 0x3F)); }
}
inline bool hex4(P& p, unsigned* out) { if (p.e - p.s < 4) return false; unsigned v = 0; for (int i = 0; i < 4; ++i) { char c = p.s[i];
v <<= 4; if (c >= '0' && c <= '9') v |= c - '0'; else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10; else return false; } p.s += 4; *out = v; return true; }
inline bool str(P& p, std::string* out) {
    if (p.s >= p.e || *p.s != '"') return fail(p, "expected string"); ++p.s;
    while (p.s < p.e && *p.s != '"') {
        unsigned char c = (unsigned char)*p.s++;
        if (c < 0x20) return fail(p, "control character in string");
        if (c != '\\') { *out += (char)c; continue; }
        if (p.s >= p.e) return fail(p, "EOF in string");
        char e = *p.s++;
        switch (e) {
            case '"': *out += '"'; break; case '\\': *out += '\\'; break; case '/': *out += '/'; break; case 'b': *out += '\b'; break; case 'f': *out += '\f'; break;
            case 'n': *out += '\n'; break; case 'r': *out += '\r'; break; case 't': *out += '\t'; break;
            case 'u': { unsigned cp; if (!hex4(p, &cp)) return fail(p, "invalid unicode escape");
                if (cp >= 0xD800 && cp < 0xDC00) { unsigned lo; if (p.e - p.s < 6 || p.s[0] != '\\' || p.s[1] != 'u') return fail(p, "lone surrogate");
                p.s += 2; if (!hex4(p, &lo) || lo < 0xDC00 || lo > 0xDFFF) return fail(p, "invalid surrogate");
                cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00); }
                else if (cp >= 0xDC00 && cp < 0xE000) return fail(p, "lone surrogate");
                put_utf8(*out, cp); break; }
            default: return fail(p, "invalid escape");
        }
    }
    if (p.s >= p.e) return fail(p, "EOF while parsing a string"); ++p.s; return true;
}
inline bool value(P& p, JsonValue* v, int depth) {
    if (depth > 64) return fail(p, "recursion limit exceeded");
    ws(p); if (p.s >= p.e) return fail(p, "EOF while parsing a value");
    char c = *p.s;
    if (c == '{') { ++p.s; v->type = JsonValue::Object; ws(p); if (p.s < p.e && *p.s == '}') { ++p.s; return true; }
        for (;;) { ws(p); std::string k; if (!str(p, &k)) return false; ws(p); if (p.s >= p.e || *p.s != ':') return fail(p, "expected `:`");
        ++p.s; JsonValue x; if (!value(p, &x, depth + 1)) return false; v->obj.emplace_back(std::move(k), std::move(x)); ws(p);
            if (p.s < p.e && *p.s == ',') { ++p.s; continue; } if (p.s < p.e && *p.s == '}') { ++p.s; return true; } return fail(p, "expected `,` or `}`"); } }
    if (c == '[') { ++p.s; v->type = JsonValue::Array; ws(p); if (p.s < p.e && *p.s == ']') { ++p.s; return true; }
        for (;;) { JsonValue x; if (!value(p, &x, depth + 1)) return false; v->arr.push_back(std::move(x)); ws(p); if (p.s < p.e && *p.s == ',') { ++p.s;
        continue; } if (p.s < p.e && *p.s == ']') { ++p.s; return true; } return fail(p, "expected `,` or `]`"); } }
    if (c == '"') { v->type = JsonValue::String; return str(p, &v->str); }
    if (c == 't' && p.e - p.s >= 4 && !strncmp(p.s, "true", 4)) { p.s += 4; v->type = JsonValue::Bool; v->b = true; return true; }
    if (c == 'f' && p.e - p.s >= 5 && !strncmp(p.s, "false", 5)) { p.s += 5; v->type = JsonValue::Bool; v->b = false; return true; }
    if (c == 'n' && p.e - p.s >= 4 && !strncmp(p.s, "null", 4)) { p.s += 4; v->type = JsonValue::Null; return true; }
    if (c == '-' || (c >= '0' && c <= '9')) { const char* b = p.s; if (*p.s == '-') ++p.s; if (p.s >= p.e || !(*p.s >= '0' && *p.s <= '9')) return fail(p, "invalid number");
        while (p.s < p.e && ((*p.s >= '0' && *p.s <= '9') || *p.s == '.' || *p.s == 'e' || *p.s == 'E' || *p.s == '+' || *p.s == '-')) ++p.s;
        std::string t(b, p.s); char* endp = nullptr; v->num = strtod(t.c_str(), &endp); if (!endp || *endp) return fail(p, "invalid number"); v->type = JsonValue::Number; return true; }
    return fail(p, "expected value");
}
}  // namespace detail
inline bool json_parse(const char* text, JsonValue* out, std::string* err) {
    detail::P p{text, text + strlen(text), err}; if (!detail::value(p, out, 0)) return false; detail::ws(p); if (p.s != p.e) { *err = "trailing characters"; return false; } return true;
}
// serde_json string escaping (format_escaped_str): \" \\ \b \f \n \r \t, other controls as \u00XX, everything else verbatim
inline std::string json_quote(const std::string& s) {
    std::string o = "\""; char buf[8];
    for (unsigned char c : s) {
        switch (c) { case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break;
        case '\n': o += "\\n"; break; case '\r': o += "\\r"; break; case '\t': o += "\\t"; break;
            default: if (c < 0x20) { snprintf(buf, sizeof buf, "\\u%04x", c); o += buf; } else o += (char)c; }
    }
    return o + "\"";
}
// an f32 as serde_json prints it without arbitrary_precision: widened to f64, shortest round-trip digits, always a fraction or exponent
// an f32 as serde_json writes it inside a `json!` value: widened to f64 (Value holds f64), then ryu's shortest round-trip digits in ryu's layout — plain decimals while the
// decimal point lies within 16 digits to the right or 5 zeros to the left of the first digit ("0.5", "0.699999988079071", "100000.0", "0.00001"), else d[.ddd]e[-]x without
// padding or plus sign ("1e-6", "1.5e16").  (Rounds 2-4 used printf's %g layout, which agrees on every value a threshold or a speech probability takes and differs outside.)
inline std::string json_f32(float f) {
    double d = (double)f; if (!std::isfinite(d)) return "null";
    if (d == 0.0) return std::signbit(d) ? "-0.0" : "0.0";
    char buf[48]; int prec = 0; for (prec = 0; prec <= 16; ++prec) { snprintf(buf, sizeof buf, "%.*e", prec, std::fabs(d)); if (strtod(buf, nullptr) == std::fabs(d)) break; }
    std::string digits; int e10 = 0;
    { const char* p = buf; for (; *p && *p != 'e'; ++p) if (*p >= '0' && *p <= '9') digits.push_back(*p); e10 = atoi(p + 1); }
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    const int length = (int)digits.size(), k = e10 - (length - 1), kk = length + k;
    std::string r = d < 0 ? "-" : "";
    if (0 <= k && kk <= 16) r += digits + std::string((size_t)k, '0') + ".0";
    else if (0 < kk && kk <= 16) r += digits.substr(0, (size_t)kk) + "." + digits.substr((size_t)kk);
    else if (-5 < kk && kk <= 0) r += "0." + std::string((size_t)(-kk), '0') + digits;
    else if (length == 1) r += digits + "e" + std::to_string(kk - 1);
    else r += digits.substr(0, 1) + "." + digits.substr(1) + "e" + std::to_string(kk - 1);
    return r;
}

// ------------------------------------------------------------------ UTF-8
inline bool utf8_valid(const std::string& s) {
    const unsigned char* p = (const unsigned char*)s.data(); size_t n = s.size(), i = 0;
    while (i < n) { unsigned char c = p[i]; int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 0; if (!len || i + len > n) return false;
        unsigned cp = len == 1 ? c : c & (0xFF >> (len + 1)); for (int k = 1; k < len; ++k) { if ((p[i + k] & 0xC0) != 0x80) return false; cp = (cp << 6) | (p[i + k] & 0x3F); }
        if ((len == 2 && cp < 0x80) || (len == 3 && cp < 0x800) || (len == 4 && cp < 0x10000) || cp > 0x10FFFF || (cp >= 0xD800 && cp < 0xE000)) return false; i += len; }
    return true;
}
inline bool is_ws_cp(unsigned cp) { return (cp >= 9 && cp <= 13) || cp == 0x20 || cp == 0x85 || cp == 0xA0 || cp == 0x1680 || (cp >= 0x2000 && cp <= 0x200A)
    || cp == 0x2028 || cp == 0x2029 || cp == 0x202F || cp == 0x205F || cp == 0x3000;
}
// str::trim() on valid UTF-8 (White_Space code points)
inline std::string utf8_trim(const std::string& s) {
    const unsigned char* p = (const unsigned char*)s.data(); size_t n = s.size(), b = 0, e = n;
    while (b < n) { unsigned char c = p[b]; int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : 4;
    unsigned cp = len == 1 ? c : c & (0xFF >> (len + 1)); for (int k = 1; k < len; ++k) cp = (cp << 6) | (p[b + k] & 0x3F);
    if (!is_ws_cp(cp)) break; b += len; }
    while (e > b) { size_t st = e - 1; while (st > b && (p[st] & 0xC0) == 0x80) --st;
    unsigned char c = p[st]; int len = (int)(e - st); unsigned cp = len == 1 ? c : c & (0xFF >> (len + 1));
    for (int k = 1; k < len; ++k) cp = (cp << 6) | (p[st + k] & 0x3F); if (!is_ws_cp(cp)) break; e = st; }
    return s.substr(b, e - b);
}

// ------------------------------------------------------------------ VAD gate (vad.rs:67-120 interface: 512 samples -> probability)
struct Vad { virtual ~Vad() {} virtual float process_chunk(const float* frame512) = 0; virtual void reset() {} };
struct AlwaysSpeechVad : Vad { float process_chunk(const float*) override { return 1.0f; } };
// energy gate: p = rms / (rms + 0.01)  (p >= 0.5  <=>  rms >= 0.01, about -40 dBFS)
struct EnergyVad : Vad { float process_chunk(const float* f) override { float s = 0.0f; for (int i = 0; i < 512; ++i) s += f[i] * f[i]; float rms = sqrtf(s / 512.0f); return rms / (rms + 0.01f); } };

// ------------------------------------------------------------------ segmentation state machine (lib.rs:404-494)
struct SpeechStart { std::string segment_id; uint64_t start_time_ms; float probability; };
struct SegmentCut {
    std::vector<float> samples; uint64_t start_time_ms = 0, end_time_ms = 0; const char* reason = ""; bool has_silence_duration = false; uint64_t silence_duration_ms = 0;
    std::string segment_id;
};
class Segmenter {
public:
    void configure(float threshold, uint64_t min_silence_ms, float max_secs) { threshold_ = threshold; set_min_silence_ms(min_silence_ms); max_secs_ = max_secs; }
    void set_threshold(float t) { threshold_ = t; }
    void set_min_silence_ms(uint64_t ms) { silence_threshold_frames_ = (size_t)(ms / 32); }   // lib.rs:386, 571
    void set_max_duration_secs(float s) { max_secs_ = s; }
    // feeds samples; on_cut returns false to abort (the error propagates like `?` in lib.rs:464, 478)
    void push(const float* samples, size_t n, Vad& vad, const std::function<void(const SpeechStart&)>& on_start,
              const std::function<bool(const SegmentCut&)>& on_cut, std::string* err) {
        frame_buffer_.insert(frame_buffer_.end(), samples, samples + n);
        float frame[512];
        while (frame_buffer_.size() >= 512) {
            for (int i = 0; i < 512; ++i) frame[i] = frame_buffer_[i];
            frame_buffer_.erase(frame_buffer_.begin(), frame_buffer_.begin() + 512);
            const float probability = vad.process_chunk(frame);
            const bool is_speech = probability >= threshold_;
            if (is_speech) {
                silence_frame_count_ = 0;
                if (speech_buffer_.empty()) {
                    segment_start_time_ms_ = absolute_time_ms_;
                    if (segment_counter_ != UINT64_MAX) segment_counter_++;
                    current_segment_id_ = "seg-" + std::to_string(segment_start_time_ms_) + "-" + std::to_string(segment_counter_);
                    on_start(SpeechStart{current_segment_id_, segment_start_time_ms_, probability});
                }
                speech_buffer_.insert(speech_buffer_.end(), frame, frame + 512);
                const uint64_t segment_duration_ms = absolute_time_ms_ - segment_start_time_ms_;
                const uint64_t max_duration_ms = (uint64_t)(max_secs_ * 1000.0f);
                if (segment_duration_ms >= max_duration_ms) {
                    const uint64_t end_time_ms = absolute_time_ms_ + 32;
                    if (!cut(on_cut, end_time_ms, "max_duration", false, 0)) { (void)err; return; }
                }
            } else {
                silence_frame_count_ += 1;
                if (!speech_buffer_.empty() && silence_frame_count_ >= silence_threshold_frames_) {
                    const uint64_t silence_frames = silence_frame_count_ > 0 ? (uint64_t)silence_frame_count_ - 1 : 0;
                    const uint64_t back = silence_frames * 32;
                    const uint64_t end_time_ms = absolute_time_ms_ >= back ? absolute_time_ms_ - back : 0;
                    if (!cut(on_cut, end_time_ms, "silence", true, (uint64_t)silence_frame_count_ * 32)) return;
                }
            }
            absolute_time_ms_ += 32;   // 512 samples @ 16 kHz
        }
    }
    // additive (flush_tail): hand over whatever speech is buffered when the stream ends
    // (while a segment is open, the < 512 samples that have not filled a VAD frame yet belong to it too)
    bool take_tail(SegmentCut* out) {
        if (speech_buffer_.empty()) return false;
        speech_buffer_.insert(speech_buffer_.end(), frame_buffer_.begin(), frame_buffer_.end()); frame_buffer_.clear();
        out->samples.swap(speech_buffer_); speech_buffer_.clear(); out->start_time_ms = segment_start_time_ms_;
        out->end_time_ms = absolute_time_ms_; out->reason = "flush"; out->segment_id = current_segment_id_;
        current_segment_id_.clear(); silence_frame_count_ = 0; return true;
    }
    uint64_t absolute_time_ms() const { return absolute_time_ms_; }
    size_t buffered_speech_samples() const { return speech_buffer_.size(); }
private:
    bool cut(const std::function<bool(const SegmentCut&)>& on_cut, uint64_t end_time_ms, const char* reason, bool has_sil, uint64_t sil_ms) {
        if (speech_buffer_.empty()) return true;   // lib.rs:589-591
        SegmentCut c; c.samples.swap(speech_buffer_); speech_buffer_.clear(); c.start_time_ms = segment_start_time_ms_; c.end_time_ms = end_time_ms; c.reason = reason;
        c.has_silence_duration = has_sil; c.silence_duration_ms = sil_ms; c.segment_id = current_segment_id_; current_segment_id_.clear();
        if (!on_cut(c)) return false;
        silence_frame_count_ = 0;                  // lib.rs:699
        return true;
    }
    float threshold_ = 0.5f, max_secs_ = 30.0f; size_t silence_threshold_frames_ = 21;
    std::deque<float> frame_buffer_; std::vector<float> speech_buffer_;
    uint64_t segment_start_time_ms_ = 0, segment_counter_ = 0, absolute_time_ms_ = 0; std::string current_segment_id_; size_t silence_frame_count_ = 0;
};

}  // namespace skw
