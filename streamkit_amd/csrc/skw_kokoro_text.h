// skw_kokoro_text.h — the text front end of the Kokoro TTS node, restated from the reference's Rust:
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:444-492  process: Text | Binary(UTF-8) -> sanitize -> append '.' -> accumulate -> sentences
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:696-731  sanitize_text
//   /root/reference/plugins/native/kokoro/src/kokoro_node.rs:546-559  text_preview
//   /root/reference/plugins/native/kokoro/src/sentence_splitter.rs:15-58  SentenceSplitter::extract_sentence / flush
// Pure host code on UTF-8 byte strings; Rust's `char` is a Unicode scalar value, so everything here walks code points.
// The reference's own unit-test vectors for the splitter (sentence_splitter.rs:65-95) are reproduced in tests/golden/kokoro_splitter_vectors.json
// and run against this file in tests/test_cpu_kokoro.py.
#pragma once
#include "skw_segmenter.h"   // utf8 helpers, json_quote, is_ws_cp
#include <string>

namespace skw {
namespace kokoro {

inline unsigned utf8_next(const std::string& s, size_t* i) {      // valid UTF-8 assumed (checked by the caller: String::from_utf8)
    const unsigned char* p = (const unsigned char*)s.data(); const unsigned char c = p[*i];
    const int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : 4;
    unsigned cp = len == 1 ? c : c & (0xFF >> (len + 1));
    for (int k = 1; k < len; ++k) cp = (cp << 6) | (p[*i + k] & 0x3F);
    *i += len; return cp;
}
inline void utf8_put(std::string* o, unsigned cp) {
    if (cp < 0x80) o->push_back((char)cp);
    else if (cp < 0x800) { o->push_back((char)(0xC0 | (cp >> 6))); o->push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) { o->push_back((char)(0xE0 | (cp >> 12))); o->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o->push_back((char)(0x80 | (cp & 0x3F))); }
    else { o->push_back((char)(0xF0 | (cp >> 18))); o->push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o->push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o->push_back((char)(0x80 | (cp & 0x3F))); }
}

// kokoro_node.rs:696-731.  Kept: ASCII letters / digits, space . , ! ? - ' " \n : ;, U+00E0..=U+00FF ('à'..='ÿ'), U+00C0..=U+0178 ('À'..='Ÿ'),
// CJK U+4E00..=U+9FFF and the full-width marks 。，！？、；：（）; any other White_Space becomes ' '; everything else is dropped.  Then
// `.split_whitespace().collect::<Vec<_>>().join(" ")`: runs of white space (the kept '\n' included) collapse to one space, ends trimmed.
inline bool sanitize_keeps(unsigned c) {
    if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9')) return true;
    switch (c) { case ' ': case '.': case ',': case '!': case '?': case '-': case '\'': case '"': case '\n': case ':': case ';': return true; default: break; }
    if ((c >= 0xE0 && c <= 0xFF) || (c >= 0xC0 && c <= 0x178) || (c >= 0x4E00 && c <= 0x9FFF)) return true;
    switch (c) { case 0x3002: case 0xFF0C: case 0xFF01: case 0xFF1F: case 0x3001: case 0xFF1B: case 0xFF1A: case 0xFF08: case 0xFF09: return true; default: break; }
    return false;
}
inline std::string sanitize_text(const std::string& text) {
    std::string kept;
    for (size_t i = 0; i < text.size();) { const unsigned c = utf8_next(text, &i); if (sanitize_keeps(c)) utf8_put(&kept, c); else if (is_ws_cp(c)) kept.push_back(' '); }
    std::string out; bool in_word = false;
    for (size_t i = 0; i < kept.size();) {
        const size_t at = i; const unsigned c = utf8_next(kept, &i);
        if (is_ws_cp(c)) { in_word = false; continue; }
        if (!in_word && !out.empty()) out.push_back(' ');
        in_word = true; out.append(kept, at, i - at);
    }
    return out;
}

inline bool ends_with(const std::string& s, const char* suf) { const size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0; }
// the six sentence-final marks of kokoro_node.rs:468-473 and sentence_splitter.rs:36-41: . ! ? 。 ！ ？
inline bool ends_with_final_punct(const std::string& s) {
    return ends_with(s, ".") || ends_with(s, "!") || ends_with(s, "?") || ends_with(s, "\xE3\x80\x82") || ends_with(s, "\xEF\xBC\x81") || ends_with(s, "\xEF\xBC\x9F");
}

// sentence_splitter.rs:5-58.  Note what the reference does, not what one might expect: the boundary strings are tried IN LIST ORDER and the
// first one that occurs ANYWHERE in the buffer cuts it (". " wins over an earlier "! "), `len` is bytes, and the fall-through case hands
// over the whole buffer untrimmed.
struct SentenceSplitter {
    size_t min_length = 10;
    explicit SentenceSplitter(size_t n = 10) : min_length(n) {}
    bool extract_sentence(std::string* buffer, std::string* sentence) const {
        if (buffer->size() < min_length) return false;
        static const char* const boundaries[] = {". ", ".\n", "! ", "!\n", "? ", "?\n", "\xE3\x80\x82", "\xEF\xBC\x81", "\xEF\xBC\x9F"};
        for (const char* b : boundaries) {
            const size_t pos = buffer->find(b);
            if (pos != std::string::npos) {
                const size_t end_pos = pos + strlen(b);
                *sentence = utf8_trim(buffer->substr(0, end_pos));
                buffer->erase(0, end_pos);
                return true;
            }
        }
        if (ends_with_final_punct(*buffer)) { *sentence = std::move(*buffer); buffer->clear(); return true; }
        return false;
    }
    static bool flush(std::string* buffer, std::string* out) { if (buffer->empty()) return false; *out = std::move(*buffer); buffer->clear(); return true; }
};

// kokoro_node.rs:546-559: the first max_chars CHARACTERS, "..." appended when more follow; max_chars == 0 -> none (JSON null)
inline bool text_preview(const std::string& text, size_t max_chars, std::string* out) {
    if (max_chars == 0) return false;
    size_t i = 0, n = 0;
    while (i < text.size() && n < max_chars) { utf8_next(text, &i); ++n; }
    *out = text.substr(0, i); if (i < text.size()) *out += "...";
    return true;
}

// The Transcription -> Text step of the voice-agent pipelines (samples/pipelines/dynamic/voice-agent-openai.yaml:86-95): a core::script node takes
// `packet.data.text` of a Transcription, `String(text || '').trim()`, and drops empty results; what it hands on is Text.  (The LLM call in
// between is a network service and out of scope; config 5 of BASELINE.json is STT -> TTS.)
inline bool transcription_to_text(const std::string& transcription_json, std::string* text) {
    JsonValue v; std::string err;
    if (!json_parse(transcription_json.c_str(), &v, &err) || v.type != JsonValue::Object) return false;
    const JsonValue* t = v.get("text");
    if (!t || t->type != JsonValue::String) return false;
    *text = utf8_trim(t->str);
    return !text->empty();
}

}  // namespace kokoro
}  // namespace skw
