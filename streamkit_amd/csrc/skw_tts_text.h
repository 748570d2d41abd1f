// skw_tts_text.h — the synthesiser's text front end (libskw_tts.so): tokens.txt / lexicon files and text -> token ids, as sherpa-onnx's Kokoro front end does it
// for the files the reference hands over (kokoro_node.rs:741-766): words found in a lexicon become their phoneme ids, everything else goes code point by code
// point through tokens.txt (unknown symbols dropped); pad id 0 at both ends.  (No espeak-ng phonemiser in this build: INTEGRATION.md section G.)
#ifndef SKW_TTS_TEXT_H
#define SKW_TTS_TEXT_H
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>
#define TTS_TEXT_MAX_TOKENS 510
struct TtsText {
    std::map<unsigned, int> sym2id;                              // code point -> id
    std::map<std::string, std::vector<int>> lexicon;             // lower-case word -> ids
};
static bool read_file(const char* path, std::vector<uint8_t>* out, size_t limit) {
    FILE* f = fopen(path, "rb"); if (!f) return false; uint8_t buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) { out->insert(out->end(), buf, buf + n); if (out->size() > limit) { fclose(f); return false; } }
    fclose(f); return true;
}
static unsigned next_cp(const std::string& s, size_t* i) {
    const unsigned char* p = (const unsigned char*)s.data(); const unsigned char c = p[*i]; int len = c < 0x80 ? 1 : (c >> 5) == 6 ? 2 : (c >> 4) == 14 ? 3 : (c >> 3) == 30 ? 4 : 1;
    if (*i + len > s.size()) len = 1; unsigned cp = len == 1 ? c : c & (0xFF >> (len + 1)); for (int k = 1; k < len; ++k) cp = (cp << 6) | (p[*i + k] & 0x3F); *i += len; return cp;
}
// tokens.txt: "<symbol> <id>" per line; a line that starts with a space names the space symbol (sherpa-onnx's convention)
static bool load_tokens(TtsText* t, const char* path, std::string* err) {
    std::vector<uint8_t> b; if (!read_file(path, &b, 16u << 20)) { *err = std::string("cannot read tokens file ") + path; return false; }
    std::string s((const char*)b.data(), b.size()); size_t i = 0;
    while (i < s.size()) {
        size_t e = s.find('\n', i); if (e == std::string::npos) e = s.size(); std::string line = s.substr(i, e - i); i = e + 1;
        if (!line.empty() && line.back() == '\r') line.pop_back(); if (line.empty()) continue;
        const size_t sp = line.rfind(' '); if (sp == std::string::npos) continue;
        std::string sym = line.substr(0, sp); const int id = atoi(line.c_str() + sp + 1); if (sym.empty()) sym = " ";
        size_t k = 0; const unsigned cp = next_cp(sym, &k); if (k == sym.size()) t->sym2id[cp] = id;       // single-code-point symbols (all of Kokoro's are)
    }
    if (t->sym2id.empty()) { *err = std::string("no symbols in tokens file ") + path; return false; }
    return true;
}
static std::string lower_ascii(std::string s) { for (auto& c : s) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a'); return s; }
static void load_lexicon(TtsText* t, const char* list) {      // "word ph ph ..." per line; the phonemes are symbols of tokens.txt
    if (!list) return; std::string all = list; size_t i = 0;
    while (i <= all.size()) {
        size_t e = all.find(',', i); if (e == std::string::npos) e = all.size(); const std::string path = all.substr(i, e - i); i = e + 1; if (path.empty()) continue;
        std::vector<uint8_t> b; if (!read_file(path.c_str(), &b, 256u << 20)) continue;      // missing lexicon files are not an error (kokoro_node.rs never checks them)
        std::string s((const char*)b.data(), b.size()); size_t j = 0;
        while (j < s.size()) {
            size_t le = s.find('\n', j); if (le == std::string::npos) le = s.size(); std::string line = s.substr(j, le - j); j = le + 1;
            const size_t sp = line.find_first_of(" \t"); if (sp == std::string::npos || sp == 0) continue;
            const std::string word = lower_ascii(line.substr(0, sp)); if (t->lexicon.count(word)) continue;       // first entry wins
            std::vector<int> ids; for (size_t k = sp; k < line.size();) { const unsigned cp = next_cp(line, &k);
            if (cp == ' ' || cp == '\t' || cp == '\r') continue; auto it = t->sym2id.find(cp); if (it != t->sym2id.end()) ids.push_back(it->second); }
            if (!ids.empty()) t->lexicon[word] = ids;
        }
    }
}
// text -> ids: words found in the lexicon become their phoneme ids, everything else goes code point by code point through tokens.txt
// (unknown symbols are dropped); pad id 0 at both ends; at most TTS_TEXT_MAX_TOKENS
static std::vector<int> tokenize(const TtsText* t, const std::string& text) {
    std::vector<int> ids; ids.push_back(0);
    size_t i = 0;
    while (i < text.size() && (int)ids.size() < TTS_TEXT_MAX_TOKENS - 1) {
        size_t j = i; std::string word;
        while (j < text.size()) { const unsigned char c = (unsigned char)text[j]; if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '\'') { word.push_back((char)c); ++j; } else break; }
        if (!word.empty()) {
            auto it = t->lexicon.find(lower_ascii(word));
            if (it != t->lexicon.end()) { for (int id : it->second) if ((int)ids.size() < TTS_TEXT_MAX_TOKENS - 1) ids.push_back(id); i = j; continue; }
        }
        const unsigned cp = next_cp(text, &i);
        auto it = t->sym2id.find(cp); if (it == t->sym2id.end() && cp >= 'A' && cp <= 'Z') it = t->sym2id.find(cp - 'A' + 'a');
        if (it != t->sym2id.end()) ids.push_back(it->second);
    }
    ids.push_back(0); return ids;
}

#endif
