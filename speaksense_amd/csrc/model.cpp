// ggml legacy model-file reader for the product path.
// Replaces what `WhisperContext::new_with_params` does for the reference (/root/reference/src/asr/whisper.rs:21-28),
// i.e. whisper.cpp's whisper_model_load: magic, 11 hparams, mel filterbank, vocabulary (+ synthesised special
// tokens), then {n_dims, name_len, type, ne[], name, data} records.  ftype 0/1 (f32 / f16)
// files are supported; quantised files are rejected with SS_ERR_MODEL.
#include "common.h"

#include <algorithm>
#include <cstring>

namespace ss {

static const char* const kLang[] = {"en","zh","de","es","ru","ko","fr","ja","pt","tr","pl","ca","nl","ar","sv","it","id","hi","fi","vi",
    "he","uk","el","ms","cs","ro","da","hu","ta","no","th","ur","hr","bg","lt","la","mi","ml","cy","sk","te","fa","lv","bn","sr","az",
    "sl","kn","et","mk","br","eu","is","hy","ne","mn","bs","kk","sq","sw","gl","mr","pa","si","km","sn","yo","so","af","oc","ka","be",
    "tg","sd","gu","am","yi","lo","uz","fo","ht","ps","tk","nn","mt","sa","lb","my","bo","tl","mg","as","tt","haw","ln","ha","ba","jw","su","yue"};
static const int kNLang = sizeof(kLang) / sizeof(kLang[0]);

int lang_id(const char* code) {
    for (int i = 0; i < kNLang; i++)
        if (!strcmp(code, kLang[i])) return i;
    return -1;
}

const char* lang_code(int id) { return id >= 0 && id < kNLang ? kLang[id] : nullptr; }

// whisper.cpp `tokenize(vocab, text)` (what whisper_tokenize / initial_prompt use): words = successive matches of the GPT-2 pattern
//   's|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+
// (std::regex, classic locale: alpha/digit/space are ASCII classes, every byte >= 0x80 is "other"), then each word is cut greedily into the
// longest vocabulary entries; a byte no entry starts with is skipped.  Written out by hand: the alternatives are tried in order at each position.
static inline bool is_alpha_c(unsigned char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
// whisper_full_params.suppress_non_speech_tokens (whisper.cpp v1.5.x whisper_process_logits, list after openai/whisper tokenizer.py non_speech_tokens):
// punctuation that annotates rather than transcribes, brackets and their runs, music notes; each with and without a leading space.  " -" and " '"
// are on the list only with the space: hyphens and apostrophes stay allowed inside words.
std::vector<int> non_speech_token_ids(const Vocab& vocab) {
    static const char* const kSymbols[] = {
        "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}", "~",
        "\xe3\x80\x8c", "\xe3\x80\x8d", "\xe3\x80\x8e", "\xe3\x80\x8f",                       // corner brackets U+300C..U+300F
        "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((", ")))", "[[", "]]", "{{", "}}",
        "\xe2\x99\xaa\xe2\x99\xaa", "\xe2\x99\xaa\xe2\x99\xaa\xe2\x99\xaa",                 // two and three eighth notes
        "\xe2\x99\xa9", "\xe2\x99\xaa", "\xe2\x99\xab", "\xe2\x99\xac", "\xe2\x99\xad", "\xe2\x99\xae", "\xe2\x99\xaf"};   // U+2669..U+266F
    std::vector<int> ids;
    auto take = [&](const std::string& t) {
        auto it = vocab.token_to_id.find(t);
        if (it != vocab.token_to_id.end()) ids.push_back(it->second);
    };
    for (const char* sym : kSymbols) { take(sym); take(std::string(" ") + sym); }
    take(" -"); take(" '");
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    return ids;
}

static inline bool is_digit_c(unsigned char c) { return c >= '0' && c <= '9'; }
static inline bool is_space_c(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
static inline bool is_other_c(unsigned char c) { return !is_alpha_c(c) && !is_digit_c(c) && !is_space_c(c); }
std::vector<int> tokenize(const Vocab& vocab, const std::string& text) {
    std::vector<std::string> words;
    const size_t n = text.size();
    size_t i = 0;
    auto run = [&](size_t from, bool (*cls)(unsigned char)) { size_t j = from; while (j < n && cls((unsigned char)text[j])) j++; return j; };
    while (i < n) {
        size_t e = 0;   // end of the match starting at i (0 = none yet)
        if (text[i] == '\'') {
            static const char* const suf[] = {"'s", "'t", "'re", "'ve", "'m", "'ll", "'d"};
            for (const char* sfx : suf) { const size_t l = strlen(sfx); if (text.compare(i, l, sfx) == 0) { e = i + l; break; } }
        }
        if (!e) {
            const size_t b = (text[i] == ' ' && i + 1 < n) ? i + 1 : i;   // " ?": with the space first, then (backtracking) without it
            bool (*const classes[3])(unsigned char) = {is_alpha_c, is_digit_c, is_other_c};
            for (auto cls : classes) {
                size_t j = run(b, cls);
                if (j > b) { e = j; break; }
                if (b != i) { j = run(i, cls); if (j > i) { e = j; break; } }
            }
        }
        if (!e) {   // \s+(?!\S) | \s+
            const size_t j = run(i, is_space_c);
            if (j > i) e = (j < n && j - i >= 2) ? j - 1 : j;
        }
        if (!e) e = i + 1;   // unreachable: every byte is in one of the classes
        words.push_back(text.substr(i, e - i));
        i = e;
    }
    std::vector<int> tokens;
    for (const std::string& word : words) {
        size_t a = 0;
        const size_t m = word.size();
        while (a < m) {
            size_t j = m;
            bool found = false;
            while (j > a) {
                auto it = vocab.token_to_id.find(word.substr(a, j - a));
                if (it != vocab.token_to_id.end()) { tokens.push_back(it->second); a = j; found = true; break; }
                --j;
            }
            if (!found) ++a;
        }
    }
    return tokens;
}

const HostTensor& HostModel::get(const std::string& name) const {
    auto it = t.find(name);
    if (it == t.end()) throw Error(-2, "model: missing tensor " + name);
    return it->second;
}

static inline float half_to_float(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ff, u;
    if (exp == 0) {
        if (man == 0) u = sign;
        else {
            int e = -1;
            do { e++; man <<= 1; } while (!(man & 0x400));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
        }
    } else if (exp == 31) u = sign | 0x7f800000u | (man << 13);
    else u = sign | ((exp + 112) << 23) | (man << 13);
    float f; memcpy(&f, &u, 4); return f;
}


// ggml block quantisation (QK = 32), dequantised at load: the files of script/download-ggml-model.sh:28-51 (`*-q5_0`, `*-q5_1`) and the other
// block types whisper.cpp's quantize tool writes.  ttype: 2 q4_0, 3 q4_1, 6 q5_0, 7 q5_1, 8 q8_0; returns bytes per 32-element block (0 = not quantised)
static size_t q_block_bytes(int tt) { return tt == 2 ? 18 : tt == 3 ? 20 : tt == 6 ? 22 : tt == 7 ? 24 : tt == 8 ? 34 : 0; }
static void dequant_block(int tt, const uint8_t* b, float* y, float (*h2f)(uint16_t)) {
    uint16_t dh, mh = 0;
    memcpy(&dh, b, 2);
    const float d = h2f(dh);
    float m = 0.0f;
    if (tt == 3 || tt == 7) { memcpy(&mh, b + 2, 2); m = h2f(mh); }
    if (tt == 8) {                       // q8_0: { f16 d; int8 qs[32] }
        const int8_t* qs = (const int8_t*)(b + 2);
        for (int j = 0; j < 32; j++) y[j] = qs[j] * d;
    } else if (tt == 2) {                // q4_0: { f16 d; u8 qs[16] }: (nibble - 8) * d
        const uint8_t* qs = b + 2;
        for (int j = 0; j < 16; j++) { y[j] = ((int)(qs[j] & 0x0F) - 8) * d; y[j + 16] = ((int)(qs[j] >> 4) - 8) * d; }
    } else if (tt == 3) {                // q4_1: { f16 d; f16 m; u8 qs[16] }: nibble * d + m
        const uint8_t* qs = b + 4;
        for (int j = 0; j < 16; j++) { y[j] = (qs[j] & 0x0F) * d + m; y[j + 16] = (qs[j] >> 4) * d + m; }
    } else {                             // q5_0: { f16 d; u8 qh[4]; u8 qs[16] } / q5_1: { f16 d; f16 m; u8 qh[4]; u8 qs[16] }: the fifth bits live in qh
        const uint8_t* p = b + (tt == 7 ? 4 : 2);
        uint32_t qh;
        memcpy(&qh, p, 4);
        const uint8_t* qs = p + 4;
        for (int j = 0; j < 16; j++) {
            const uint8_t xh0 = ((qh >> (j + 0)) << 4) & 0x10, xh1 = (qh >> (j + 12)) & 0x10;
            const int x0 = (qs[j] & 0x0F) | xh0, x1 = (qs[j] >> 4) | xh1;
            if (tt == 6) { y[j] = (x0 - 16) * d; y[j + 16] = (x1 - 16) * d; }
            else { y[j] = x0 * d + m; y[j + 16] = x1 * d + m; }
        }
    }
}

namespace {
struct File {
    FILE* f;
    explicit File(const char* p) : f(fopen(p, "rb")) {}
    ~File() { if (f) fclose(f); }
    bool rd(void* p, size_t n) { return fread(p, 1, n, f) == n; }
};
}  // namespace

void load_ggml_model(const char* path, HostModel& m, bool vocab_only) {
    File F(path);
    if (!F.f) throw Error(-2, std::string("model: cannot open ") + path);
    uint32_t magic = 0;
    if (!F.rd(&magic, 4) || magic != 0x67676d6c) throw Error(-2, "model: bad magic (not a ggml legacy file)");
    if (!F.rd(&m.hp, sizeof(HParams))) throw Error(-2, "model: truncated header");
    const HParams& hp = m.hp;
    // whisper_model_load: "qntvr = ftype / GGML_QNT_VERSION_FACTOR; ftype %= GGML_QNT_VERSION_FACTOR"; the tensor records carry their own type
    const int ftype = hp.ftype % 1000;
    if (!(ftype == 0 || ftype == 1 || ftype == 2 || ftype == 3 || ftype == 7 || ftype == 8 || ftype == 9))
        throw Error(-2, "model: ftype " + std::to_string(hp.ftype) + " not supported (f32, f16, q4_0, q4_1, q5_0, q5_1, q8_0)");
    // a malformed header must come back as SS_ERR_MODEL, never as SIGFPE: check every field before dividing by any of them
    const int32_t* hv = &hp.n_vocab;
    for (int i = 0; i < 10; i++) if (hv[i] <= 0) throw Error(-2, "model: non-positive hyper-parameter in the header");
    if (hp.n_vocab > (1 << 20) || hp.n_audio_ctx > 1500 || hp.n_text_ctx > 448 || hp.n_mels > 128 || hp.n_audio_layer > 256 || hp.n_text_layer > 256 ||
        hp.n_audio_state > 8192 || hp.n_text_state > 8192)
        throw Error(-2, "model: hyper-parameter out of range");
    if (hp.n_audio_state % hp.n_audio_head || hp.n_text_state % hp.n_text_head || hp.n_audio_state / hp.n_audio_head != 64 ||
        hp.n_text_state / hp.n_text_head != 64)
        throw Error(-2, "model: head dim must be 64");
    int32_t nm = 0, nf = 0;
    if (!F.rd(&nm, 4) || !F.rd(&nf, 4) || nf != kNBins || nm != hp.n_mels) throw Error(-2, "model: bad mel filterbank header");
    m.filt_n_mel = nm; m.filt_n_fft = nf;
    m.filters.resize((size_t)nm * nf);
    if (!F.rd(m.filters.data(), m.filters.size() * 4)) throw Error(-2, "model: truncated filterbank");
    int32_t nv = 0;
    if (!F.rd(&nv, 4) || nv <= 0 || nv > hp.n_vocab) throw Error(-2, "model: bad vocab size");
    Vocab& v = m.vocab;
    v.id_to_token.resize(nv);
    for (int i = 0; i < nv; i++) {
        uint32_t len = 0;
        if (!F.rd(&len, 4) || len > 4096) throw Error(-2, "model: bad vocab entry");
        std::string s(len, '\0');
        if (len && !F.rd(&s[0], len)) throw Error(-2, "model: truncated vocab");
        v.id_to_token[i] = s;
        v.token_to_id[s] = i;
    }
    v.n_vocab = hp.n_vocab;
    if (v.is_multilingual()) {
        v.token_eot++; v.token_sot++;
        const int dt = v.num_languages() - 98;
        v.token_translate += dt; v.token_transcribe += dt; v.token_solm += dt; v.token_prev += dt;
        v.token_nosp += dt; v.token_not += dt; v.token_beg += dt;
    }
    if (nv < hp.n_vocab) {
        v.id_to_token.resize(hp.n_vocab);
        for (int i = nv; i < hp.n_vocab; i++) {
            std::string w;
            if (i > v.token_beg) w = "[_TT_" + std::to_string(i - v.token_beg) + "]";
            else if (i == v.token_eot) w = "[_EOT_]";
            else if (i == v.token_sot) w = "[_SOT_]";
            else if (i == v.token_translate) w = "[_TRANSLATE_]";
            else if (i == v.token_transcribe) w = "[_TRANSCRIBE_]";
            else if (i == v.token_solm) w = "[_SOLM_]";
            else if (i == v.token_prev) w = "[_PREV_]";
            else if (i == v.token_nosp) w = "[_NOSP_]";
            else if (i == v.token_not) w = "[_NOT_]";
            else if (i == v.token_beg) w = "[_BEG_]";
            else if (i > v.token_sot && i <= v.token_sot + v.num_languages() && i - v.token_sot - 1 < kNLang)
                w = std::string("[_LANG_") + kLang[i - v.token_sot - 1] + "]";
            else w = "[_extra_token_" + std::to_string(i) + "]";
            v.id_to_token[i] = w;
            v.token_to_id[w] = i;
        }
    }
    if (vocab_only) return;
    while (true) {
        int32_t nd = 0, nl = 0, tt = 0;
        if (!F.rd(&nd, 4)) break;  // clean EOF
        if (!F.rd(&nl, 4) || !F.rd(&tt, 4) || nd < 1 || nd > 4 || nl <= 0 || nl > 256) throw Error(-2, "model: bad tensor record");
        HostTensor T;
        T.ne.resize(nd); T.ttype = tt;
        size_t n = 1;
        for (int i = 0; i < nd; i++) {
            int32_t e = 0;
            if (!F.rd(&e, 4) || e <= 0) throw Error(-2, "model: bad tensor dims");
            T.ne[i] = e; n *= (size_t)e;
        }
        std::string name(nl, '\0');
        if (!F.rd(&name[0], nl)) throw Error(-2, "model: truncated tensor name");
        T.f32.resize(n);
        if (tt == 0) {
            if (!F.rd(T.f32.data(), n * 4)) throw Error(-2, "model: truncated tensor " + name);
        } else if (tt == 1) {
            std::vector<uint16_t> h(n);
            if (!F.rd(h.data(), n * 2)) throw Error(-2, "model: truncated tensor " + name);
            for (size_t i = 0; i < n; i++) T.f32[i] = half_to_float(h[i]);
        } else if (q_block_bytes(tt)) {
            // dequantised here, converted to the engine's operand type (f16 / bf16) at upload: the decoder streams 2-byte weights
            const size_t bb = q_block_bytes(tt);
            if (T.ne[0] % 32) throw Error(-2, "model: quantised tensor " + name + " has a row length that is not a multiple of 32");
            std::vector<uint8_t> raw(n / 32 * bb);
            if (!F.rd(raw.data(), raw.size())) throw Error(-2, "model: truncated tensor " + name);
            for (size_t b = 0; b < n / 32; b++) dequant_block(tt, raw.data() + b * bb, T.f32.data() + b * 32, half_to_float);
            m.n_quantised++;
        } else {
            throw Error(-2, "model: unsupported tensor type " + std::to_string(tt) + " for " + name);
        }
        m.t[name] = std::move(T);
    }
}

}  // namespace ss
