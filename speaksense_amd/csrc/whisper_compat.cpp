// whisper.h-compatible shim over the ss_* API (include/whisper_compat.h).  Each whisper_full* call is one chunk through the batch
// former, so states running concurrently on one context share device batches -- which the reference's per-stream
// states (/root/reference/src/grpc/handlers/asr.rs:164) and per-task states (schedule/processors/transcribe.rs:100) do.
// The whole function list of whisper.h v1.5.4 is exported; what this engine cannot honour returns an error, never aborts.
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#include <vector>

#include "../../include/speaksense.h"
#include "../../include/whisper_compat.h"
#include "common.h"

struct whisper_context { ss_engine* eng; whisper_state* default_state; };
struct whisper_state {
    ss_session* ses; whisper_context* ctx;
    std::vector<float> pcm;      // samples last given to whisper_pcm_to_mel* (language detection runs on them)
    std::vector<float> mel;      // [n_mel][n_len] from the device log-mel
    int n_len = 0;
    std::string token_text;      // whisper_full_get_token_text's return buffer
    std::vector<float> enc;      // whisper_encode*: encoder output of the window [n_audio_ctx][n_audio_state] (host copy; the cross-K/V lives in the session)
    std::vector<float> logits;   // whisper_decode*: raw logits of the last token
    bool encoded = false;
};

static ggml_log_callback g_log_cb = nullptr;
static void* g_log_ud = nullptr;
static void wlog(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (g_log_cb) g_log_cb(2 /* GGML_LOG_LEVEL_ERROR */, buf, g_log_ud);
    else fputs(buf, stderr);
}
static int unsupported(const char* fn) { wlog("%s: not supported by the MI355X engine (use whisper_full*)\n", fn); return -1; }

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
static int hp(struct whisper_context* ctx, int idx) {
    int32_t h[11] = {0};
    if (!ctx || ss_engine_hparams(ctx->eng, h) != SS_OK) return 0;
    return h[idx];
}
static int sp(struct whisper_context* ctx, int idx) {
    int32_t t[9] = {0};
    if (!ctx || ss_engine_special_tokens(ctx->eng, t) != SS_OK) return 0;
    return t[idx];
}
static whisper_state* dstate(struct whisper_context* ctx) {
    if (!ctx) return nullptr;
    if (!ctx->default_state) ctx->default_state = whisper_init_state(ctx);
    return ctx->default_state;
}

extern "C" {

// ---- context / state lifetime -------------------------------------------------------------------------------------------------------
struct whisper_context_params whisper_context_default_params(void) {
    whisper_context_params p;
    memset(&p, 0, sizeof(p));
    p.use_gpu = true;
#ifdef SS_WHISPER_H_POST_1_5_4
    p.gpu_device = 0;
    p.dtw_token_timestamps = false;
    p.dtw_aheads_preset = WHISPER_AHEADS_NONE;
    p.dtw_n_top = -1;
    p.dtw_aheads.n_heads = 0; p.dtw_aheads.heads = nullptr;
    p.dtw_mem_size = 1024 * 1024 * 128;
#endif
    return p;
}
struct whisper_context_params* whisper_context_default_params_by_ref(void) {
    whisper_context_params* p = (whisper_context_params*)malloc(sizeof(whisper_context_params));
    *p = whisper_context_default_params();
    return p;
}
void whisper_free_context_params(struct whisper_context_params* params) { free(params); }
void whisper_free_params(struct whisper_full_params* params) { free(params); }

struct whisper_context* whisper_init_from_file_with_params_no_state(const char* path_model, struct whisper_context_params cparams) {
    if (!path_model) return nullptr;
    ss_engine_opts o;
    memset(&o, 0, sizeof(o));
    if (!cparams.use_gpu) wlog("whisper_init: use_gpu = false ignored -- the MI355X engine has no CPU path\n");
#ifdef SS_WHISPER_H_POST_1_5_4
    o.device = env_int("SS_DEVICE", cparams.gpu_device);
    if (cparams.dtw_token_timestamps) wlog("whisper_init: dtw_token_timestamps is not implemented -- t_dtw stays -1 (token_timestamps t0 / t1 are computed)\n");
#else
    o.device = env_int("SS_DEVICE", 0);
#endif
    const char* dt = getenv("SS_DTYPE");
    o.dtype = (dt && !strcmp(dt, "bf16")) ? SS_DTYPE_BF16 : (dt && !strcmp(dt, "fp8")) ? SS_DTYPE_FP8 : SS_DTYPE_F16;
    o.max_batch = env_int("SS_MAX_BATCH", 8);
    o.max_decoders = 5;
    o.batch_wait_us = env_int("SS_BATCH_WAIT_US", 2000);
    ss_engine* e = nullptr;
    if (ss_engine_create(path_model, &o, &e) != SS_OK) {
        wlog("whisper_init_from_file_with_params_no_state: %s\n", ss_last_error());
        return nullptr;
    }
    return new whisper_context{e, nullptr};
}
struct whisper_context* whisper_init_from_file_with_params(const char* path_model, struct whisper_context_params params) {
    whisper_context* ctx = whisper_init_from_file_with_params_no_state(path_model, params);
    if (ctx) ctx->default_state = whisper_init_state(ctx);
    return ctx;
}
// the model loader of this library reads a file: a buffer / custom loader is spooled to a temporary file first
struct whisper_context* whisper_init_from_buffer_with_params_no_state(void* buffer, size_t buffer_size, struct whisper_context_params params) {
    if (!buffer || buffer_size == 0) return nullptr;
    char tmpl[] = "/tmp/ss_model_XXXXXX";
    const int fd = mkstemp(tmpl);
    if (fd < 0) { wlog("whisper_init_from_buffer: cannot create a temporary file\n"); return nullptr; }
    size_t off = 0;
    while (off < buffer_size) {
        const ssize_t w = write(fd, (const char*)buffer + off, buffer_size - off);
        if (w <= 0) break;
        off += (size_t)w;
    }
    close(fd);
    whisper_context* ctx = off == buffer_size ? whisper_init_from_file_with_params_no_state(tmpl, params) : nullptr;
    unlink(tmpl);
    return ctx;
}
struct whisper_context* whisper_init_from_buffer_with_params(void* buffer, size_t buffer_size, struct whisper_context_params params) {
    whisper_context* ctx = whisper_init_from_buffer_with_params_no_state(buffer, buffer_size, params);
    if (ctx) ctx->default_state = whisper_init_state(ctx);
    return ctx;
}
struct whisper_context* whisper_init_with_params_no_state(struct whisper_model_loader* loader, struct whisper_context_params params) {
    if (!loader || !loader->read) return nullptr;
    std::vector<char> buf;
    char chunk[1 << 16];
    while (true) {
        if (loader->eof && loader->eof(loader->context)) break;
        const size_t n = loader->read(loader->context, chunk, sizeof(chunk));
        if (n == 0) break;
        buf.insert(buf.end(), chunk, chunk + n);
    }
    if (loader->close) loader->close(loader->context);
    return whisper_init_from_buffer_with_params_no_state(buf.data(), buf.size(), params);
}
struct whisper_context* whisper_init_with_params(struct whisper_model_loader* loader, struct whisper_context_params params) {
    whisper_context* ctx = whisper_init_with_params_no_state(loader, params);
    if (ctx) ctx->default_state = whisper_init_state(ctx);
    return ctx;
}
struct whisper_context* whisper_init_from_file(const char* path_model) { return whisper_init_from_file_with_params(path_model, whisper_context_default_params()); }
struct whisper_context* whisper_init_from_buffer(void* b, size_t n) { return whisper_init_from_buffer_with_params(b, n, whisper_context_default_params()); }
struct whisper_context* whisper_init(struct whisper_model_loader* l) { return whisper_init_with_params(l, whisper_context_default_params()); }
struct whisper_context* whisper_init_from_file_no_state(const char* path_model) { return whisper_init_from_file_with_params_no_state(path_model, whisper_context_default_params()); }
struct whisper_context* whisper_init_from_buffer_no_state(void* b, size_t n) { return whisper_init_from_buffer_with_params_no_state(b, n, whisper_context_default_params()); }
struct whisper_context* whisper_init_no_state(struct whisper_model_loader* l) { return whisper_init_with_params_no_state(l, whisper_context_default_params()); }

struct whisper_state* whisper_init_state(struct whisper_context* ctx) {
    if (!ctx) return nullptr;
    ss_session* s = ss_session_create(ctx->eng);
    if (!s) return nullptr;
    whisper_state* st = new whisper_state();
    st->ses = s; st->ctx = ctx;
    return st;
}
int whisper_ctx_init_openvino_encoder(struct whisper_context*, const char*, const char*, const char*) { return 1; }
void whisper_free_state(struct whisper_state* state) {
    if (!state) return;
    ss_session_free(state->ses);
    delete state;
}
void whisper_free(struct whisper_context* ctx) {
    if (!ctx) return;
    if (ctx->default_state) whisper_free_state(ctx->default_state);
    ss_engine_free(ctx->eng);
    delete ctx;
}

// ---- low-level API ------------------------------------------------------------------------------------------------------------------
int whisper_pcm_to_mel_with_state(struct whisper_context* ctx, struct whisper_state* state, const float* samples, int n_samples, int) {
    if (!ctx || !state || !samples || n_samples <= 0) return -1;
    const int n_len = ss_mel_n_len(n_samples);
    state->mel.assign((size_t)hp(ctx, 9) * n_len, 0.0f);
    if (ss_log_mel(ctx->eng, samples, n_samples, state->mel.data(), n_len) != SS_OK) { wlog("whisper_pcm_to_mel: %s\n", ss_last_error()); return -1; }
    state->n_len = n_len;
    state->pcm.assign(samples, samples + n_samples);
    return 0;
}
int whisper_pcm_to_mel(struct whisper_context* ctx, const float* samples, int n_samples, int n_threads) {
    return whisper_pcm_to_mel_with_state(ctx, dstate(ctx), samples, n_samples, n_threads);
}
int whisper_pcm_to_mel_phase_vocoder(struct whisper_context*, const float*, int, int) { return unsupported("whisper_pcm_to_mel_phase_vocoder"); }
int whisper_pcm_to_mel_phase_vocoder_with_state(struct whisper_context*, struct whisper_state*, const float*, int, int) {
    return unsupported("whisper_pcm_to_mel_phase_vocoder_with_state");
}
int whisper_set_mel_with_state(struct whisper_context* ctx, struct whisper_state* state, const float* data, int n_len, int n_mel) {
    if (!ctx || !state || !data || n_len <= 0) return -1;
    if (n_mel != hp(ctx, 9)) { wlog("whisper_set_mel: invalid number of mel bands: %d (expected %d)\n", n_mel, hp(ctx, 9)); return -1; }
    state->mel.assign(data, data + (size_t)n_len * n_mel);
    state->n_len = n_len;
    state->pcm.clear();
    return 0;
}
int whisper_set_mel(struct whisper_context* ctx, const float* data, int n_len, int n_mel) { return whisper_set_mel_with_state(ctx, dstate(ctx), data, n_len, n_mel); }
int whisper_encode_with_state(struct whisper_context* ctx, struct whisper_state* state, int offset, int) {
    if (!ctx || !state) return -1;
    if (state->mel.empty() || state->n_len <= 0) { wlog("whisper_encode: no spectrogram (call whisper_pcm_to_mel or whisper_set_mel first)\n"); return -1; }
    if (offset < 0) { wlog("whisper_encode: negative offset %d\n", offset); return -1; }     // offset >= n_len: a window of zeros, as whisper.cpp pads it (mel_window_kernel zero-fills)
    state->enc.resize((size_t)hp(ctx, 1) * hp(ctx, 2));
    state->encoded = false;
    if (ss_encode(ctx->eng, state->mel.data(), state->n_len, offset, state->enc.data()) != SS_OK ||
        ss_session_set_encoder(state->ses, state->enc.data()) != SS_OK) { wlog("whisper_encode: %s\n", ss_last_error()); return -1; }
    state->encoded = true;
    return 0;
}
int whisper_encode(struct whisper_context* ctx, int offset, int n_threads) { return whisper_encode_with_state(ctx, dstate(ctx), offset, n_threads); }
int whisper_decode_with_state(struct whisper_context* ctx, struct whisper_state* state, const whisper_token* tokens, int n_tokens, int n_past, int) {
    if (!ctx || !state || !tokens || n_tokens <= 0) return -1;
    if (!state->encoded) { wlog("whisper_decode: no encoder output (call whisper_encode first)\n"); return -1; }
    state->logits.resize((size_t)hp(ctx, 0));
    if (ss_session_decode(state->ses, tokens, n_tokens, n_past, state->logits.data()) != SS_OK) {
        wlog("whisper_decode: %s\n", ss_last_error());
        state->logits.clear();
        return -1;
    }
    return 0;
}
int whisper_decode(struct whisper_context* ctx, const whisper_token* tokens, int n_tokens, int n_past, int n_threads) {
    return whisper_decode_with_state(ctx, dstate(ctx), tokens, n_tokens, n_past, n_threads);
}
float* whisper_get_logits_from_state(struct whisper_state* state) { return state && !state->logits.empty() ? state->logits.data() : nullptr; }
float* whisper_get_logits(struct whisper_context* ctx) { return ctx && ctx->default_state ? whisper_get_logits_from_state(ctx->default_state) : nullptr; }

int whisper_tokenize(struct whisper_context* ctx, const char* text, whisper_token* tokens, int n_max_tokens) {
    if (!ctx || !text) return -1;
    if (!tokens && n_max_tokens == 0) {   // whisper.h has no size query: n_max_tokens < needed returns -(needed) also here (whisper_token_count negates it)
        const int need = ss_engine_tokenize(ctx->eng, text, nullptr, 0);
        return need > 0 ? -need : need;
    }
    const int n = ss_engine_tokenize(ctx->eng, text, tokens, n_max_tokens);
    if (n == SS_ERR_BUFFER) {   // whisper.h's convention: -(tokens needed)
        const int need = ss_engine_tokenize(ctx->eng, text, nullptr, 0);
        wlog("whisper_tokenize: too many resulting tokens: %d (max %d)\n", need, n_max_tokens);
        return -need;
    }
    return n;
}
int whisper_lang_max_id(void) { return 99; }
int whisper_lang_id(const char* lang) {
    if (!lang) return -1;
    const int id = ss::lang_id(lang);
    if (id >= 0) return id;
    static const char* const full[] = {"english","chinese","german","spanish","russian","korean","french","japanese","portuguese","turkish","polish","catalan","dutch",
        "arabic","swedish","italian","indonesian","hindi","finnish","vietnamese","hebrew","ukrainian","greek","malay","czech","romanian","danish","hungarian","tamil",
        "norwegian","thai","urdu","croatian","bulgarian","lithuanian","latin","maori","malayalam","welsh","slovak","telugu","persian","latvian","bengali","serbian",
        "azerbaijani","slovenian","kannada","estonian","macedonian","breton","basque","icelandic","armenian","nepali","mongolian","bosnian","kazakh","albanian",
        "swahili","galician","marathi","punjabi","sinhala","khmer","shona","yoruba","somali","afrikaans","occitan","georgian","belarusian","tajik","sindhi","gujarati",
        "amharic","yiddish","lao","uzbek","faroese","haitian creole","pashto","turkmen","nynorsk","maltese","sanskrit","luxembourgish","myanmar","tibetan","tagalog",
        "malagasy","assamese","tatar","hawaiian","lingala","hausa","bashkir","javanese","sundanese","cantonese"};
    for (int i = 0; i < 100; i++) if (!strcmp(lang, full[i])) return i;     // whisper_lang_id also accepts the full names
    wlog("whisper_lang_id: unknown language '%s'\n", lang);
    return -1;
}
const char* whisper_lang_str(int id) {
    const char* c = ss::lang_code(id);
    if (!c) wlog("whisper_lang_str: unknown language id %d\n", id);
    return c;
}
const char* whisper_lang_str_full(int id) {
    static const char* const full[] = {"english","chinese","german","spanish","russian","korean","french","japanese","portuguese","turkish","polish","catalan","dutch",
        "arabic","swedish","italian","indonesian","hindi","finnish","vietnamese","hebrew","ukrainian","greek","malay","czech","romanian","danish","hungarian","tamil",
        "norwegian","thai","urdu","croatian","bulgarian","lithuanian","latin","maori","malayalam","welsh","slovak","telugu","persian","latvian","bengali","serbian",
        "azerbaijani","slovenian","kannada","estonian","macedonian","breton","basque","icelandic","armenian","nepali","mongolian","bosnian","kazakh","albanian",
        "swahili","galician","marathi","punjabi","sinhala","khmer","shona","yoruba","somali","afrikaans","occitan","georgian","belarusian","tajik","sindhi","gujarati",
        "amharic","yiddish","lao","uzbek","faroese","haitian creole","pashto","turkmen","nynorsk","maltese","sanskrit","luxembourgish","myanmar","tibetan","tagalog",
        "malagasy","assamese","tatar","hawaiian","lingala","hausa","bashkir","javanese","sundanese","cantonese"};
    if (id < 0 || id >= 100) { wlog("whisper_lang_str_full: unknown language id %d\n", id); return nullptr; }
    return full[id];
}
int whisper_lang_auto_detect_with_state(struct whisper_context* ctx, struct whisper_state* state, int offset_ms, int, float* lang_probs) {
    if (!ctx || !state) return -1;
    if (offset_ms < 0) { wlog("whisper_lang_auto_detect: offset %dms is before the start of the audio\n", offset_ms); return -1; }
    if (state->pcm.empty()) { wlog("whisper_lang_auto_detect: no samples (call whisper_pcm_to_mel first; a mel set with whisper_set_mel cannot be used)\n"); return -2; }
    if (offset_ms != 0) { wlog("whisper_lang_auto_detect: only offset_ms = 0 is supported\n"); return -2; }
    ss_params p;
    ss_default_params(&p);
    p.detect_language = 1;
    const int rc = ss_transcribe(state->ses, state->pcm.data(), (int)state->pcm.size(), &p);
    if (rc != SS_OK) { wlog("whisper_lang_auto_detect: %s\n", ss_last_error()); return rc == SS_ERR_LANG ? -2 : -6; }
    const int id = ss_result_lang_id(state->ses);
    if (lang_probs) { for (int i = 0; i <= whisper_lang_max_id(); i++) lang_probs[i] = 0.0f; if (id >= 0) lang_probs[id] = 1.0f; }
    return id;
}
int whisper_lang_auto_detect(struct whisper_context* ctx, int offset_ms, int n_threads, float* lang_probs) {
    return whisper_lang_auto_detect_with_state(ctx, dstate(ctx), offset_ms, n_threads, lang_probs);
}

int whisper_n_len_from_state(struct whisper_state* state) { return state ? state->n_len : 0; }
int whisper_n_len(struct whisper_context* ctx) { return ctx && ctx->default_state ? ctx->default_state->n_len : 0; }
int whisper_n_vocab(struct whisper_context* ctx) { return hp(ctx, 0); }
int whisper_n_audio_ctx(struct whisper_context* ctx) { return hp(ctx, 1); }
int whisper_n_text_ctx(struct whisper_context* ctx) { return hp(ctx, 5); }
int whisper_is_multilingual(struct whisper_context* ctx) { return hp(ctx, 0) >= 51865; }
int whisper_model_n_vocab(struct whisper_context* ctx) { return hp(ctx, 0); }
int whisper_model_n_audio_ctx(struct whisper_context* ctx) { return hp(ctx, 1); }
int whisper_model_n_audio_state(struct whisper_context* ctx) { return hp(ctx, 2); }
int whisper_model_n_audio_head(struct whisper_context* ctx) { return hp(ctx, 3); }
int whisper_model_n_audio_layer(struct whisper_context* ctx) { return hp(ctx, 4); }
int whisper_model_n_text_ctx(struct whisper_context* ctx) { return hp(ctx, 5); }
int whisper_model_n_text_state(struct whisper_context* ctx) { return hp(ctx, 6); }
int whisper_model_n_text_head(struct whisper_context* ctx) { return hp(ctx, 7); }
int whisper_model_n_text_layer(struct whisper_context* ctx) { return hp(ctx, 8); }
int whisper_model_n_mels(struct whisper_context* ctx) { return hp(ctx, 9); }
int whisper_model_ftype(struct whisper_context* ctx) { return hp(ctx, 10); }
int whisper_model_type(struct whisper_context* ctx) {   // e_model: by n_audio_layer, as whisper_model_load does
    switch (hp(ctx, 4)) { case 4: return 1; case 6: return 2; case 12: return 3; case 24: return 4; case 32: return 5; default: return 0; }
}
const char* whisper_model_type_readable(struct whisper_context* ctx) {
    static const char* const names[] = {"unknown", "tiny", "base", "small", "medium", "large"};
    return names[whisper_model_type(ctx)];
}

const char* whisper_token_to_str(struct whisper_context* ctx, whisper_token token) { return ctx ? ss_engine_token_str(ctx->eng, token) : nullptr; }
whisper_token whisper_token_eot(struct whisper_context* ctx) { return sp(ctx, 0); }
whisper_token whisper_token_sot(struct whisper_context* ctx) { return sp(ctx, 1); }
whisper_token whisper_token_translate(struct whisper_context* ctx) { return sp(ctx, 2); }
whisper_token whisper_token_transcribe(struct whisper_context* ctx) { return sp(ctx, 3); }
whisper_token whisper_token_solm(struct whisper_context* ctx) { return sp(ctx, 4); }
whisper_token whisper_token_prev(struct whisper_context* ctx) { return sp(ctx, 5); }
whisper_token whisper_token_nosp(struct whisper_context* ctx) { return sp(ctx, 6); }
whisper_token whisper_token_not(struct whisper_context* ctx) { return sp(ctx, 7); }
whisper_token whisper_token_beg(struct whisper_context* ctx) { return sp(ctx, 8); }
whisper_token whisper_token_lang(struct whisper_context* ctx, int lang_id) { return sp(ctx, 1) + 1 + lang_id; }

void whisper_print_timings(struct whisper_context* ctx) {
    if (!ctx) return;
    double ms[4] = {0, 0, 0, 0};
    int64_t cnt[4] = {0, 0, 0, 0};
    int32_t nl = 0;
    if (ss_engine_totals(ctx->eng, ms, cnt, &nl) != SS_OK) return;
    wlog("whisper_print_timings: device time since load over %d lane(s): mel %.2f ms, encode %.2f ms (%lld windows), decode %.2f ms (%lld passes, %lld rows)\n",
         nl, ms[0], ms[1], (long long)cnt[2], ms[2], (long long)cnt[0], (long long)cnt[1]);
}
void whisper_reset_timings(struct whisper_context*) {}   // the totals are cumulative; callers take differences
const char* whisper_print_system_info(void) { return "HIP = 1 | MI355X (gfx950) = 1 | MFMA f16/bf16 = 1 | CPU fallback = 0 | "; }
void whisper_log_set(ggml_log_callback log_callback, void* user_data) { g_log_cb = log_callback; g_log_ud = user_data; }

// ---- whisper_full ---------------------------------------------------------------------------------------------------------------------
struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy) {
    whisper_full_params p;
    memset(&p, 0, sizeof(p));
    p.strategy = strategy;
    p.n_threads = 4;
    p.n_max_text_ctx = 16384;
    p.no_context = true;
    p.print_progress = true;
    p.print_timestamps = true;
    p.thold_pt = 0.01f;
    p.thold_ptsum = 0.01f;
    p.language = "en";
    p.suppress_blank = true;
    p.temperature = 0.0f;
    p.max_initial_ts = 1.0f;
    p.length_penalty = -1.0f;
    p.temperature_inc = 0.2f;
    p.entropy_thold = 2.4f;
    p.logprob_thold = -1.0f;
    p.no_speech_thold = 0.6f;
    p.greedy.best_of = strategy == WHISPER_SAMPLING_GREEDY ? 5 : -1;
    p.beam_search.beam_size = strategy == WHISPER_SAMPLING_BEAM_SEARCH ? 5 : -1;
    p.beam_search.patience = -1.0f;
    p.grammar_penalty = 100.0f;
    return p;
}
struct whisper_full_params* whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy) {
    whisper_full_params* p = (whisper_full_params*)malloc(sizeof(whisper_full_params));
    *p = whisper_full_default_params(strategy);
    return p;
}

// whisper_full_params -> ss_params; returns SS_OK or the refusal
static int map_params(struct whisper_context* ctx, const struct whisper_full_params& params, ss_params& p) {
    // features of whisper_full this path does not implement are refused, never silently ignored
    if (params.strategy != WHISPER_SAMPLING_GREEDY || params.speed_up || params.n_grammar_rules > 0 || params.logits_filter_callback)
        return SS_ERR_UNSUPPORTED;
    // Honoured beyond what the reference sets: suppress_non_speech_tokens (whisper.rs:156 false), max_len + split_on_word (whisper.rs:167,161:
    // whisper_wrap_segment; like whisper.cpp only together with token_timestamps), token_timestamps (whisper.rs:160:
    // whisper_full_get_token_data(..).t0 / t1 / vlen).  The four plain callbacks are honoured at CHUNK granularity (whisper_full_with_state
    // below): the windows of a chunk complete inside a device batch shared with other states, so nothing can fire from inside it.
    ss_default_params(&p);
    p.best_of = params.greedy.best_of > 0 ? params.greedy.best_of : 1;
    p.temperature = params.temperature; p.temperature_inc = params.temperature_inc; p.entropy_thold = params.entropy_thold;
    p.logprob_thold = params.logprob_thold; p.max_initial_ts = params.max_initial_ts; p.length_penalty = params.length_penalty;
    p.no_context = params.no_context; p.single_segment = params.single_segment; p.no_timestamps = params.no_timestamps;
    p.suppress_blank = params.suppress_blank; p.tdrz_enable = params.tdrz_enable; p.print_special = params.print_special;
    p.max_tokens = params.max_tokens; p.audio_ctx = params.audio_ctx; p.translate = params.translate;
    p.token_timestamps = params.token_timestamps; p.thold_pt = params.thold_pt; p.thold_ptsum = params.thold_ptsum;
    p.suppress_non_speech_tokens = params.suppress_non_speech_tokens; p.max_len = params.max_len; p.split_on_word = params.split_on_word;
    if (params.language) { strncpy(p.language, params.language, sizeof(p.language) - 1); p.language[sizeof(p.language) - 1] = 0; }
    else p.language[0] = 0;   // nullptr / "" / "auto": detect
    p.n_max_text_ctx = params.n_max_text_ctx; p.offset_ms = params.offset_ms; p.duration_ms = params.duration_ms;
    p.detect_language = params.detect_language;
    p.prompt_tokens = params.prompt_tokens; p.prompt_n_tokens = params.prompt_n_tokens; p.initial_prompt = params.initial_prompt;
    return SS_OK;
}

// whisper.cpp's to_timestamp(t, comma = false): t in centiseconds -> "HH:MM:SS.mmm"
static void wcpp_timestamp(int64_t t, char* buf, size_t n) {
    int64_t msec = t * 10;
    const int64_t hr = msec / (1000 * 60 * 60); msec -= hr * (1000 * 60 * 60);
    const int64_t mn = msec / (1000 * 60); msec -= mn * (1000 * 60);
    const int64_t sec = msec / 1000; msec -= sec * 1000;
    snprintf(buf, n, "%02d:%02d:%02d.%03d", (int)hr, (int)mn, (int)sec, (int)msec);
}
int whisper_full_with_state(struct whisper_context* ctx, struct whisper_state* state, struct whisper_full_params params, const float* samples,
                            int n_samples) {
    if (!ctx || !state) return -1;
    ss_params p;
    int rc = map_params(ctx, params, p);
    if (rc != SS_OK) return rc;
    state->encoded = false;      // the chunk's windows take over the session's cross-K/V and self-KV: a later whisper_decode needs a new whisper_encode
    state->logits.clear();
    // Callbacks, at chunk granularity (whisper.cpp fires them per window from inside whisper_full; here a chunk's windows run inside a device batch):
    //   abort_callback          polled once before the chunk is submitted: true -> the call returns -6 like a whisper_encode that was aborted
    //   encoder_begin_callback  called once before the chunk is submitted: false -> -6 ("encoder_begin_callback returned false - aborting")
    //   progress_callback       called with 100 when the chunk has completed (whisper.cpp: whole percentages at window starts)
    //   new_segment_callback    called ONCE when the chunk has completed, n_new = all its segments (whisper.cpp: once per window with that window's)
    if (params.abort_callback && params.abort_callback(params.abort_callback_user_data)) { wlog("whisper_full_with_state: aborted by abort_callback\n"); return -6; }
    if (params.encoder_begin_callback && !params.encoder_begin_callback(ctx, state, params.encoder_begin_callback_user_data)) {
        wlog("whisper_full_with_state: encoder_begin_callback returned false - aborting\n");
        return -6;
    }
    ss_ticket* t = nullptr;
    rc = ss_submit(state->ses, samples, n_samples, &p, &t);
    if (rc != SS_OK) return rc;
    rc = ss_wait(t);
    if (rc == SS_OK) {
        if (params.progress_callback) params.progress_callback(ctx, state, 100, params.progress_callback_user_data);
        const int n_new = ss_result_n_segments(state->ses);
        if (params.new_segment_callback && n_new > 0) params.new_segment_callback(ctx, state, n_new, params.new_segment_callback_user_data);
    }
    // The reference switches whisper.cpp's own segment printing on (/root/reference/src/asr/whisper.rs:145-150: print_realtime, print_timestamps,
    // print_progress all true), so a drop-in that stays silent changes what the service's stdout shows.  whisper_full_with_state prints each
    // segment as its window is finalised -- "[%s --> %s]  %s\n" with to_timestamp() times under print_timestamps, the bare text otherwise; here the
    // windows of a chunk complete inside a device batch, so the same lines appear when the call returns.  (print_progress reports whole
    // percentages at window starts on stderr: a one-window chunk prints nothing there, and nothing is printed here.)
    if (rc == SS_OK && params.print_realtime) {
        const int n = ss_result_n_segments(state->ses);
        for (int i = 0; i < n; i++) {
            const char* text = ss_result_segment_text(state->ses, i);
            if (params.print_timestamps) {
                char a[32], b[32];
                wcpp_timestamp(ss_result_segment_t0(state->ses, i), a, sizeof(a));
                wcpp_timestamp(ss_result_segment_t1(state->ses, i), b, sizeof(b));
                printf("[%s --> %s]  %s\n", a, b, text ? text : "");
            } else {
                printf("%s", text ? text : "");
            }
        }
        fflush(stdout);
    }
    return rc;
}
int whisper_full(struct whisper_context* ctx, struct whisper_full_params params, const float* samples, int n_samples) {
    if (!ctx) return -1;
    return whisper_full_with_state(ctx, dstate(ctx), params, samples, n_samples);
}
int whisper_full_parallel(struct whisper_context* ctx, struct whisper_full_params params, const float* samples, int n_samples, int n_processors) {
    if (!ctx) return -1;
    if (n_processors <= 1) return whisper_full(ctx, params, samples, n_samples);
    // whisper.cpp: chunk 0 = [0, offset + per) on the context's own state with the caller's offset, chunks 1.. = per samples each on fresh
    // states with offset 0, the last one taking the remainder; results appended with their time offset, start times clamped to the previous end
    const int offset_samples = (16000 * params.offset_ms) / 1000;
    const int per = (n_samples - offset_samples) / n_processors;
    if (per <= 0) return whisper_full(ctx, params, samples, n_samples);
    ss_params p0;
    int rc = map_params(ctx, params, p0);
    if (rc != SS_OK) return rc;
    ss_params pi = p0;
    pi.offset_ms = 0;
    whisper_state* st0 = dstate(ctx);
    std::vector<ss_session*> extra;
    std::vector<ss_ticket*> tickets(n_processors, nullptr);
    rc = ss_submit(st0->ses, samples, offset_samples + per, &p0, &tickets[0]);
    for (int i = 0; rc == SS_OK && i < n_processors - 1; i++) {
        const int start = offset_samples + (i + 1) * per;
        const int n_cur = (i + 1 == n_processors - 1) ? n_samples - start : per;
        ss_session* s = ss_session_create(ctx->eng);
        extra.push_back(s);
        rc = s ? ss_submit(s, samples + start, n_cur, &pi, &tickets[i + 1]) : SS_ERR_ARG;
    }
    int ret = rc;
    for (int i = 0; i < n_processors; i++) if (tickets[i]) { const int r = ss_wait(tickets[i]); if (r != SS_OK && ret == SS_OK) ret = r; }
    if (ret == SS_OK) {
        const int64_t offset_t = (int64_t)params.offset_ms / 10;
        for (size_t i = 0; i < extra.size(); i++)
            ss_result_append(st0->ses, extra[i], 100 * ((int64_t)(i + 1) * per) / 16000 + offset_t);
    }
    for (ss_session* s : extra) ss_session_free(s);
    if (ret == SS_OK) {   // callbacks as in whisper_full_with_state: once, on the merged result
        if (params.progress_callback) params.progress_callback(ctx, st0, 100, params.progress_callback_user_data);
        const int n_new = ss_result_n_segments(st0->ses);
        if (params.new_segment_callback && n_new > 0) params.new_segment_callback(ctx, st0, n_new, params.new_segment_callback_user_data);
    }
    return ret;
}

int whisper_full_n_segments_from_state(struct whisper_state* state) { return state ? ss_result_n_segments(state->ses) : 0; }
const char* whisper_full_get_segment_text_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_text(state->ses, i) : nullptr; }
int64_t whisper_full_get_segment_t0_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_t0(state->ses, i) : 0; }
int64_t whisper_full_get_segment_t1_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_t1(state->ses, i) : 0; }
bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state* state, int i) {
    return state ? ss_result_segment_speaker_turn_next(state->ses, i) != 0 : false;
}
int whisper_full_lang_id_from_state(struct whisper_state* state) { return state ? ss_result_lang_id(state->ses) : -1; }
int whisper_full_n_tokens_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_n_tokens(state->ses, i) : 0; }
whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state* state, int i, int k) {
    whisper_token_data d;
    memset(&d, 0, sizeof(d));
    d.t0 = d.t1 = -1;
#ifdef SS_WHISPER_H_POST_1_5_4
    d.t_dtw = -1;
#endif
    float f[4] = {0, 0, 0, 0};
    if (state && ss_result_segment_token(state->ses, i, k, &d.id, &d.tid, f) == SS_OK) {
        d.p = f[0]; d.plog = f[1]; d.pt = f[2]; d.ptsum = f[3];
        (void)ss_result_segment_token_times(state->ses, i, k, &d.t0, &d.t1, &d.vlen);
    }
    return d;
}
whisper_token whisper_full_get_token_id_from_state(struct whisper_state* state, int i, int k) { return whisper_full_get_token_data_from_state(state, i, k).id; }
float whisper_full_get_token_p_from_state(struct whisper_state* state, int i, int k) { return whisper_full_get_token_data_from_state(state, i, k).p; }
const char* whisper_full_get_token_text_from_state(struct whisper_context* ctx, struct whisper_state* state, int i, int k) {
    if (!ctx || !state) return nullptr;
    return ss_engine_token_str(ctx->eng, whisper_full_get_token_id_from_state(state, i, k));
}

int whisper_full_n_segments(struct whisper_context* ctx) { return ctx && ctx->default_state ? whisper_full_n_segments_from_state(ctx->default_state) : 0; }
int whisper_full_lang_id(struct whisper_context* ctx) { return ctx && ctx->default_state ? whisper_full_lang_id_from_state(ctx->default_state) : -1; }
const char* whisper_full_get_segment_text(struct whisper_context* ctx, int i) {
    return ctx && ctx->default_state ? whisper_full_get_segment_text_from_state(ctx->default_state, i) : nullptr;
}
int64_t whisper_full_get_segment_t0(struct whisper_context* ctx, int i) { return ctx && ctx->default_state ? whisper_full_get_segment_t0_from_state(ctx->default_state, i) : 0; }
int64_t whisper_full_get_segment_t1(struct whisper_context* ctx, int i) { return ctx && ctx->default_state ? whisper_full_get_segment_t1_from_state(ctx->default_state, i) : 0; }
bool whisper_full_get_segment_speaker_turn_next(struct whisper_context* ctx, int i) {
    return ctx && ctx->default_state ? whisper_full_get_segment_speaker_turn_next_from_state(ctx->default_state, i) : false;
}
int whisper_full_n_tokens(struct whisper_context* ctx, int i) { return ctx && ctx->default_state ? whisper_full_n_tokens_from_state(ctx->default_state, i) : 0; }
const char* whisper_full_get_token_text(struct whisper_context* ctx, int i, int k) {
    return ctx && ctx->default_state ? whisper_full_get_token_text_from_state(ctx, ctx->default_state, i, k) : nullptr;
}
whisper_token whisper_full_get_token_id(struct whisper_context* ctx, int i, int k) {
    return ctx && ctx->default_state ? whisper_full_get_token_id_from_state(ctx->default_state, i, k) : 0;
}
whisper_token_data whisper_full_get_token_data(struct whisper_context* ctx, int i, int k) {
    return whisper_full_get_token_data_from_state(ctx ? ctx->default_state : nullptr, i, k);
}
float whisper_full_get_token_p(struct whisper_context* ctx, int i, int k) { return ctx && ctx->default_state ? whisper_full_get_token_p_from_state(ctx->default_state, i, k) : 0.0f; }

int whisper_bench_memcpy(int) { return unsupported("whisper_bench_memcpy"); }
const char* whisper_bench_memcpy_str(int) { return "whisper_bench_memcpy: ggml CPU benchmark, not part of the MI355X engine\n"; }
int whisper_bench_ggml_mul_mat(int) { return unsupported("whisper_bench_ggml_mul_mat"); }
const char* whisper_bench_ggml_mul_mat_str(int) { return "whisper_bench_ggml_mul_mat: ggml CPU benchmark, not part of the MI355X engine\n"; }

}  // extern "C"
