// whisper.h-compatible shim over the ss_* API (include/whisper_compat.h).  Each call is one chunk through the batch
// former, so states running concurrently on one context share device batches -- which the reference's per-stream
// states (/root/reference/src/grpc/handlers/asr.rs:164) and per-task states (schedule/processors/transcribe.rs:100) do.
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/speaksense.h"
#include "../../include/whisper_compat.h"
#include "common.h"

struct whisper_context { ss_engine* eng; whisper_state* default_state; };
struct whisper_state { ss_session* ses; whisper_context* ctx; };

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

extern "C" {

struct whisper_context_params whisper_context_default_params(void) {
    whisper_context_params p;
    p.use_gpu = true;
    return p;
}

struct whisper_context* whisper_init_from_file_with_params_no_state(const char* path_model, struct whisper_context_params) {
    ss_engine_opts o;
    memset(&o, 0, sizeof(o));
    o.device = env_int("SS_DEVICE", 0);
    const char* dt = getenv("SS_DTYPE");
    o.dtype = (dt && !strcmp(dt, "bf16")) ? SS_DTYPE_BF16 : SS_DTYPE_F16;
    o.max_batch = env_int("SS_MAX_BATCH", 8);
    o.max_decoders = 5;
    o.batch_wait_us = env_int("SS_BATCH_WAIT_US", 2000);
    ss_engine* e = nullptr;
    if (ss_engine_create(path_model, &o, &e) != SS_OK) {
        fprintf(stderr, "whisper_init_from_file_with_params_no_state: %s\n", ss_last_error());
        return nullptr;
    }
    return new whisper_context{e, nullptr};
}
struct whisper_context* whisper_init_from_file_with_params(const char* path_model, struct whisper_context_params params) {
    whisper_context* ctx = whisper_init_from_file_with_params_no_state(path_model, params);
    if (ctx) ctx->default_state = whisper_init_state(ctx);
    return ctx;
}
struct whisper_state* whisper_init_state(struct whisper_context* ctx) {
    if (!ctx) return nullptr;
    ss_session* s = ss_session_create(ctx->eng);
    if (!s) return nullptr;
    return new whisper_state{s, ctx};
}
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

int whisper_full_with_state(struct whisper_context* ctx, struct whisper_state* state, struct whisper_full_params params, const float* samples,
                            int n_samples) {
    if (!ctx || !state) return -1;
    // features of whisper_full this path does not implement are refused, never silently ignored
    if (params.strategy != WHISPER_SAMPLING_GREEDY || params.speed_up || params.suppress_non_speech_tokens || params.n_grammar_rules > 0 ||
        params.logits_filter_callback || params.max_len > 0)
        return SS_ERR_UNSUPPORTED;
    // callbacks would have to fire from inside the device batch.  token_timestamps / split_on_word (the reference sets both, whisper.rs:160-161)
    // only act together with max_len > 0, refused above.
    if (params.new_segment_callback || params.progress_callback || params.encoder_begin_callback || params.abort_callback) return SS_ERR_UNSUPPORTED;
    if (params.audio_ctx != 0 && params.audio_ctx != whisper_n_audio_ctx(ctx)) return SS_ERR_UNSUPPORTED;
    ss_params p;
    ss_default_params(&p);
    p.best_of = params.greedy.best_of > 0 ? params.greedy.best_of : 1;
    p.temperature = params.temperature; p.temperature_inc = params.temperature_inc; p.entropy_thold = params.entropy_thold;
    p.logprob_thold = params.logprob_thold; p.max_initial_ts = params.max_initial_ts; p.length_penalty = params.length_penalty;
    p.no_context = params.no_context; p.single_segment = params.single_segment; p.no_timestamps = params.no_timestamps;
    p.suppress_blank = params.suppress_blank; p.tdrz_enable = params.tdrz_enable; p.print_special = params.print_special;
    p.max_tokens = params.max_tokens; p.audio_ctx = params.audio_ctx; p.translate = params.translate;
    if (params.language) { strncpy(p.language, params.language, sizeof(p.language) - 1); p.language[sizeof(p.language) - 1] = 0; }
    else p.language[0] = 0;   // nullptr / "" / "auto": detect
    p.n_max_text_ctx = params.n_max_text_ctx; p.offset_ms = params.offset_ms; p.duration_ms = params.duration_ms;
    p.detect_language = params.detect_language;
    p.prompt_tokens = params.prompt_tokens; p.prompt_n_tokens = params.prompt_n_tokens; p.initial_prompt = params.initial_prompt;
    ss_ticket* t = nullptr;
    int rc = ss_submit(state->ses, samples, n_samples, &p, &t);
    if (rc != SS_OK) return rc;
    return ss_wait(t);
}
int whisper_full(struct whisper_context* ctx, struct whisper_full_params params, const float* samples, int n_samples) {
    if (!ctx) return -1;
    if (!ctx->default_state) ctx->default_state = whisper_init_state(ctx);
    return whisper_full_with_state(ctx, ctx->default_state, params, samples, n_samples);
}

int whisper_full_n_segments_from_state(struct whisper_state* state) { return state ? ss_result_n_segments(state->ses) : 0; }
const char* whisper_full_get_segment_text_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_text(state->ses, i) : nullptr; }
int64_t whisper_full_get_segment_t0_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_t0(state->ses, i) : 0; }
int64_t whisper_full_get_segment_t1_from_state(struct whisper_state* state, int i) { return state ? ss_result_segment_t1(state->ses, i) : 0; }
bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state* state, int i) {
    return state ? ss_result_segment_speaker_turn_next(state->ses, i) != 0 : false;
}
int whisper_full_n_segments(struct whisper_context* ctx) { return ctx && ctx->default_state ? whisper_full_n_segments_from_state(ctx->default_state) : 0; }
const char* whisper_full_get_segment_text(struct whisper_context* ctx, int i) {
    return ctx && ctx->default_state ? whisper_full_get_segment_text_from_state(ctx->default_state, i) : nullptr;
}
int64_t whisper_full_get_segment_t0(struct whisper_context* ctx, int i) { return ctx && ctx->default_state ? whisper_full_get_segment_t0_from_state(ctx->default_state, i) : 0; }
int64_t whisper_full_get_segment_t1(struct whisper_context* ctx, int i) { return ctx && ctx->default_state ? whisper_full_get_segment_t1_from_state(ctx->default_state, i) : 0; }

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
int whisper_n_vocab(struct whisper_context* ctx) { return hp(ctx, 0); }
int whisper_n_audio_ctx(struct whisper_context* ctx) { return hp(ctx, 1); }
int whisper_n_text_ctx(struct whisper_context* ctx) { return hp(ctx, 5); }
int whisper_is_multilingual(struct whisper_context* ctx) { return hp(ctx, 0) >= 51865; }
whisper_token whisper_token_eot(struct whisper_context* ctx) { return sp(ctx, 0); }
whisper_token whisper_token_sot(struct whisper_context* ctx) { return sp(ctx, 1); }
whisper_token whisper_token_beg(struct whisper_context* ctx) { return sp(ctx, 8); }
const char* whisper_token_to_str(struct whisper_context* ctx, whisper_token token) { return ctx ? ss_engine_token_str(ctx->eng, token) : nullptr; }
int whisper_lang_id(const char* lang) { return lang ? ss::lang_id(lang) : -1; }

}  // extern "C"
