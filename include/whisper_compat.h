/* whisper_compat.h -- the subset of whisper.cpp's C API (whisper.h, v1.5.x) that the reference reaches through
 * whisper-rs 0.11.1 / whisper-rs-sys 0.9.0, exported by libspeaksense_hip.so with whisper.h-identical names and
 * struct layouts, so the -sys crate can link this library in place of libwhisper (SURVEY.md §8b, C-level boundary).
 *
 * Reference call sites replaced (all in /root/reference/src/asr/whisper.rs):
 *   :23  WhisperContext::new_with_params      -> whisper_context_default_params, whisper_init_from_file_with_params_no_state
 *   :31  ctx.create_state()                   -> whisper_init_state
 *   :131-172 FullParams::new + setters        -> whisper_full_default_params (struct passed BY VALUE)
 *   :75  state.full(params, &audio)           -> whisper_full_with_state
 *   :77  state.full_n_segments()              -> whisper_full_n_segments_from_state
 *   :85  state.full_get_segment_text(i)       -> whisper_full_get_segment_text_from_state  (valid until the next full)
 *   :92-93 full_get_segment_t0 / _t1          -> whisper_full_get_segment_t0/_t1_from_state (centiseconds)
 *   :95  full_get_segment_speaker_turn_next   -> whisper_full_get_segment_speaker_turn_next_from_state
 *   Drop of WhisperState / WhisperContext     -> whisper_free_state, whisper_free
 *
 * whisper.cpp itself is not in /root/reference nor in the build image: the layouts below are restated from
 * whisper.h v1.5.4 and must be re-verified against the vendored header before linking.
 * Engine options that whisper.h has no field for come from the environment: SS_DEVICE, SS_DTYPE (f16|bf16),
 * SS_MAX_BATCH, SS_BATCH_WAIT_US.  whisper_full_with_state goes through the batch former (ss_submit/ss_wait),
 * so concurrent states (one per gRPC stream / REST task) share device batches.
 */
#ifndef SS_WHISPER_COMPAT_H
#define SS_WHISPER_COMPAT_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

struct whisper_context;
struct whisper_state;
typedef int32_t whisper_token;

struct whisper_context_params { bool use_gpu; };

enum whisper_sampling_strategy { WHISPER_SAMPLING_GREEDY, WHISPER_SAMPLING_BEAM_SEARCH };

typedef void (*whisper_new_segment_callback)(struct whisper_context*, struct whisper_state*, int, void*);
typedef void (*whisper_progress_callback)(struct whisper_context*, struct whisper_state*, int, void*);
typedef bool (*whisper_encoder_begin_callback)(struct whisper_context*, struct whisper_state*, void*);
typedef bool (*whisper_abort_callback)(void*);
typedef void (*whisper_logits_filter_callback)(struct whisper_context*, struct whisper_state*, const void*, int, float*, void*);
struct whisper_grammar_element;

struct whisper_full_params {
    enum whisper_sampling_strategy strategy;
    int n_threads;
    int n_max_text_ctx;
    int offset_ms;
    int duration_ms;
    bool translate;
    bool no_context;
    bool no_timestamps;
    bool single_segment;
    bool print_special;
    bool print_progress;
    bool print_realtime;
    bool print_timestamps;
    bool token_timestamps;
    float thold_pt;
    float thold_ptsum;
    int max_len;
    bool split_on_word;
    int max_tokens;
    bool speed_up;
    bool debug_mode;
    int audio_ctx;
    bool tdrz_enable;
    const char* initial_prompt;
    const whisper_token* prompt_tokens;
    int prompt_n_tokens;
    const char* language;
    bool detect_language;
    bool suppress_blank;
    bool suppress_non_speech_tokens;
    float temperature;
    float max_initial_ts;
    float length_penalty;
    float temperature_inc;
    float entropy_thold;
    float logprob_thold;
    float no_speech_thold;
    struct { int best_of; } greedy;
    struct { int beam_size; float patience; } beam_search;
    whisper_new_segment_callback new_segment_callback;
    void* new_segment_callback_user_data;
    whisper_progress_callback progress_callback;
    void* progress_callback_user_data;
    whisper_encoder_begin_callback encoder_begin_callback;
    void* encoder_begin_callback_user_data;
    whisper_abort_callback abort_callback;
    void* abort_callback_user_data;
    whisper_logits_filter_callback logits_filter_callback;
    void* logits_filter_callback_user_data;
    const struct whisper_grammar_element** grammar_rules;
    size_t n_grammar_rules;
    size_t i_start_rule;
    float grammar_penalty;
};

struct whisper_context_params whisper_context_default_params(void);
struct whisper_context* whisper_init_from_file_with_params_no_state(const char* path_model, struct whisper_context_params params);
struct whisper_context* whisper_init_from_file_with_params(const char* path_model, struct whisper_context_params params);
struct whisper_state* whisper_init_state(struct whisper_context* ctx);
void whisper_free_state(struct whisper_state* state);
void whisper_free(struct whisper_context* ctx);

struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy);
int whisper_full_with_state(struct whisper_context* ctx, struct whisper_state* state, struct whisper_full_params params,
                            const float* samples, int n_samples);
int whisper_full(struct whisper_context* ctx, struct whisper_full_params params, const float* samples, int n_samples);

int whisper_full_n_segments_from_state(struct whisper_state* state);
const char* whisper_full_get_segment_text_from_state(struct whisper_state* state, int i_segment);
int64_t whisper_full_get_segment_t0_from_state(struct whisper_state* state, int i_segment);
int64_t whisper_full_get_segment_t1_from_state(struct whisper_state* state, int i_segment);
bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state* state, int i_segment);
int whisper_full_n_segments(struct whisper_context* ctx);
const char* whisper_full_get_segment_text(struct whisper_context* ctx, int i_segment);
int64_t whisper_full_get_segment_t0(struct whisper_context* ctx, int i_segment);
int64_t whisper_full_get_segment_t1(struct whisper_context* ctx, int i_segment);

int whisper_n_vocab(struct whisper_context* ctx);
int whisper_n_text_ctx(struct whisper_context* ctx);
int whisper_n_audio_ctx(struct whisper_context* ctx);
int whisper_is_multilingual(struct whisper_context* ctx);
whisper_token whisper_token_eot(struct whisper_context* ctx);
whisper_token whisper_token_sot(struct whisper_context* ctx);
whisper_token whisper_token_beg(struct whisper_context* ctx);
const char* whisper_token_to_str(struct whisper_context* ctx, whisper_token token);
int whisper_lang_id(const char* lang);

#ifdef __cplusplus
}
#endif
#endif
