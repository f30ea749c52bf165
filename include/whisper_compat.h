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

/* Which snapshot of whisper.h the by-value structs follow is a BUILD-TIME choice, because the crate's vendored snapshot cannot be read offline
 * (whisper-rs-sys 0.9.0, /root/reference/Cargo.lock:3898-3901):
 *   default                      whisper.h v1.5.0 .. v1.5.4: `whisper_context_params { bool use_gpu; }`, whisper_token_data without t_dtw
 *                                -> the whisper_* symbols inside libspeaksense_hip.so
 *   -DSS_WHISPER_H_POST_1_5_4    whisper.h v1.5.5 (the next release): `gpu_device` and the DTW token-timestamp fields in whisper_context_params
 *                                (48 bytes, passed in memory instead of in a register), `t_dtw` in whisper_token_data (56 instead of 48 bytes,
 *                                returned through a hidden pointer either way)
 *                                -> libspeaksense_whisper_post154.so: the same shim source compiled with the macro, linked BEFORE
 *                                   libspeaksense_hip.so (speaksense_amd/build.py builds both; INTEGRATION.md section A shows the link line)
 * tests/golden/abi_layout.txt and abi_layout_post_1_5_4.txt hold the two layouts field by field (tests/c_harness/layout.c prints them from this
 * header); a maintainer diffs the one that applies against bindgen's layout tests of the vendored header before linking.  v1.6.0 added
 * `bool flash_attn` after `use_gpu`: a third snapshot, not built (no caller of the reference's crate version can see it). */
#ifdef SS_WHISPER_H_POST_1_5_4
enum whisper_alignment_heads_preset {
    WHISPER_AHEADS_NONE, WHISPER_AHEADS_N_TOP_MOST, WHISPER_AHEADS_CUSTOM, WHISPER_AHEADS_TINY_EN, WHISPER_AHEADS_TINY, WHISPER_AHEADS_BASE_EN,
    WHISPER_AHEADS_BASE, WHISPER_AHEADS_SMALL_EN, WHISPER_AHEADS_SMALL, WHISPER_AHEADS_MEDIUM_EN, WHISPER_AHEADS_MEDIUM, WHISPER_AHEADS_LARGE_V1,
    WHISPER_AHEADS_LARGE_V2, WHISPER_AHEADS_LARGE_V3
};
typedef struct whisper_ahead { int n_text_layer; int n_head; } whisper_ahead;
typedef struct whisper_aheads { size_t n_heads; const whisper_ahead* heads; } whisper_aheads;
struct whisper_context_params {
    bool use_gpu;
    int gpu_device;                 /* -> ss_engine_opts.device (env SS_DEVICE overrides) */
    bool dtw_token_timestamps;      /* DTW token timestamps are not implemented: a context created with it logs one line, t_dtw stays -1 */
    enum whisper_alignment_heads_preset dtw_aheads_preset;
    int dtw_n_top;
    struct whisper_aheads dtw_aheads;
    size_t dtw_mem_size;
};
#else
struct whisper_context_params { bool use_gpu; };
#endif

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

typedef struct whisper_token_data {
    whisper_token id;   /* token id */
    whisper_token tid;  /* forced timestamp token id */
    float p;            /* probability of the token */
    float plog;         /* log probability of the token */
    float pt;           /* probability of the timestamp token */
    float ptsum;        /* sum of probabilities of all timestamp tokens */
    int64_t t0;         /* token-level timestamps (whisper_full_params.token_timestamps), 10 ms units; -1 when the flag was off */
    int64_t t1;
#ifdef SS_WHISPER_H_POST_1_5_4
    int64_t t_dtw;      /* DTW token-level timestamp: always -1 here (whisper.cpp's own value when DTW is off) */
#endif
    float vlen;         /* voice length of the token */
} whisper_token_data;

typedef struct whisper_model_loader {
    void* context;
    size_t (*read)(void* ctx, void* output, size_t read_size);
    bool (*eof)(void* ctx);
    void (*close)(void* ctx);
} whisper_model_loader;

typedef void (*ggml_log_callback)(int level, const char* text, void* user_data);

/* ---- the complete function list of whisper.h v1.5.4 ------------------------------------------------------------------------------------
 * Every symbol whisper-rs-sys 0.9.0's bindings can reference resolves against libspeaksense_hip.so.  What the reference calls (listed at the
 * top of this file) is implemented; so are the helpers that need no new device code and, since round 5, the low-level
 * whisper_encode / whisper_decode / whisper_get_logits (whisper-rs: state.encode / state.decode / state.get_logits; the reference does not call
 * them) on the engine's stage hooks.  What this engine has no equivalent for (the phase-vocoder mel, OpenVINO, ggml's CPU benchmarks) returns
 * an error code and logs one line -- never aborts, never pretends to have worked. */
struct whisper_context_params whisper_context_default_params(void);
struct whisper_context_params* whisper_context_default_params_by_ref(void);
struct whisper_context* whisper_init_from_file_with_params(const char* path_model, struct whisper_context_params params);
struct whisper_context* whisper_init_from_buffer_with_params(void* buffer, size_t buffer_size, struct whisper_context_params params);
struct whisper_context* whisper_init_with_params(struct whisper_model_loader* loader, struct whisper_context_params params);
struct whisper_context* whisper_init_from_file_with_params_no_state(const char* path_model, struct whisper_context_params params);
struct whisper_context* whisper_init_from_buffer_with_params_no_state(void* buffer, size_t buffer_size, struct whisper_context_params params);
struct whisper_context* whisper_init_with_params_no_state(struct whisper_model_loader* loader, struct whisper_context_params params);
struct whisper_context* whisper_init_from_file(const char* path_model);                 /* deprecated forms: default context params */
struct whisper_context* whisper_init_from_buffer(void* buffer, size_t buffer_size);
struct whisper_context* whisper_init(struct whisper_model_loader* loader);
struct whisper_context* whisper_init_from_file_no_state(const char* path_model);
struct whisper_context* whisper_init_from_buffer_no_state(void* buffer, size_t buffer_size);
struct whisper_context* whisper_init_no_state(struct whisper_model_loader* loader);
struct whisper_state* whisper_init_state(struct whisper_context* ctx);
int whisper_ctx_init_openvino_encoder(struct whisper_context* ctx, const char* model_path, const char* device, const char* cache_dir);   /* 1: not built (as whisper.cpp without OpenVINO) */
void whisper_free_state(struct whisper_state* state);
void whisper_free(struct whisper_context* ctx);
void whisper_free_params(struct whisper_full_params* params);
void whisper_free_context_params(struct whisper_context_params* params);

/* mel: computed on the device, kept in the state (whisper_n_len reports it); it feeds whisper_encode* and the language detection */
int whisper_pcm_to_mel(struct whisper_context* ctx, const float* samples, int n_samples, int n_threads);
int whisper_pcm_to_mel_with_state(struct whisper_context* ctx, struct whisper_state* state, const float* samples, int n_samples, int n_threads);
int whisper_pcm_to_mel_phase_vocoder(struct whisper_context* ctx, const float* samples, int n_samples, int n_threads);               /* -1 */
int whisper_pcm_to_mel_phase_vocoder_with_state(struct whisper_context* ctx, struct whisper_state* state, const float* samples, int n_samples, int n_threads);
int whisper_set_mel(struct whisper_context* ctx, const float* data, int n_len, int n_mel);
int whisper_set_mel_with_state(struct whisper_context* ctx, struct whisper_state* state, const float* data, int n_len, int n_mel);
/* encoder over the window that starts at mel frame `offset` of the state's spectrogram (whisper_pcm_to_mel* / whisper_set_mel*), then the
 * cross-K/V of every decoder layer: the device kernels of the batch path, one window (ss_encode + ss_session_set_encoder).  0 = ok */
int whisper_encode(struct whisper_context* ctx, int offset, int n_threads);
int whisper_encode_with_state(struct whisper_context* ctx, struct whisper_state* state, int offset, int n_threads);
/* n_tokens tokens at positions n_past .., attending to this state's last whisper_encode* (ss_session_decode).  0 = ok.
 * Deviation from whisper.cpp: the decoder context behind these two calls (one cross-K/V + one self-KV slot) exists ONCE per context, not once
 * per state.  A state keeps it from its whisper_encode* until another state's whisper_encode* or a whisper_full* that runs on the same lane; after
 * that its whisper_decode* returns -1 (never another state's audio) until it calls whisper_encode* again and decodes from n_past = 0.  n_past
 * beyond the positions decoded since the last whisper_encode* is also -1.  An `offset` at or beyond the spectrogram encodes a window of zeros. */
int whisper_decode(struct whisper_context* ctx, const whisper_token* tokens, int n_tokens, int n_past, int n_threads);
int whisper_decode_with_state(struct whisper_context* ctx, struct whisper_state* state, const whisper_token* tokens, int n_tokens, int n_past, int n_threads);
/* raw logits [n_vocab] of the LAST token of the last whisper_decode* (whisper.cpp v1.5.x computes that row only); owned by the state, valid until
 * its next whisper_decode* / whisper_full*; NULL before the first decode */
float* whisper_get_logits(struct whisper_context* ctx);
float* whisper_get_logits_from_state(struct whisper_state* state);

int whisper_tokenize(struct whisper_context* ctx, const char* text, whisper_token* tokens, int n_max_tokens);   /* count, or -needed */
int whisper_lang_max_id(void);
int whisper_lang_id(const char* lang);
const char* whisper_lang_str(int id);
const char* whisper_lang_str_full(int id);
/* runs the engine's detection on the samples last given to whisper_pcm_to_mel*; lang_probs (optional, whisper_lang_max_id()+1 floats): 1.0 at
 * the detected id, 0 elsewhere (the device path keeps the argmax only) */
int whisper_lang_auto_detect(struct whisper_context* ctx, int offset_ms, int n_threads, float* lang_probs);
int whisper_lang_auto_detect_with_state(struct whisper_context* ctx, struct whisper_state* state, int offset_ms, int n_threads, float* lang_probs);

int whisper_n_len(struct whisper_context* ctx);
int whisper_n_len_from_state(struct whisper_state* state);
int whisper_n_vocab(struct whisper_context* ctx);
int whisper_n_text_ctx(struct whisper_context* ctx);
int whisper_n_audio_ctx(struct whisper_context* ctx);
int whisper_is_multilingual(struct whisper_context* ctx);
int whisper_model_n_vocab(struct whisper_context* ctx);
int whisper_model_n_audio_ctx(struct whisper_context* ctx);
int whisper_model_n_audio_state(struct whisper_context* ctx);
int whisper_model_n_audio_head(struct whisper_context* ctx);
int whisper_model_n_audio_layer(struct whisper_context* ctx);
int whisper_model_n_text_ctx(struct whisper_context* ctx);
int whisper_model_n_text_state(struct whisper_context* ctx);
int whisper_model_n_text_head(struct whisper_context* ctx);
int whisper_model_n_text_layer(struct whisper_context* ctx);
int whisper_model_n_mels(struct whisper_context* ctx);
int whisper_model_ftype(struct whisper_context* ctx);
int whisper_model_type(struct whisper_context* ctx);
const char* whisper_model_type_readable(struct whisper_context* ctx);

const char* whisper_token_to_str(struct whisper_context* ctx, whisper_token token);
whisper_token whisper_token_eot(struct whisper_context* ctx);
whisper_token whisper_token_sot(struct whisper_context* ctx);
whisper_token whisper_token_solm(struct whisper_context* ctx);
whisper_token whisper_token_prev(struct whisper_context* ctx);
whisper_token whisper_token_nosp(struct whisper_context* ctx);
whisper_token whisper_token_not(struct whisper_context* ctx);
whisper_token whisper_token_beg(struct whisper_context* ctx);
whisper_token whisper_token_lang(struct whisper_context* ctx, int lang_id);
whisper_token whisper_token_translate(struct whisper_context* ctx);
whisper_token whisper_token_transcribe(struct whisper_context* ctx);

void whisper_print_timings(struct whisper_context* ctx);
void whisper_reset_timings(struct whisper_context* ctx);
const char* whisper_print_system_info(void);
void whisper_log_set(ggml_log_callback log_callback, void* user_data);

struct whisper_full_params* whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy);
struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy);
/* whisper_full_params callbacks fire at CHUNK granularity (a chunk's windows run inside a device batch shared with other states): abort_callback and
 * encoder_begin_callback once before the chunk is submitted (true / false -> -6), progress_callback(100) and new_segment_callback(n_new = all
 * segments of the call) once when it has completed.  logits_filter_callback and grammar rules are refused (-9). */
int whisper_full(struct whisper_context* ctx, struct whisper_full_params params, const float* samples, int n_samples);
int whisper_full_with_state(struct whisper_context* ctx, struct whisper_state* state, struct whisper_full_params params,
                            const float* samples, int n_samples);
/* splits the audio into n_processors chunks that run as ONE device batch (whisper.cpp: one thread + state each), merged with its offset rule */
int whisper_full_parallel(struct whisper_context* ctx, struct whisper_full_params params, const float* samples, int n_samples, int n_processors);

int whisper_full_n_segments(struct whisper_context* ctx);
int whisper_full_n_segments_from_state(struct whisper_state* state);
int whisper_full_lang_id(struct whisper_context* ctx);
int whisper_full_lang_id_from_state(struct whisper_state* state);
int64_t whisper_full_get_segment_t0(struct whisper_context* ctx, int i_segment);
int64_t whisper_full_get_segment_t0_from_state(struct whisper_state* state, int i_segment);
int64_t whisper_full_get_segment_t1(struct whisper_context* ctx, int i_segment);
int64_t whisper_full_get_segment_t1_from_state(struct whisper_state* state, int i_segment);
bool whisper_full_get_segment_speaker_turn_next(struct whisper_context* ctx, int i_segment);
bool whisper_full_get_segment_speaker_turn_next_from_state(struct whisper_state* state, int i_segment);
const char* whisper_full_get_segment_text(struct whisper_context* ctx, int i_segment);
const char* whisper_full_get_segment_text_from_state(struct whisper_state* state, int i_segment);
int whisper_full_n_tokens(struct whisper_context* ctx, int i_segment);
int whisper_full_n_tokens_from_state(struct whisper_state* state, int i_segment);
const char* whisper_full_get_token_text(struct whisper_context* ctx, int i_segment, int i_token);
const char* whisper_full_get_token_text_from_state(struct whisper_context* ctx, struct whisper_state* state, int i_segment, int i_token);
whisper_token whisper_full_get_token_id(struct whisper_context* ctx, int i_segment, int i_token);
whisper_token whisper_full_get_token_id_from_state(struct whisper_state* state, int i_segment, int i_token);
whisper_token_data whisper_full_get_token_data(struct whisper_context* ctx, int i_segment, int i_token);
whisper_token_data whisper_full_get_token_data_from_state(struct whisper_state* state, int i_segment, int i_token);
float whisper_full_get_token_p(struct whisper_context* ctx, int i_segment, int i_token);
float whisper_full_get_token_p_from_state(struct whisper_state* state, int i_segment, int i_token);

int whisper_bench_memcpy(int n_threads);                       /* CPU micro-benchmarks of ggml: nothing to run here */
const char* whisper_bench_memcpy_str(int n_threads);
int whisper_bench_ggml_mul_mat(int n_threads);
const char* whisper_bench_ggml_mul_mat_str(int n_threads);

#ifdef __cplusplus
}
#endif
#endif
