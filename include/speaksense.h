/* speaksense.h -- C ABI of libspeaksense_hip.so: the MI355X-native Whisper transcription path.
 *
 * Drop-in boundary.  The reference reaches its inference engine only through
 *   /root/reference/src/asr/mod.rs:58-73     trait AsrEngine { create_state, transcribe_with_state, transcribe }
 *   /root/reference/src/asr/whisper.rs:21-28 WhisperAsr::new           -> ss_engine_create
 *   /root/reference/src/asr/whisper.rs:30-39 WhisperAsr::create_state  -> ss_session_create
 *   /root/reference/src/asr/whisper.rs:75    state.full(params, &audio)-> ss_transcribe / ss_submit + ss_wait
 *   /root/reference/src/asr/whisper.rs:77-95 full_n_segments / full_get_segment_{text,t0,t1,speaker_turn_next}
 *                                                                      -> ss_result_*
 * Plain pointers and sizes only; 0 = ok, negative = failure; no exceptions cross this boundary.
 * A whisper.h-compatible subset (what whisper-rs-sys binds) is in whisper_compat.h and is implemented on top
 * of these entry points.  INTEGRATION.md shows the Rust-side binding.
 *
 * Threading: an engine is shared by any number of sessions and threads (the reference shares one
 * Arc<WhisperContext> between tokio tasks, whisper.rs:17,26).  A session is used by one caller at a time
 * (the reference guards it with a Mutex, whisper.rs:38,51).  ss_submit never blocks on the GPU: the engine's
 * batch former groups queued 30 s chunks from different sessions into one device batch.
 */
#ifndef SPEAKSENSE_H
#define SPEAKSENSE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ss_engine ss_engine;
typedef struct ss_session ss_session;
typedef struct ss_ticket ss_ticket;

/* ABI check for bindings.  ss_params / ss_engine_opts are passed by pointer and COPIED by the library (`P = *params`), so a caller built against an
 * older header would make it read past the caller's struct.  A binding asserts, once at load time, that ss_abi_version() == SS_ABI_VERSION and
 * that ss_sizeof_params() / ss_sizeof_engine_opts() equal its own sizeof (binding.py and rust/asr_hip.rs do).  Bumped whenever a struct of this
 * header changes size or a field changes meaning: 4 = round 4's ss_params (128 bytes, token_timestamps in the former reserved0);
 * 5 = round 5 (ss_process_logits_row, these three getters; layouts unchanged); 6 = ss_params grows by suppress_non_speech_tokens / max_len /
 * split_on_word (144 bytes). */
#define SS_ABI_VERSION 6
int32_t ss_abi_version(void);
int32_t ss_sizeof_params(void);
int32_t ss_sizeof_engine_opts(void);

enum { SS_DTYPE_BF16 = 0, SS_DTYPE_F16 = 1, SS_DTYPE_FP8 = 2 };
enum {
    SS_OK = 0,
    SS_ERR_ARG = -1,       /* null / out-of-range argument */
    SS_ERR_MODEL = -2,     /* model file missing, bad magic, unsupported tensor type, missing tensor */
    SS_ERR_LANG = -3,      /* unknown language for a multilingual model */
    SS_ERR_DEVICE = -4,    /* no HIP device / HIP runtime error (message via ss_last_error) */
    SS_ERR_AUDIO_CTX = -5, /* audio_ctx larger than the model's (whisper_full returns -5) */
    SS_ERR_BUFFER = -6,    /* caller's output buffer too small (ss_*_tokenize) */
    SS_ERR_UNSUPPORTED = -9
};

typedef struct ss_engine_opts {
    int32_t device;       /* HIP device ordinal (one engine per GPU; one process per GPU under torchrun) */
    int32_t dtype;        /* SS_DTYPE_F16 (ggml's arithmetic type: the parity configuration; used when opts == NULL) | SS_DTYPE_BF16 |
                             SS_DTYPE_FP8: the f16 engine with the encoder-block and cross-KV projections (98 % of the path's FLOPs) on OCP e4m3
                             weights and activations, MX-scaled fp8 MFMA; needs n_audio_state % 256 == 0 (base and larger).  No reference
                             counterpart: whisper.cpp has no fp8; parity is against the oracle's FP8 mode with the same rounding points */
    int32_t max_batch;    /* windows encoded+decoded together on the device (default 8, <= 128; max_batch x max_decoders <= 1024).  Device memory PER LANE,
                             B = max_batch, ND = max_decoders, d = n_audio_state = n_text_state, L = n_text_layer, V = n_vocab rounded up to 64:
                               cross-KV cache  L x B x 2 x 1500 x d x 2 bytes (1 byte + 1/64 in SS_DTYPE_FP8)      large-v3: 245.8 MB per window
                               self-KV caches  L x (B x ND) x 2 x 448 x d x 2 bytes                                large-v3:  73.4 MB per decoder
                               encoder workspace  ~ B x 1500 x d x 26 bytes                                        large-v3:  50 MB per window
                               decoder rows    128 x (2 V x 4 + 14 d) bytes                                        large-v3:  56 MB
                             plus the weights once per engine (f16 file size + 0.9 GB of e4m3 copies in SS_DTYPE_FP8).  large-v3, B = 32, ND = 5:
                             ~21 GB per lane.  ss_engine_create adds this up, compares it with hipMemGetInfo and fails with SS_ERR_ARG and the numbers
                             in ss_last_error() BEFORE allocating anything if it does not fit (B = 128, ND = 5, 3 lanes of large-v3 would need ~250 GB) */
    int32_t max_decoders; /* decoders per window at temperature > 0 (reference: Greedy{best_of:5}, whisper.rs:132) */
    int32_t batch_wait_us;/* how long the batch former waits for more chunks before launching a partial batch */
    int32_t n_lanes;      /* device batches in flight at once over one copy of the weights (0 = default 2; env SS_LANES overrides): each lane has
                             its own stream, workspaces and KV caches, so one group's encoder pass and another's decode chain overlap */
    int32_t compat;       /* SS_COMPAT_* flags: which variant of a whisper.cpp-version-dependent behaviour the engine reproduces.  0 = whisper.cpp
                             v1.5.0 .. v1.5.4, the range whisper-rs-sys 0.9.0 vendors (/root/reference/Cargo.lock:3888-3907); env SS_COMPAT (a number
                             or names joined by '+': "rng_state", "openai_ts_rules") overrides it -- the whisper.h shim has no options struct */
    int32_t reserved;
} ss_engine_opts;

/* Version-dependent behaviour (DESIGN.md section 2 holds the ledger: fact, upstream version range of each variant, switch).  Both variants of
 * every flag are under the oracle's tests (same flag values there: oracle/binding.py COMPAT_*). */
enum {
    SS_COMPAT_RNG_STATE = 1,       /* whisper.cpp <= v1.4.x: ONE std::mt19937(0) in whisper_state; the best_of decoders draw from it in decoder order.
                                      Default (flag clear, >= v1.5.0): every decoder owns a generator -- decoder 0's is seeded once per state and carried
                                      across calls, decoders 1.. are re-seeded with 0 by every whisper_full call */
    SS_COMPAT_OPENAI_TS_RULES = 2, /* OpenAI's timestamp rules where whisper.cpp's differ: the first sampled token must be a timestamp (whisper.cpp only
                                      applies max_initial_ts); a timestamp may not repeat the last one unless a pair is open (`<=`; whisper.cpp `<`);
                                      <|0.00|> counts as a timestamp seen (whisper.cpp: id > token_beg) */
    SS_COMPAT_OPENAI_HISTORY = 4   /* what a later window of ONE call is conditioned on ([prev] + history + [sot ..]; `no_context` only clears the history a call
                                      starts with): OpenAI / HF take the tokens of the window's segments -- a window ending in a timestamp pair contributes
                                      everything but the pair's second timestamp -- and at most 222 of them; whisper.cpp (default) every token up to
                                      result_len, at most 224.  Exists so that multi-window sequences can be held to HF generate() */
};

/* The whisper_full_params fields the reference sets (whisper.rs:131-173) plus its per-request overrides
 * (language whisper.rs:60-63, stream mode whisper.rs:65-69, tdrz whisper.rs:136-139).  ss_default_params()
 * returns exactly what build_params() + stream-mode produce. */
typedef struct ss_params {
    int32_t best_of;          /* 5 */
    float temperature;        /* 0.0 */
    float temperature_inc;    /* 0.2 (whisper_full_default_params) */
    float entropy_thold;      /* 2.4 */
    float logprob_thold;      /* -1.0 */
    float max_initial_ts;     /* 1.0 */
    float length_penalty;     /* -1.0 */
    int32_t no_context;       /* 1 in stream mode (both reference callers); 0 keeps the session's prompt_past across calls */
    int32_t single_segment;   /* 0 */
    int32_t no_timestamps;    /* 0 */
    int32_t suppress_blank;   /* 1 */
    int32_t tdrz_enable;      /* AsrParams.speaker_diarization */
    int32_t print_special;    /* 0 */
    int32_t max_tokens;       /* 0 */
    int32_t audio_ctx;        /* 0 = full (n_audio_ctx = 1500; whisper.rs:144 passes 1500).  A smaller positive multiple of 4 shortens the encoder to the first
                                 audio_ctx positions (whisper.cpp's exp_n_audio_ctx): 2 audio_ctx mel frames per window, that many cross-attention keys;
                                 larger than the model's -> SS_ERR_AUDIO_CTX (-5, as whisper_full); other values and language detection with a
                                 shortened context -> SS_ERR_UNSUPPORTED */
    int32_t translate;        /* 0 */
    int32_t fixed_steps;      /* bench Mode F: >0 = exactly this many greedy steps, EOT suppressed, no fallback */
    char language[8];         /* "en" default; AsrParams.language.  "auto" or "" = detect (whisper_lang_auto_detect on the first window) */
    /* ---- whisper_full_params fields the reference leaves at their defaults but a whisper-rs caller can reach ---- */
    int32_t n_max_text_ctx;   /* 16384: how much of prompt_past conditions the next window (min with n_text_ctx/2) */
    int32_t offset_ms;        /* 0: start offset */
    int32_t duration_ms;      /* 0 = to the end */
    int32_t detect_language;  /* 1 = only detect the language (result: ss_result_lang_id), no transcription */
    const int32_t* prompt_tokens;  /* prepended to the session's prompt_past (whisper_full_params.prompt_tokens); copied by the call */
    int32_t prompt_n_tokens;
    int32_t token_timestamps; /* 1 (whisper.rs:160): whisper.cpp's token-level times for every segment's tokens (ss_result_segment_token_times); with
                                 max_len = 0 (whisper.rs:167) they change neither the text nor the segment times */
    const char* initial_prompt;    /* used when prompt_tokens is NULL: tokenised with the model's vocabulary (whisper_tokenize) */
    float thold_pt;           /* 0.01 (whisper.rs:170): a token's best-timestamp share must exceed this to pin its start time */
    float thold_ptsum;        /* 0.01 (whisper.rs:171): and so must its total timestamp probability */
    int32_t suppress_non_speech_tokens; /* 0 (whisper.rs:156): 1 = whisper.cpp's non_speech_tokens list (symbols, brackets, music notes; with and
                                 without a leading space) and " -" / " '" are masked at every step, as far as the model's vocabulary has them */
    int32_t max_len;          /* 0 (whisper.rs:167): > 0 = whisper_wrap_segment -- every segment is cut into pieces of at most max_len bytes of text at
                                 token boundaries, the pieces' times taken from the token-level times (acts only with token_timestamps = 1) */
    int32_t split_on_word;    /* 1 (whisper.rs:161): with max_len > 0, cut only before a token that starts with ' ' */
} ss_params;

void ss_default_params(ss_params* p);
const char* ss_last_error(void); /* thread-local, valid until the next failing call on this thread */

/* ---- engine / session lifetime ---------------------------------------------------------------------- */
int ss_engine_create(const char* ggml_model_path, const ss_engine_opts* opts, ss_engine** out);
/* May be called with tickets outstanding: chunks still queued fail with SS_ERR_DEVICE, chunks a device group is running finish normally, threads
 * blocked in ss_wait are woken and have left the engine before this returns.  Tickets waited for LATER return their recorded status without
 * touching the engine, sessions may be freed later too.  What the caller must not do is let another thread ENTER ss_submit / ss_wait / a blocking
 * transcribe call on this engine while it is being freed (the reference drops its Arc<WhisperContext> only when the last user is gone,
 * src/asr/whisper.rs:17,26). */
void ss_engine_free(ss_engine* e);
int ss_engine_hparams(const ss_engine* e, int32_t out11[11]);       /* n_vocab ... ftype, ggml header order */
int ss_engine_special_tokens(const ss_engine* e, int32_t out9[9]);  /* eot sot translate transcribe solm prev nosp not beg */
const char* ss_engine_token_str(const ss_engine* e, int32_t id);
/* whisper_tokenize: text -> ids with whisper.cpp's rule (GPT-2 pre-split, then greedy longest match in the vocabulary).
 * Returns the number of tokens (>= 0); SS_ERR_BUFFER when n_max is too small (nothing written).  ids == NULL with n_max == 0 is the size
 * query: returns the number of tokens the text needs.  Both tokenize entry points follow this one convention. */
int ss_engine_tokenize(const ss_engine* e, const char* text, int32_t* ids, int32_t n_max);
/* the same with only the vocabulary of a model file loaded (host only: no GPU, no engine); negative = error */
int ss_model_tokenize(const char* ggml_model_path, const char* text, int32_t* ids, int32_t n_max);

ss_session* ss_session_create(ss_engine* e);
/* A session whose chunks are still queued or running is written to by the engine's workers: ss_session_free BLOCKS until the engine has completed
 * every chunk submitted on it (completion does not need ss_wait; after ss_engine_free everything is complete), then frees it.  The reference owns
 * its WhisperState the same way: behind a Mutex held for the whole transcribe call (src/asr/whisper.rs:34-38,51). */
void ss_session_free(ss_session* s);

/* ---- one model on several GPUs of a node (wiring it slots under: /root/reference/src/main.rs:38-39,59,71 -- one
 * WhisperAsr shared by the gRPC handler and the REST worker) ----------------------------------------------------------------------------
 * Chunks are independent (both reference callers force no_context), so the pool is N engines -- one per listed HIP device, weights replicated,
 * no collective -- and a router: every submitted chunk goes to the engine with the fewest chunks queued or running, ties round-robin
 * (north_star: "sharded round-robin across the 8 GPUs").  A session's state (prompt_past, sampler) lives on the host, so its chunks may run
 * on different GPUs; results come back on the session as usual.  device_ids may repeat (tests: two engines on one GPU). */
typedef struct ss_pool ss_pool;
int ss_pool_create(const char* ggml_model_path, const int32_t* device_ids, int32_t n_devices, const ss_engine_opts* opts /* .device ignored */,
                   ss_pool** out);
void ss_pool_free(ss_pool* p);
int32_t ss_pool_n_engines(const ss_pool* p);
ss_engine* ss_pool_engine(ss_pool* p, int32_t i);              /* borrowed: hparams, tokens, timing of engine i */
ss_session* ss_pool_session_create(ss_pool* p);               /* freed with ss_session_free */
/* ss_wait as usual.  A session may have several tickets outstanding (as on a single engine: its chunks then run one after another in
 * submission order); while it has, its chunks stay on the engine that holds them -- the router only moves a session between chunks. */
int ss_pool_submit(ss_pool* p, ss_session* s, const float* pcm, int32_t n_samples, const ss_params* params, ss_ticket** out);
int32_t ss_pool_last_engine(const ss_session* s);             /* index of the engine that ran (or runs) the session's last chunk */
/* the routing rule alone (host only): index of the engine a chunk goes to given each engine's load and the round-robin cursor */
int32_t ss_pool_pick(const int32_t* load, int32_t n_engines, uint32_t cursor);

/* ---- transcription ---------------------------------------------------------------------------------- */
/* Blocking: n chunks (one per session, sessions distinct) processed as one device batch stream.
 * pcm[i]: n_samples[i] mono f32 samples @16 kHz in [-1,1].  pcm_on_device != 0: pcm[i] are device pointers
 * on the engine's GPU (bench: inputs resident in HBM).  Returns 0 or the first failing chunk's error. */
int ss_transcribe_batch(ss_engine* e, ss_session* const* sessions, const float* const* pcm, const int32_t* n_samples,
                        int32_t n, const ss_params* params, int32_t pcm_on_device);
/* One chunk on one session (== whisper_full_with_state). */
int ss_transcribe(ss_session* s, const float* pcm, int32_t n_samples, const ss_params* params);
/* Non-blocking: copies the samples, returns a ticket; the engine batches across sessions. */
int ss_submit(ss_session* s, const float* pcm, int32_t n_samples, const ss_params* params, ss_ticket** out);
/* The same; pcm_on_device != 0: `pcm` is a device pointer on the engine's GPU, not copied -- it must stay valid until ss_wait returns. */
int ss_submit_ex(ss_session* s, const float* pcm, int32_t n_samples, const ss_params* params, int32_t pcm_on_device, ss_ticket** out);
/* Blocks until the chunk is complete; returns its status; frees the ticket.  Valid after its session or its engine has been freed (the status was
 * recorded on the ticket; results live on the session and are gone with it).  A ticket that is never waited for leaks its own memory (the copy of
 * the samples) and nothing else: the engine keeps no reference to it once the chunk is complete. */
int ss_wait(ss_ticket* t);
/* Non-blocking: 1 when the chunk is complete (ss_wait will not block), 0 while it is queued or running.  The ticket stays valid. */
int ss_ticket_ready(const ss_ticket* t);

/* ---- results of the session's last chunk (valid until its next transcribe/submit) --------------------
 * With another chunk of the SAME session queued or running the engine is writing these: reading them then is a data race, and the bulk getters
 * (ss_result_tokens / _sampled_tokens / _trace_tokens) fill arrays the caller sized from an earlier ss_result_n_* call.  Wait for every ticket of a
 * session before reading its results (the reference holds its state's Mutex for the whole call, src/asr/whisper.rs:34-38,51). */
int32_t ss_result_n_segments(const ss_session* s);
const char* ss_result_segment_text(const ss_session* s, int32_t i);   /* bytes from the vocab, NUL-terminated */
int64_t ss_result_segment_t0(const ss_session* s, int32_t i);         /* centiseconds, as whisper.cpp reports */
int64_t ss_result_segment_t1(const ss_session* s, int32_t i);
int32_t ss_result_segment_speaker_turn_next(const ss_session* s, int32_t i);
int32_t ss_result_segment_n_tokens(const ss_session* s, int32_t i);     /* whisper_full_n_tokens: tokens of segment i (timestamp tokens included) */
int ss_result_segment_token(const ss_session* s, int32_t i, int32_t k, int32_t* id, int32_t* tid, float out4[4] /* p, plog, pt, ptsum */);
/* whisper_token_data.t0 / t1 / vlen of token k of segment i: token-level times in 10 ms units from whisper.cpp's whisper_exp_compute_token_level_timestamps
 * (timestamp-token evidence, voice-length interpolation, signal-energy adjustment); -1 / -1 / 0 when ss_params.token_timestamps was 0 */
int ss_result_segment_token_times(const ss_session* s, int32_t i, int32_t k, int64_t* t0, int64_t* t1, float* vlen);
/* append src's segments to dst shifted by t_offset centiseconds, start times clamped to the previous end (the merge step of whisper_full_parallel) */
int ss_result_append(ss_session* dst, const ss_session* src, int64_t t_offset);
int32_t ss_result_n_tokens(const ss_session* s);                      /* accepted tokens over all windows */
int ss_result_tokens(const ss_session* s, int32_t* ids, float* plog); /* plog may be NULL */
/* every id the winning decoder of each window SAMPLED, in order, including the tail past result_len that whisper.cpp (and ss_result_tokens)
 * drops: the stream a step-by-step replay on another implementation needs (tests: forced replay on the oracle) */
int32_t ss_result_n_sampled_tokens(const ss_session* s);
int ss_result_sampled_tokens(const ss_session* s, int32_t* ids);
/* every id ANY decoder of the chunk sampled -- the attempts that failed and the best_of decoders that lost included -- in the order
 * whisper_full_with_state calls whisper_sample_token: window, temperature attempt, step, decoder.  With it another implementation can replay the
 * chunk call by call, sampled (t > 0) attempts included: it draws the same uniforms from the same generator and checks that each id here is
 * the one its own cumulative distribution selects, or sits within rounding of the boundary (tests: oracle `full(trace=...)`). */
int32_t ss_result_n_trace_tokens(const ss_session* s);
int ss_result_trace_tokens(const ss_session* s, int32_t* ids);
int32_t ss_result_lang_id(const ss_session* s);                        /* language used by the last chunk (whisper_full_lang_id), -1 for .en models */
int ss_result_counters(const ss_session* s, int32_t out4[4]);         /* n_encode, n_decode_steps, n_fail, n_windows */

/* ---- the session's sampler state --------------------------------------------------------------------- */
/* The one piece of sampler state a whisper_state carries from call to call is ONE std::mt19937(0) that is never reseeded: decoder 0's generator
 * (whisper.cpp >= v1.5.0, the default; decoders 1.. are re-seeded by every call and carry nothing) or whisper_state::rng shared by all decoders
 * (SS_COMPAT_RNG_STATE).  So in the reference (one state per stream / per REST task, chunks one after another) a chunk that falls back to
 * sampling sees that generator advanced by every earlier chunk's draws from it.  ss_session_rng_draws: invocations of the carried generator
 * consumed by this session so far.  ss_session_rng_discard: advance the carried generator of a (fresh) session by n invocations -- with the
 * two, chunks of one reference state can be decoded as a batch on fresh sessions and still give the serial result
 * (speaksense_amd/rest.py::TranscribeProcessor).  ss_session_rng_draws_decoder: the same count for the generator of decoder j >= 1 since the
 * last chunk started (0 under SS_COMPAT_RNG_STATE, where those decoders own none). */
int64_t ss_session_rng_draws(const ss_session* s);
int ss_session_rng_discard(ss_session* s, int64_t n);
int64_t ss_session_rng_draws_decoder(const ss_session* s, int32_t decoder);

/* ---- per-stage hooks for parity tests (host f32 in / out; each runs the same device kernels) -------- */
int32_t ss_mel_n_len(int32_t n_samples);
int ss_log_mel(ss_engine* e, const float* pcm, int32_t n_samples, float* mel_out /* [n_mel][n_len] */, int32_t n_len);
/* whisper.cpp get_signal_energy(pcm, n, 32), the per-sample signal the token-level timestamps consult (host in, host out; same kernel as the batch path) */
int ss_signal_energy(ss_engine* e, const float* pcm, int32_t n_samples, float* energy_out /* [n_samples] */);
int ss_encode(ss_engine* e, const float* mel /* [n_mel][n_len] */, int32_t n_len, int32_t seek,
              float* enc_out /* [n_audio_ctx][n_audio_state] */);
/* the same over a shortened context (ss_params.audio_ctx: a positive multiple of 4 <= n_audio_ctx): the first audio_ctx positions only */
int ss_encode_ctx(ss_engine* e, const float* mel /* [n_mel][n_len] */, int32_t n_len, int32_t seek, int32_t audio_ctx,
                  float* enc_out /* [audio_ctx][n_audio_state] */);
/* Decoder hook on a session: set encoder output (computes cross-KV), then decode tokens at n_past; logits_out: [n_vocab] of the LAST token,
 * before any rule.  The hook's cross-K/V and self-KV live in ONE slot per engine (lane 0, slot 0), owned by the session that last called
 * ss_session_set_encoder*: ss_session_decode from any other session, after a transcription ran on lane 0, or with n_past beyond the positions
 * decoded since that call, fails with SS_ERR_ARG instead of answering from someone else's audio. */
int ss_session_set_encoder(ss_session* s, const float* enc /* [n_audio_ctx][n_audio_state] */);
/* the same for the output of a shortened context ([audio_ctx][n_audio_state], ss_encode_ctx): ss_session_decode then attends over audio_ctx keys */
int ss_session_set_encoder_ctx(ss_session* s, const float* enc, int32_t audio_ctx);
int ss_session_decode(ss_session* s, const int32_t* tokens, int32_t n_tokens, int32_t n_past, float* logits_out);
/* The decoder PASS the batched engine runs, as a stage hook: cross-KV cache slot `window` (< max_batch, lane 0) is filled from an encoder
 * output, then ONE launch carries n_rows (1..128) token rows -- row i = token[i] at position pos[i] of self-KV slot slot[i]
 * (< max_batch * max_decoders) attending to cross-KV window cross[i]; rows of one slot must be at consecutive positions, earlier rows first.
 * The kernels are selected by the row count exactly as in ss_transcribe_batch (the GEMVs take 1 / 2 / 4 / 8 column tiles of 16 rows; rows x
 * heads >= 320: the one-workgroup cross-attention; a row's logits are bit-identical whichever forms its pass selects).  logits_out: [n_sample_rows][n_vocab] raw logits of the listed rows, before any rule. */
int ss_engine_set_encoder_window(ss_engine* e, int32_t window, const float* enc /* [n_audio_ctx][n_audio_state] */);
/* fp8 engines only (SS_ERR_UNSUPPORTED otherwise): the FIRST quantisation point of the path -- LayerNorm 1 of encoder block 0 for the window at
 * `seek` of a log-mel spectrogram -- as the e4m3 projections read it: codes [n_audio_ctx][n_audio_state] and one E8M0 exponent byte per
 * (row, 64-column block) [n_audio_ctx][n_audio_state / 64]; value = e4m3(code) * 2^(exp - 127).  tests/test_gpu_fp8.py counts the codes that
 * differ from the oracle's. */
int ss_engine_fp8_first_quant(ss_engine* e, const float* mel, int32_t n_len, int32_t seek, uint8_t* codes, uint8_t* exps);
int ss_engine_decode_rows(ss_engine* e, const int32_t* token, const int32_t* pos, const int32_t* slot, const int32_t* cross, int32_t n_rows,
                          const int32_t* sample_rows, int32_t n_sample_rows, float* logits_out);
/* Fused logits rules + log-softmax + greedy pick on the device for one row of raw logits.
 * hist: tokens sampled so far in this window.  out6: id, p, plog, tid, pt, ptsum. */
int ss_process_logits(ss_engine* e, const float* raw_logits, const int32_t* hist, int32_t n_hist, int32_t has_ts,
                      int32_t seek_delta, const ss_params* params, float out6[6]);
/* The same call, also returning the processed row: logprobs_out[n_vocab] = log-softmax over the ids the rules leave, -INFINITY where a rule masks
 * the id (whisper_process_logits' `logprobs`).  tests/test_gpu_golden.py holds every mask bit to tests/golden/hf_rules_golden.npz. */
int ss_process_logits_row(ss_engine* e, const float* raw_logits, const int32_t* hist, int32_t n_hist, int32_t has_ts,
                          int32_t seek_delta, const ss_params* params, float out6[6], float* logprobs_out);

/* ---- audio pre-stage (SURVEY.md §8f "next" #1) ------------------------------------------------------ */
/* `denoise_audio(samples, &DenoiseConfig)` of /root/reference/src/audio/mod.rs:507-523 (the call the gRPC handler makes at
 * src/grpc/handlers/asr.rs:196 and the stream pre-processor at mod.rs:133-134) on the engine's GPU.  Same semantics incl. the
 * unnormalised inverse FFT and the x10 gain.  frame_size must be 2048 (the default); n >= 2048 (the reference panics below).
 * force_type: -1 = decide as the reference does; 0/1/2 force Stationary/NonStationary/Mixed (tests). */
typedef struct ss_denoise_config {   /* DenoiseConfig, mod.rs:41-48 */
    int32_t frame_size; float overlap; float strength; float noise_gate; int32_t enable_noise_reduction; float threshold;
} ss_denoise_config;
void ss_default_denoise_config(ss_denoise_config* c);
int ss_denoise_audio(ss_engine* e, const float* pcm, int32_t n_samples, const ss_denoise_config* cfg, int32_t force_type, float* out,
                     int32_t* noise_type, float* norm_var, float* device_ms);

/* ---- file front end (SURVEY.md §8f "next" #4) --------------------------------------------------------- */
/* `create_resampler(from, 16000)` + `resample_chunk` per 4096-sample read (/root/reference/src/audio/mod.rs:235-257: rubato 0.16.0
 * SincFixedIn, sinc_len 256, cutoff 0.95, linear, oversampling 256, BlackmanHarris2) for a whole MONO stream in one call.
 * Only the n / 4096 full read chunks are resampled: the reference's resampler rejects the short last read (and every read of a
 * multi-channel file), which ends that file's processing.  chunk_lens (optional, n / 4096 entries): samples produced per read chunk,
 * i.e. the normalisation chunks ss_preprocess_stream needs.  out_cap >= ss_resample_max_out(n, from_rate). */
int64_t ss_resample_max_out(int64_t n_samples, int32_t from_rate);
int ss_resample_stream(ss_engine* e, const float* pcm, int64_t n_samples, int32_t from_rate, float* out, int64_t out_cap, int64_t* n_out,
                       int32_t* chunk_lens, float* device_ms);

/* The stream pre-processor in front of the REST path: `StreamAudioProcessor` (/root/reference/src/audio/mod.rs:67-155) as
 * parse_audio_file_stream drives it (mod.rs:158-233), for a whole mono 16 kHz stream in one call: per-read-chunk peak normalisation
 * (mod.rs:91, 408-411), 2048-sample frames (last one zero padded, mod.rs:143-154), energy gain (110-130; incl. the reference's NaN noise floor,
 * which makes the gain 0.1), per-frame denoise_audio (133-134) and the noise gate (138).  Read chunks: `chunk_lens[n_chunks]` (what the
 * resampler returned per 4096-sample read) or, when NULL, uniform `chunk_len` (4096 / channels for a 16 kHz file).
 * out: ss_preprocess_n_out(n) floats = the concatenated callbacks the REST chunker (schedule/processors/transcribe.rs:100-142) buffers.
 * gains_out (optional): one gain per frame. */
int64_t ss_preprocess_n_out(int64_t n_samples);
int ss_preprocess_stream(ss_engine* e, const float* pcm, int64_t n_samples, const int32_t* chunk_lens, int32_t n_chunks, int32_t chunk_len,
                         const ss_denoise_config* cfg, float* out, float* gains_out, float* device_ms);

/* ---- timing hooks for bench.py (HIP events on the engine's own stream) ------------------------------ */
/* ms spent in the phases of the last ss_transcribe_batch: [0] mel, [1] encoder+cross-KV, [2] decode, [3] total */
int ss_engine_last_timing(const ss_engine* e, float out_ms[4]);
/* work of the last device group: [0] decoder passes (each streams the decoder weights once: prompt pass + one per later step),
 * [1] decoder rows over those passes, [2] encoder windows, [3] 0.  With last_timing[2] this gives the in-pipeline decode-step time. */
int ss_engine_last_counters(const ss_engine* e, int64_t out4[4]);
/* Cumulative since engine creation, summed over the lanes: device ms [mel, encoder+cross-KV, decode, total] and work [decoder passes, decoder
 * rows, encoder windows, chunks admitted into a running group, windows started while other windows of their group were decoding, decode-step
 * graphs evicted from the per-lane LRU of 256 (a count that keeps growing means the (rows, sampled rows) shapes of the load thrash it)].  Differences around a timed region give the in-pipeline averages when several groups run concurrently. */
int ss_engine_totals(const ss_engine* e, double out_ms[4], int64_t out_cnt[6], int32_t* n_lanes);
/* The same counters for one lane (0 <= lane < n_lanes): how the batch former spread the work (tests: lane levelling). */
int ss_engine_lane_counters(const ss_engine* e, int32_t lane, int64_t out_cnt[6]);
/* hipMemGetInfo on the engine's device: bytes free / total (leak checks of the soak test; a service's health endpoint) */
int ss_engine_mem_info(const ss_engine* e, int64_t* free_bytes, int64_t* total_bytes);
/* Element offset of entry (row n, column k) of a [N][K] matrix in the fragment-major layout the decode-step GEMVs read their weights and their
 * activation rows in (kernels.h dec_wpack_off: the 16 x 32 block one MFMA 16x16x32 consumes is one contiguous kilobyte).  Host only, no GPU: the
 * layout is a permutation of [0, N K) for N % 16 == 0, K % 32 == 0, which the CPU tests check.  Returns -1 on a bad argument. */
int64_t ss_dec_weight_offset(int64_t n, int32_t k, int32_t K);
/* average device time (ms) of `reps` launches of the dominant encoder GEMM (FC1: M=batch*1500, N=4d, K=d) on the
 * engine's stream, and its algorithmic FLOPs per launch: the roofline probe bench.py reports. */
int ss_engine_probe_gemm(ss_engine* e, int32_t batch, int32_t reps, float* avg_ms, double* flops_per_launch);

/* Self-test of the tiled MFMA GEMM at an arbitrary shape (N % 128 == 0, K % 64 == 0) in the engine's operand type: seeded operands, result
 * compared on the device with a one-thread-per-output reference.  kind: 0 bias -> T, 1 bias + GELU -> T, 2 bias + f32 residual (in place),
 * 6 bias -> f32.  The parity tests run small models (one tile per workgroup); this reaches the multi-tile paths at the large-v3 shapes. */
int ss_engine_selftest_gemm(ss_engine* e, int32_t M, int32_t N, int32_t K, int32_t kind, float* max_err, float* max_ref);
/* The same with two extras.  fp8 != 0: the e4m3 GEMM on the MX-scaled MFMA (N % 256 == 0, K % 256 == 0; seeded e4m3 codes, per-row-block exponent
 * bytes and per-column weight scales; kind: 0 -> T, 1 GELU -> e4m3 + exponent bytes (max_err then is the excess over the 2^-4 quantisation step),
 * 2 f32 residual, 5 -> f32).  reps > 0: the launch is then repeated reps times and its average duration returned in *avg_ms (may be NULL). */
int ss_engine_selftest_gemm_ex(ss_engine* e, int32_t M, int32_t N, int32_t K, int32_t kind, int32_t fp8, int32_t reps, float* max_err, float* max_ref,
                               float* avg_ms);

/* Host utility (no device needed): the OCP e4m3 code of each float, round to nearest even, saturating at +-448 -- the converter the fp8 engine
 * quantises its weights with at load (w / scale_n, scale_n = amax_n / 448), identical to gfx950's v_cvt_pk_fp8_f32. */
int ss_e4m3_from_f32(const float* x, uint8_t* codes, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
