// Host runtime of the MI355X Whisper path: model residency, batch workspaces, the batched window/decoder state
// machine (whisper_full_with_state semantics per window, many windows per device batch) and the async batch former.
#pragma once
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <random>
#include <thread>
#include <algorithm>
#include <atomic>

#include "../../include/speaksense.h"
#include "common.h"
#include "kernels.h"

namespace ss {

struct TokenData {
    int id = 0, tid = 0; float p = 0, plog = 0, pt = 0, ptsum = 0;
    int64_t t0 = -1, t1 = -1; float vlen = 0;   // whisper_token_data: token-level times (ss_params.token_timestamps; 10 ms units, -1 = not computed), voice length
};
struct Segment { int64_t t0, t1; std::string text; bool speaker_turn_next; std::vector<TokenData> tokens; };   // tokens: whisper_full_get_token_*

// std::mt19937 that counts how often it was invoked: a session's sampling history is then summarised by one number, and a fresh session
// can be put into the state an older one had (`discard`), which is what lets independent chunks of one reference state run as a batch.
struct CountingRng {
    typedef std::mt19937::result_type result_type;
    std::mt19937 g{0};
    uint64_t n = 0;
    static constexpr result_type min() { return std::mt19937::min(); }
    static constexpr result_type max() { return std::mt19937::max(); }
    result_type operator()() { n++; return g(); }
    void discard(uint64_t k) { g.discard(k); n += k; }
};

struct Session {
    struct EngineBase* eng = nullptr;
    std::vector<Segment> segments;
    std::vector<TokenData> tokens;
    std::vector<int> trace;        // every id ANY decoder sampled (failed attempts and losing best_of decoders included), in whisper_full's call order
    std::vector<int> sampled;      // every id the winning decoder of each window sampled (incl. the tail past result_len that `tokens` drops)
    int n_encode = 0, n_decode = 0, n_fail = 0, n_windows = 0;
    std::vector<int> prompt_past;  // whisper_state::prompt_past: text context carried between windows (and calls, unless no_context)
    int pool_engine = -1;          // ss_pool: engine index of the last chunk
    std::atomic<int> in_flight{0}; // chunks submitted (ss_submit) that the engine has not completed yet: they write into this session.  ss_pool_submit keeps
                                   // such a session on its engine; ss_session_free waits for 0.  Decremented by the engine under its qmu when the chunk is done
    int lang_id = -1;              // whisper_full_lang_id: language of the last chunk (given or detected)
    int64_t t_beg = 0, t_last = 0; int tid_last = 0;   // whisper_state::t_beg / t_last / tid_last: carried from segment to segment of one chunk by the token-level timestamps
    CountingRng rng;  // the generator a state carries from call to call, seeded with 0 once, never reseeded: decoder 0's (whisper.cpp >= 1.5.0,
                      // whisper_init_state) or whisper_state::rng shared by all decoders (SS_COMPAT_RNG_STATE, <= 1.4.x)
    std::vector<CountingRng> rng_dec;   // [j - 1] = generator of decoder j >= 1: re-seeded with 0 by every chunk ("TAGS: WHISPER_DECODER_INIT" of
                                        // whisper_full_with_state); unused under SS_COMPAT_RNG_STATE
};

struct Job {  // one chunk handed to transcribe (== one whisper_full_with_state call)
    Session* sess = nullptr;
    struct EngineBase* eng = nullptr;   // async submit: the engine whose queue holds this job (ss_wait waits there, whatever sess->eng says by then)
    const float* pcm = nullptr;
    int n_samples = 0;
    bool pcm_on_device = false;
    ss_params P{};
    int status = 0;
    std::string err;           // what failed (travels on the ticket: the worker's thread-local message would be lost)
    std::vector<float> owned;  // async submit keeps its own copy
    bool queued = false;       // came through submit() (counts in `load`, lives in `running` while in flight)
    std::vector<int> prompt_tokens;   // P.prompt_tokens / tokenised P.initial_prompt, captured at the API boundary (P's pointers are not kept)
    // async completion (set under the engine's qmu; read without it by ss_wait's fast path, hence atomic)
    std::atomic<bool> done{false};
};

struct EngineBase {
    HostModel hm;
    ss_engine_opts opts{};
    int compat = 0;   // SS_COMPAT_* in effect (opts.compat, overridden by env SS_COMPAT)
    std::mutex mu;  // serialises device work
    std::mutex stat_mu;   // guards last_ms / last_cnt / tot_ms / tot_cnt: a metrics reader must never wait for `mu`, which a worker holds for a whole group
    float last_ms[4] = {0, 0, 0, 0};
    long last_cnt[4] = {0, 0, 0, 0};   // last group: decoder passes, decoder rows, encoder windows
    std::atomic<int> last_lane{0};     // (lane 0 only) index of the lane whose group finished last: what ss_engine_last_timing / _counters report
    virtual ~EngineBase() {}
    virtual void run_jobs(std::vector<Job*>& jobs) = 0;  // blocking, any count (grouped by max_batch); takes this lane's `mu`
    virtual void run_jobs_locked(std::vector<Job*>& jobs) = 0;   // caller holds `mu`
    virtual void log_mel_host(const float* pcm, int n, float* out, int n_len) = 0;
    virtual void signal_energy_host(const float* pcm, int n, float* out) = 0;
    virtual void encode_host(const float* mel, int n_len, int seek, float* enc_out, int audio_ctx = 0) = 0;   // audio_ctx: 0 = n_audio_ctx
    // The stage hooks' decoder context (cross-K/V slot 0 + self-KV slot 0 of lane 0) is ONE per engine, not one per session: `owner` (the calling
    // session) takes it in set_encoder_host, any device group / other hook on lane 0 drops it, and decode_host refuses a caller that does not hold it
    // or asks for history (n_past) beyond what it has decoded since -- two whisper_states on one context can therefore never read each other's audio.
    std::atomic<const void*> hook_owner{nullptr};   // the session that holds the stage hooks' decoder context; ss_session_free drops it (a later session at the same address must not inherit it)
    virtual void set_encoder_host(const float* enc, int audio_ctx = 0, const void* owner = nullptr) = 0;   // audio_ctx: rows of `enc` (0 = n_audio_ctx); decode_host then attends over that many keys
    virtual void decode_host(const int32_t* tokens, int n, int n_past, float* logits_out, const void* owner = nullptr) = 0;
    virtual void set_encoder_window_host(const float* enc, int window) = 0;
    virtual void fp8_first_quant_host(const float* mel, int n_len, int seek, uint8_t* codes, uint8_t* exps) = 0;
    virtual void decode_rows_host(const int32_t* token, const int32_t* pos, const int32_t* slot, const int32_t* cross, int n, const int32_t* samp_rows, int n_samp,
                                  float* logits_out) = 0;
    virtual void process_logits_host(const float* raw, const int32_t* hist, int n_hist, int has_ts, int seek_delta, const ss_params& P, float out6[6],
                                     float* logprobs_out = nullptr) = 0;
    virtual void probe_gemm(int batch, int reps, float* avg_ms, double* flops) = 0;
    virtual void denoise_host(const float* pcm, int n, const ss_denoise_config& cfg, int force_type, float* out, int* noise_type, float* norm_var, float* ms) = 0;
    virtual void selftest_gemm(int M, int N, int K, int kind, float* max_err, float* max_ref) = 0;
    virtual void selftest_gemm_ex(int M, int N, int K, int kind, int fp8, int reps, float* max_err, float* max_ref, float* avg_ms) = 0;
    virtual void resample_stream_host(const float* pcm, int64_t n, int from_rate, float* out, int64_t out_cap, int64_t* n_out, int32_t* chunk_lens, float* ms) = 0;
    virtual void preprocess_stream_host(const float* pcm, int64_t n, const int32_t* chunk_lens, int n_chunks, int chunk_len, const ss_denoise_config& cfg,
                                        float* out, float* gains_out, float* ms) = 0;

    // Lanes: independent (stream, workspaces, KV caches, staging) contexts over ONE resident copy of the weights.  A decode step is a chain of
    // ~400 short dependent launches that leaves most of the chip idle, and the encoder pass of the next group is MFMA-bound while a decode is
    // latency-bound: groups running on different lanes overlap both.  lane(0) is this engine.
    virtual int n_lanes() const { return 1; }
    virtual EngineBase* lane(int) { return this; }
    void run_jobs_any(std::vector<Job*>& jobs);   // blocking: on a lane that is free right now, else round-robin
    void run_jobs_parallel(std::vector<Job*>& jobs);   // blocking: groups of max_batch spread over the lanes (one thread per lane in use)
    // cumulative device time (ms: mel, encoder+cross-KV, decode, total) and work (decoder passes, decoder rows, encoder windows) of this lane since
    // creation; ss_engine_totals sums over the lanes
    double tot_ms[4] = {0, 0, 0, 0};
    long tot_cnt[6] = {0, 0, 0, 0, 0, 0};   // ..., [4] windows started while other windows of the group were decoding, [5] decode-step graphs evicted from the LRU

    // async batch former: one worker thread per lane; one of them at a time forms the next batch from the queue
    std::vector<std::thread> workers;
    std::mutex qmu, form_mu;
    std::condition_variable qcv, donecv;
    std::atomic<int> load{0};      // chunks queued or running (ss_pool routing)
    std::deque<Job*> queue;
    std::vector<std::pair<Job*, EngineBase*>> running;     // popped, not yet done, with the lane that runs it (a session has at most one chunk in flight)
    EngineBase* owner = this;      // the engine whose queue feeds this lane (lane 0)
    bool from_queue = false;       // this lane is running a batch its worker popped: chunks complete one by one, more may be admitted between windows
    // continuous batching at window granularity (called by a lane from inside its device group, on `owner`):
    void job_finished(Job* j, EngineBase* lane_);                         // this chunk's results are final: wake its waiter now, not when the group ends
    int admit_more(int n_max, EngineBase* lane_, std::vector<Job*>& out); // pop up to n_max queued chunks (sessions not in flight) for a running group
    bool stop = false;
    int n_waiters = 0;             // threads inside wait() / wait_session_idle() (under qmu): stop_worker lets them leave before the engine goes
    void complete_locked(Job* j);  // caller holds qmu: the chunk's status is final -- bookkeeping, then `done`
    std::atomic<unsigned> rr{0};
    std::atomic<int> workers_free{0};   // workers not running a batch right now (waiting for, or forming, one): queued chunks are theirs first
    void start_worker();
    void stop_worker();
    void submit(Job* j);
    void wait(Job* j);
    // ss_session_free with chunks of the session still in flight: a counted waiter like wait(); `registry` (capi.cpp g_live_mu, held by the caller,
    // who has just seen this engine in the live set) is released as soon as the count is taken
    void wait_session_idle(Session& s, std::unique_lock<std::mutex>& registry);
};

EngineBase* make_engine_bf16(const char* path, const ss_engine_opts& o);
EngineBase* make_engine_f16(const char* path, const ss_engine_opts& o);

}  // namespace ss
