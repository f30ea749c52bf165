// EngineT<T>: device-resident model + batch workspaces + the batched whisper_full_with_state state machine.
//
// Host-side mirror of what the reference reaches through `state.full(params, &audio)` (/root/reference/src/asr/whisper.rs:75):
// whisper.cpp's whisper_full_with_state window loop (seek / temperature fallback / segmenting),
// re-organised so that many independent 30 s windows (from different sessions) advance in lock-step on one GPU:
// one encoder pass over W windows (M = W*1500 rows per GEMM), then KV-cached decoding of all their decoders as rows of
// the same GEMVs so every decoder weight is read once per pass for the whole batch.  Sampling rules run on the device;
// only {id,p,plog,tid,pt,ptsum} per row return to the host each step.  No CPU fallback: without a HIP device
// engine creation fails with SS_ERR_DEVICE.
#include "engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>

namespace ss {

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------
static inline uint16_t f32_to_bf16_bits(float x) {
    uint32_t u; memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // overflow -> inf (after rounding)
    if (x < 0x33000001u) return (uint16_t)sign;              // underflow -> 0
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7fffffu) | 0x800000u;
    if (e < -14) {  // subnormal half
        const int shift = -14 - e + 13;
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3ffu);
    const uint32_t rem = m & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
    return (uint16_t)(sign | r);
}
template <typename T> static inline uint16_t to_bits(float x);
template <> inline uint16_t to_bits<bf16>(float x) { return f32_to_bf16_bits(x); }
template <> inline uint16_t to_bits<f16>(float x) { return f32_to_f16_bits(x); }

// Every lane works on its own NON-blocking stream and never touches the legacy stream at run time: an operation on the legacy stream would
// have to synchronise with every blocking stream, which HIP refuses while another lane is capturing its decode graph.  Allocations made while
// an engine method runs are zeroed on that lane's stream (t_alloc_stream), the ones made at construction on the legacy stream + a device sync.
static thread_local hipStream_t t_alloc_stream = nullptr;
struct AllocStreamScope { hipStream_t prev; explicit AllocStreamScope(hipStream_t s) : prev(t_alloc_stream) { t_alloc_stream = s; } ~AllocStreamScope() { t_alloc_stream = prev; } };
struct DBuf {
    void* p = nullptr; size_t bytes = 0;
    void alloc(size_t n, bool zero = true) {
        free();
        if (n == 0) n = 16;
        SS_HIP(hipMalloc(&p, n)); bytes = n;
        if (zero) {
            if (t_alloc_stream) { SS_HIP(hipMemsetAsync(p, 0, n, t_alloc_stream)); SS_HIP(hipStreamSynchronize(t_alloc_stream)); }
            else { SS_HIP(hipMemset(p, 0, n)); SS_HIP(hipDeviceSynchronize()); }
        }
    }
    void ensure(size_t n) { if (n > bytes) alloc(n); }
    void free() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
    ~DBuf() { free(); }
    template <typename U> U* as() const { return (U*)p; }
};

// pinned, device-mapped host memory that only grows (per batch slot: the signal energy of the slot's chunk -- written by the kernel straight over
// PCIe, read by the token-level timestamps on the host)
struct HBuf {
    void* p = nullptr; void* dev = nullptr; size_t bytes = 0;
    void ensure(size_t n) {
        if (n <= bytes) return;
        free();
        SS_HIP(hipHostMalloc(&p, n, hipHostMallocMapped)); bytes = n;
        SS_HIP(hipHostGetDevicePointer(&dev, p, 0));
    }
    void free() { if (p) { (void)hipHostFree(p); p = nullptr; dev = nullptr; bytes = 0; } }
    ~HBuf() { free(); }
    HBuf() = default;
    HBuf(const HBuf&) = delete;
    HBuf& operator=(const HBuf&) = delete;
    HBuf(HBuf&& o) noexcept : p(o.p), dev(o.dev), bytes(o.bytes) { o.p = nullptr; o.dev = nullptr; o.bytes = 0; }
};

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ss_engine_opts.compat, overridden by env SS_COMPAT: a number, or flag names joined by '+' / ',' ("rng_state", "openai_ts_rules"; "" or "v1.5" = 0)
static int resolve_compat(int from_opts) {
    const char* e = getenv("SS_COMPAT");
    if (!e) return from_opts;
    std::string s(e);
    if (s.empty() || s == "v1.5") return 0;
    if (s.find_first_not_of("0123456789") == std::string::npos) return atoi(e);
    int flags = 0;
    size_t pos = 0;
    while (pos <= s.size()) {
        size_t end = s.find_first_of("+,", pos);
        if (end == std::string::npos) end = s.size();
        const std::string name = s.substr(pos, end - pos);
        if (name == "rng_state") flags |= SS_COMPAT_RNG_STATE;
        else if (name == "openai_ts_rules") flags |= SS_COMPAT_OPENAI_TS_RULES;
        else if (name == "openai_history") flags |= SS_COMPAT_OPENAI_HISTORY;
        else if (!name.empty()) throw Error(SS_ERR_ARG, "SS_COMPAT: unknown flag '" + name + "'");
        pos = end + 1;
    }
    return flags;
}

// per-decoder state (whisper_decoder + whisper_sequence)
struct Dec {
    std::vector<TokenData> tokens;
    int result_len = 0;
    double sum_logprobs_all = 0, sum_logprobs = -INFINITY, avg_logprobs = -INFINITY, entropy = 0, score = -INFINITY;
    int seek_delta = 0;
    bool failed = false, completed = false, has_ts = false;
    int n_fed = 0;   // tokens fed to the device so far (prompt first, then sampled tokens)
    int i = 0;       // index of the next token to sample
    bool active = false;
    int slot = 0;
    std::vector<int> sampled;   // every id this decoder sampled, before the truncation to result_len (ss_result_sampled_tokens)
};
struct Window {
    Job* job = nullptr;
    int cross = 0;   // cross-KV index in this batch
    int n_keys = 0;  // whisper_full_params.audio_ctx when it shortens the window (RowCtl.n_keys); 0 = the model's n_audio_ctx
    int it = 0;      // temperature ladder position
    bool pending = true;
    std::vector<float> temperatures;
    std::vector<int> prompt;
    std::vector<Dec> decs;
    int best = 0;
    bool skip = false;     // detect_language only: the window is not decoded
    bool done = false;     // accepted attempt, ready to be finalised
    std::vector<int> trace;   // every id any decoder of this window sampled, in whisper_full's call order: attempt, step, decoder (ss_result_trace_tokens)
};
struct JobState {
    Job* job; int slot; int n_len = 0, n_len_org = 0, seek = 0, seek_start = 0, seek_end = 0; bool alive = false;
    std::vector<int> prompt_init;
    bool need_detect = false;   // language "auto": resolved after the first window's encoder pass
    bool has_window = false;    // a window of this chunk is encoded and decoding right now
    const float* energy = nullptr; int n_energy = 0;   // token_timestamps: whisper_state::energy of this chunk (the slot's pinned host buffer, written by the device)
};

// ------------------------------------------------------------------------------------------------
// whisper_exp_compute_token_level_timestamps (whisper.cpp v1.5.4, "experimental"; run on every new segment when whisper_full_params.token_timestamps
// is set -- the reference sets it, /root/reference/src/asr/whisper.rs:160, thresholds :170-171).  Host arithmetic on ~100 tokens per window, as in
// whisper.cpp; the per-sample signal energy it consults comes from the device (kernels_mel.hip signal_energy_kernel).  Three passes over the
// segment's tokens: anchors from the timestamp distribution of each sampled token, proportional fill by "voice length", then the energy walk.
// ------------------------------------------------------------------------------------------------
static float token_voice_length(const std::string& text) {
    float v = 0.0f;
    for (const char c : text) {
        switch (c) {
            case ' ': v += 0.01f; break;
            case ',': v += 2.00f; break;
            case '.': case '!': case '?': v += 3.00f; break;
            default: v += (c >= '0' && c <= '9') ? 3.00f : 1.00f;
        }
    }
    return v;
}
static void token_level_times(Session& st, const Vocab& vocab, Segment& seg, const float* en, int n_samples, float thold_pt, float thold_ptsum) {
    std::vector<TokenData>& tk = seg.tokens;
    const int n = (int)tk.size();
    if (n_samples <= 0 || n == 0) return;     // "no signal data available"
    if (n == 1) { tk[0].t0 = seg.t0; tk[0].t1 = seg.t1; return; }
    const auto to_sample = [n_samples](int64_t t) { return std::max(0, std::min(n_samples - 1, (int)((t * kSampleRate) / 100))); };
    const auto to_time = [](int i) { return (int64_t)((100ll * i) / kSampleRate); };
    // pass 1: anchors
    if (tk[0].id == vocab.token_beg) {
        tk[0].t0 = tk[0].t1 = tk[1].t0 = seg.t0;
        st.t_beg = st.t_last = seg.t0; st.tid_last = vocab.token_beg;
    } else {
        tk[0].t0 = st.t_last;
    }
    for (int j = 0; j < n; j++) {
        TokenData& t = tk[j];
        t.vlen = token_voice_length(vocab.id_to_token[t.id]);
        const int64_t when = st.t_beg + 2 * (int64_t)(t.tid - vocab.token_beg);
        if (t.pt > thold_pt && t.ptsum > thold_ptsum && t.tid > st.tid_last && when <= seg.t1) {
            if (j > 0) tk[j - 1].t1 = when;
            t.t0 = when;
            st.tid_last = t.tid;
        }
    }
    tk[n - 2].t1 = seg.t1;
    tk[n - 1].t0 = tk[n - 1].t1 = seg.t1;
    st.t_last = seg.t1;
    // pass 2: every maximal run first .. last whose members before `last` have no end time shares [t0(first), t1(last)] by voice length
    for (int first = 0; first < n;) {
        int last = first;
        while (last < n && tk[last].t1 < 0) last++;
        if (last >= n) last = n - 1;
        if (last > first) {
            double total = 0.0;
            for (int j = first; j <= last; j++) total += tk[j].vlen;
            const double span = (double)(tk[last].t1 - tk[first].t0);
            for (int j = first; j < last; j++) {
                const double cut = tk[j].t0 + span * tk[j].vlen / total;
                tk[j].t1 = tk[j + 1].t0 = (int64_t)cut;
            }
        }
        first = last + 1;
    }
    for (int j = 0; j + 1 < n; j++) {
        if (tk[j].t1 < 0) tk[j + 1].t0 = tk[j].t1;
        if (j > 0 && tk[j - 1].t1 > tk[j].t0) { tk[j].t0 = tk[j - 1].t1; tk[j].t1 = std::max(tk[j].t0, tk[j].t1); }
    }
    // pass 3: text tokens snap to where the signal energy crosses half its mean over the token +- 1/8 s.  The means first: a token's window depends
    // only on its pass-2 times (the walk below changes token j alone, after its own mean was taken), and each mean is a left-to-right f32 sum of
    // ~4 000 + duration samples -- a chain of dependent adds, ~5 us per token, ~0.5 ms per 30 s chunk on the lane's worker thread.  Four tokens'
    // chains are interleaved so that the adder pipeline is shared; every chain keeps its own order, so the sums are whisper.cpp's bit for bit.
    const int margin = kSampleRate / 8;
    std::vector<int> wa(n), wb(n), wlo(n), whi(n), text_tok;
    std::vector<float> mean_sum(n, 0.0f);
    for (int j = 0; j < n; j++) {
        if (tk[j].id >= vocab.token_eot) continue;
        wa[j] = to_sample(tk[j].t0); wb[j] = to_sample(tk[j].t1);
        wlo[j] = std::max(wa[j] - margin, 0); whi[j] = std::min(wb[j] + margin, n_samples);
        text_tok.push_back(j);
    }
    {
        size_t g = 0;
        for (; g + 4 <= text_tok.size(); g += 4) {
            const int j0 = text_tok[g], j1 = text_tok[g + 1], j2 = text_tok[g + 2], j3 = text_tok[g + 3];
            const float *p0 = en + wlo[j0], *p1 = en + wlo[j1], *p2 = en + wlo[j2], *p3 = en + wlo[j3];
            const int l0 = std::max(0, whi[j0] - wlo[j0]), l1 = std::max(0, whi[j1] - wlo[j1]), l2 = std::max(0, whi[j2] - wlo[j2]), l3 = std::max(0, whi[j3] - wlo[j3]);
            const int common = std::min(std::min(l0, l1), std::min(l2, l3));
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
            for (int k = 0; k < common; k++) { s0 += p0[k]; s1 += p1[k]; s2 += p2[k]; s3 += p3[k]; }
            for (int k = common; k < l0; k++) s0 += p0[k];
            for (int k = common; k < l1; k++) s1 += p1[k];
            for (int k = common; k < l2; k++) s2 += p2[k];
            for (int k = common; k < l3; k++) s3 += p3[k];
            mean_sum[j0] = s0; mean_sum[j1] = s1; mean_sum[j2] = s2; mean_sum[j3] = s3;
        }
        for (; g < text_tok.size(); g++) {
            const int j = text_tok[g];
            float s = 0.0f;
            for (int k = wlo[j]; k < whi[j]; k++) s += en[k];
            mean_sum[j] = s;
        }
    }
    for (int j = 0; j < n; j++) {
        if (tk[j].id >= vocab.token_eot) continue;
        int a = wa[j];
        const int b = wb[j];
        const float level = (float)(0.5 * mean_sum[j] / (whi[j] - wlo[j]));
        int k = a;
        if (en[k] > level && j > 0) {                       // voiced at the start: the token began earlier, but not before its predecessor ended
            while (k > 0 && en[k] > level) k--;
            tk[j].t0 = to_time(k);
            if (tk[j].t0 < tk[j - 1].t1) tk[j].t0 = tk[j - 1].t1; else a = k;
        } else {                                            // silent at the start: move in to the first voiced sample
            while (en[k] < level && k < b) k++;
            a = k; tk[j].t0 = to_time(k);
        }
        k = b;
        if (en[k] > level) {                                // voiced at the end: extend, up to the next token's start
            while (k < n_samples - 1 && en[k] > level) k++;
            tk[j].t1 = to_time(k);
            // whisper.cpp guards this read of tokens[j + 1] with `j < ns - 1` where ns is the SAMPLE count of the window (always true in practice) and
            // so reads past the vector when the segment ends in a text token; the bound the read needs is used here (oracle: same decision)
            if (j + 1 < n && tk[j].t1 > tk[j + 1].t0) tk[j].t1 = tk[j + 1].t0;
        } else {                                            // silent at the end: move back to the last voiced sample
            while (en[k] < level && k > a) k--;
            tk[j].t1 = to_time(k);
        }
    }
}

static void sequence_score(const ss_params& P, Dec& q) {  // whisper_sequence_score
    if (q.result_len == 0) return;
    double result = 0.0;
    for (int i = 0; i < q.result_len; i++) result += q.tokens[i].plog;
    q.sum_logprobs = result;
    q.avg_logprobs = result / q.result_len;
    double penalty = q.result_len;
    if (P.length_penalty > 0.0f) penalty = pow((5.0 + penalty) / 6.0, P.length_penalty);
    q.score = result / penalty;
    int cnt = 0;
    double entropy = 0.0;
    std::map<int, int> tc;
    for (int i = std::max(0, q.result_len - 32); i < q.result_len; i++) { tc[q.tokens[i].id]++; cnt++; }
    for (auto& kv : tc) { const double p = kv.second / (double)cnt; entropy -= p * log(p); }
    q.entropy = entropy;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
struct EngineT : EngineBase {
    hipStream_t st = nullptr;
    int B = 8, ND = 5, S = 40;  // windows per batch, decoders per window, decoder slots
    int d, da, H, Ha, L, La, n_mel, n_ctx, n_tctx, n_vocab, n_vocab_pad, K1, Tpad;
    float qscale;
    int dtype_is_f16;

    // ---- weights ----
    DBuf w_all;  // one arena
    struct EncL { float *ln1w, *ln1b, *bqkv, *bo, *ln2w, *ln2b, *b1, *b2; T *wqkv, *wo, *w1, *w2;
                  uint8_t *wqkv8, *wo8, *w18, *w28; float *sqkv, *so, *s1, *s2; };   // fp8 engine: e4m3 codes + one f32 scale per output channel
    struct DecL { float *ln1w, *ln1b, *bqkv, *bo, *lncw, *lncb, *bcq, *bco, *ln2w, *ln2b, *b1, *b2; T *wqkv, *wo, *wcq, *wco, *w1, *w2; };
    std::vector<EncL> enc;
    std::vector<DecL> dec;
    T *conv1w, *conv2w, *tok_emb, *crosskv_w;
    uint8_t* crosskv_w8 = nullptr; float* crosskv_s = nullptr;
    bool fp8_enc = false;   // SS_DTYPE_FP8: encoder projections and the cross-KV projection in e4m3 on the MX-scaled MFMA; everything else as the f16 engine
    float *conv1b, *conv2b, *enc_pos, *lnpostw, *lnpostb, *dec_pos, *crosskv_b, *lnw, *lnb;
    MelTables mt{};

    // ---- workspaces ----
    std::vector<DBuf> pcm_d, mel_d, fmax_d;  // per batch slot
    // per batch slot, sized on first use: signal energy (token_timestamps) in pinned host memory the kernel writes directly.  A device buffer + a
    // hipMemcpyAsync back cost ~0.3 ms of HOST time per chunk in the copy call (r04_x: -1.4 % on the headline); the 1.9 MB per 30 s chunk are
    // instead stored over PCIe by the kernel itself, complete when the stream reaches the first decode step's synchronisation
    std::vector<HBuf> energy_h;
    bool energy_sized = false, mel_sized = false, pcm_sized = false;
    DBuf x0, h1, x, ln, qk, vT, att, ff, encT, encF, cross, kself, vself;
    DBuf ln8, ln_sc, att8, att_sc, ff8, ff_sc;   // fp8 engine: quantised activations + their exponent bytes
    DBuf cross_sc;                               // fp8 engine: exponent bytes of the e4m3 cross cache, [L][B][kv][h][t]
    long Mpad = 0;
    DBuf lnd, qd, attd, ffd, logits, probs, cscratch, ctl_d;
    // Host staging for one decoder launch: control blocks, sampling-row indices and the uniform draws.  H2D copies from pinned memory read
    // their source when the copy EXECUTES, and the stream may be backlogged (encoder pass, earlier launches of the same round), so every
    // launch fills its own block of a ring and a block is only rewritten after the event recorded behind its copies has completed.
    struct Stage { RowCtl ctl[2 * kPartRows]; int rowidx[kPartRows]; double u[kPartRows]; };
    static constexpr int kStageRing = 16;
    Stage* stage_h = nullptr;      // pinned ring
    hipEvent_t stage_ev[kStageRing];
    bool stage_busy[kStageRing] = {};
    int stage_cur = 0;
    RowCtl* ctl_h = nullptr;       // = stage_h[stage_cur].ctl  ([0, kPartRows) rows, [kPartRows, 2 kPartRows) sampling rows)
    int* rowidx_h = nullptr;       // = stage_h[stage_cur].rowidx
    void stage_acquire() {         // next free block of the ring becomes ctl_h / rowidx_h / u_h
        stage_cur = (stage_cur + 1) % kStageRing;
        if (stage_busy[stage_cur]) { SS_HIP(hipEventSynchronize(stage_ev[stage_cur])); stage_busy[stage_cur] = false; }
        ctl_h = stage_h[stage_cur].ctl; rowidx_h = stage_h[stage_cur].rowidx; u_h = stage_h[stage_cur].u;
    }
    void stage_release() {         // call after the last H2D copy of the launch has been enqueued
        SS_HIP(hipEventRecord(stage_ev[stage_cur], st)); stage_busy[stage_cur] = true;
    }
    SampleOut* samp_h = nullptr;   // pinned, mapped
    SampleOut* samp_hb[2] = {nullptr, nullptr};   // fused steps alternate between two result buffers (a chained step may be in flight)
    hipEvent_t ev_step[2];
    int step_parity = 0;
    double* u_h = nullptr;         // = stage_h[stage_cur].u: one uniform draw per sampled row (t > 0)
    DBuf u_d;
    hipEvent_t ev[4];
    std::vector<hipEvent_t> enc_event_pool;   // timing events of the encoder sections (run_group), reused across groups
    int lane_index = 0;

    std::vector<std::unique_ptr<EngineT>> extra_lanes;   // lanes 1.. (lane 0 = this); they borrow this engine's weight arena
    int n_lanes() const override { return 1 + (int)extra_lanes.size(); }
    EngineBase* lane(int i) override { return i == 0 ? (EngineBase*)this : (EngineBase*)extra_lanes[i - 1].get(); }

    // donor != nullptr: a lane of `donor` -- same device, same weights (pointers into the donor's arena), own stream / workspaces / caches
    // lane_idx / n_lanes_total: this lane's place in the engine (decides its CU partition, if any); lane 0 works them out itself
    EngineT(const char* path, const ss_engine_opts& o, EngineT* donor = nullptr, int lane_idx = 0, int n_lanes_total = 0) {
        opts = o;
        lane_index = lane_idx;
        if (!donor) {
            n_lanes_total = o.n_lanes > 0 ? o.n_lanes : 2;
            if (const char* lv = getenv("SS_LANES")) n_lanes_total = atoi(lv);
            n_lanes_total = std::min(std::max(n_lanes_total, 1), 8);
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) throw Error(SS_ERR_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
        if (o.device < 0 || o.device >= ndev) throw Error(SS_ERR_DEVICE, "bad device ordinal");
        SS_HIP(hipSetDevice(o.device));
        if (donor) { hm.hp = donor->hm.hp; hm.vocab = donor->hm.vocab; hm.filt_n_mel = donor->hm.filt_n_mel; hm.filt_n_fft = donor->hm.filt_n_fft; }
        else load_ggml_model(path, hm);
        const HParams& hp = hm.hp;
        B = o.max_batch > 0 ? o.max_batch : 8;
        ND = o.max_decoders > 0 ? o.max_decoders : 5;
        S = B * ND;
        d = hp.n_text_state; da = hp.n_audio_state; H = hp.n_text_head; Ha = hp.n_audio_head; L = hp.n_text_layer; La = hp.n_audio_layer;
        n_mel = hp.n_mels; n_ctx = hp.n_audio_ctx; n_tctx = hp.n_text_ctx; n_vocab = hp.n_vocab; n_vocab_pad = round_up(n_vocab, 64);
        K1 = round_up(3 * n_mel, 64);
        Tpad = round_up(n_ctx, 64);   // V^T rows padded (zero) to whole 64-key chunks for the LDS-staged attention kernel
        nc = n_ctx; Tp = Tpad;
        qscale = powf(64.0f, -0.25f);
        dtype_is_f16 = sizeof(T) == 2 && std::is_same<T, f16>::value;
        fp8_enc = o.dtype == SS_DTYPE_FP8;
        if (fp8_enc && (da % 256 || d % 64)) throw Error(SS_ERR_UNSUPPORTED, "fp8: n_audio_state must be a multiple of 256 (k-step groups of the e4m3 GEMM)");
        if (d % 128 || da % 128) throw Error(SS_ERR_MODEL, "model: state size must be a multiple of 128");
        if (n_ctx % 4 || n_tctx > 448) throw Error(SS_ERR_MODEL, "model: unsupported context sizes");
        if (const char* vg = getenv("SS_VT_GEMM")) vt_gemm = atoi(vg) != 0;   // before the workspaces are sized: only the =0 form needs the Q | K | V buffer
        if (!donor) check_memory_fits(n_lanes_total);
        SS_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        AllocStreamScope alloc_scope(st);
        for (auto& e : ev) SS_HIP(hipEventCreate(&e));
        for (auto& e : ev_step) SS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        if (donor) {
            enc = donor->enc; dec = donor->dec;
            conv1w = donor->conv1w; conv2w = donor->conv2w; tok_emb = donor->tok_emb; crosskv_w = donor->crosskv_w;
            crosskv_w8 = donor->crosskv_w8; crosskv_s = donor->crosskv_s;
            conv1b = donor->conv1b; conv2b = donor->conv2b; enc_pos = donor->enc_pos; lnpostw = donor->lnpostw; lnpostb = donor->lnpostb;
            dec_pos = donor->dec_pos; crosskv_b = donor->crosskv_b; lnw = donor->lnw; lnb = donor->lnb; mt = donor->mt;
        } else {
            upload_weights();
        }
        alloc_workspaces();
        plan_decode();
        // two switches that change HOW the step is issued, not which kernels run (both under test: tests/test_gpu_variants.py::test_decode_issue_modes)
        { const char* gv = getenv("SS_DECODE_GRAPH"); use_graph = !(gv && gv[0] == '0'); }     // 0: launch the step kernel by kernel instead of replaying its hipGraph
        { const char* cv = getenv("SS_DECODE_CHAIN"); chain_steps = !(cv && cv[0] == '0'); }   // 0: wait for every step's samples before enqueuing the next step
        if (const char* sm = getenv("SS_CB_START_MIN")) cb_start_min = std::max(1, atoi(sm));
        { const char* lf = getenv("SS_LN_FUSE"); ln_fuse = !(lf && lf[0] == '0'); }
        if (const char* lr = getenv("SS_LN_FUSE_ROWS")) ln_fuse_rows = std::min(16, std::max(1, atoi(lr)));
        compat = donor ? donor->compat : resolve_compat(o.compat);
        if (!donor) {
            for (int i = 1; i < n_lanes_total; i++) { extra_lanes.emplace_back(new EngineT(path, o, this, i, n_lanes_total)); extra_lanes.back()->owner = this; }
            start_worker();
        }
    }
    ~EngineT() override {
        stop_worker();          // joins the workers of every lane (they live in lane 0); a lane itself has none
        extra_lanes.clear();    // before the weight arena goes
        if (st) (void)hipStreamSynchronize(st);   // a chained decode step may still be in flight: it writes into the pinned buffers freed below
        if (stage_h) (void)hipHostFree(stage_h);
        for (auto& e : stage_ev) (void)hipEventDestroy(e);
        reap_retired_graphs();
        for (auto& kv : step_graphs) { if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec); if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph); }
        if (samp_h) (void)hipHostFree(samp_h);
        for (auto& e : ev) (void)hipEventDestroy(e);
        for (auto& e : enc_event_pool) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
    }

    // ------------------------------------------------------------------------------------------
    // weights: one arena; 2-D weights converted to T ([N][K], K contiguous), the rest f32
    // ------------------------------------------------------------------------------------------
    struct Arena { std::vector<uint8_t> host; size_t off = 0;
        size_t take(size_t bytes) { size_t o = (off + 255) & ~(size_t)255; off = o + bytes; if (host.size() < off) host.resize(off); return o; } };
    Arena ar;
    std::vector<std::pair<void**, size_t>> fix;
    template <typename P> void reg(P*& ptr, size_t off) { fix.push_back({(void**)&ptr, off}); }

    void put_f32(float*& ptr, const std::vector<float>& v) {
        size_t o = ar.take(v.size() * 4); memcpy(ar.host.data() + o, v.data(), v.size() * 4); reg(ptr, o);
    }
    // a decode-step GEMV weight [N][K] in the fragment-major layout the GEMV kernels read (kernels.h dec_wpack_off); rows N .. N_alloc - 1 are zero
    void put_T_packed(T*& ptr, const float* v, size_t N, size_t K, size_t N_alloc = 0) {
        if (N_alloc < N) N_alloc = N;
        if (N_alloc % 16 || K % 32) throw Error(SS_ERR_MODEL, "model: decoder weight shape not a multiple of 16 x 32");
        size_t o = ar.take(N_alloc * K * 2);
        uint16_t* dst = (uint16_t*)(ar.host.data() + o);
        memset(dst, 0, N_alloc * K * 2);
        for (size_t n = 0; n < N; n++)
            for (size_t k = 0; k < K; k += 8) {
                uint16_t* q = dst + dec_wpack_off((long)n, (int)k, (int)K);
                for (int e = 0; e < 8; e++) q[e] = to_bits<T>(v[n * K + k + e]);
            }
        reg(ptr, o);
    }
    void put_T(T*& ptr, const float* v, size_t n, size_t n_alloc = 0) {
        if (n_alloc < n) n_alloc = n;
        size_t o = ar.take(n_alloc * 2);
        uint16_t* dst = (uint16_t*)(ar.host.data() + o);
        for (size_t i = 0; i < n; i++) dst[i] = to_bits<T>(v[i]);
        for (size_t i = n; i < n_alloc; i++) dst[i] = 0;
        reg(ptr, o);
    }
    // [N][K] weight -> e4m3 codes + per-output-channel scale (amax_n / 448): w ~ code * scale
    void put_f8(uint8_t*& ptr, float*& sc, const float* v, size_t N, size_t K) {
        size_t o = ar.take(N * K);
        std::vector<float> scales(N);
        for (size_t n = 0; n < N; n++) {
            float amax = 0.f;
            for (size_t k = 0; k < K; k++) amax = std::max(amax, fabsf(v[n * K + k]));
            scales[n] = amax > 0.f ? amax / 448.0f : 1.0f;
        }
        uint8_t* dst = ar.host.data() + o;
        for (size_t n = 0; n < N; n++) {
            const float sn = scales[n];
            for (size_t k = 0; k < K; k++) dst[n * K + k] = f32_to_e4m3(v[n * K + k] / sn);
        }
        reg(ptr, o);
        put_f32(sc, scales);
    }
    // a projection weight: T in the f16 / bf16 engines, e4m3 + scales in the fp8 engine (which then carries no T copy of it)
    void put_proj(T*& ptrT, uint8_t*& ptr8, float*& sc, const float* v, size_t N, size_t K) {
        if (fp8_enc) { ptrT = nullptr; put_f8(ptr8, sc, v, N, K); }
        else { ptr8 = nullptr; sc = nullptr; put_T(ptrT, v, N * K); }
    }
    const std::vector<float>& W(const std::string& n) { return hm.get(n).f32; }
    void expect(const std::string& n, size_t cnt) { if (hm.get(n).f32.size() != cnt) throw Error(SS_ERR_MODEL, "model: bad shape for " + n); }

    // conv weight [co][ci][3] -> [co][Kpad] with k-major columns: col = k*ci + c
    std::vector<float> conv_reorder(const std::vector<float>& w, int co, int ci, int Kpad) {
        std::vector<float> r((size_t)co * Kpad, 0.0f);
        for (int a = 0; a < co; a++) for (int c = 0; c < ci; c++) for (int k = 0; k < 3; k++)
            r[(size_t)a * Kpad + k * ci + c] = w[((size_t)a * ci + c) * 3 + k];
        return r;
    }
    std::vector<float> cat(std::initializer_list<const std::vector<float>*> parts) {
        std::vector<float> r;
        for (auto p : parts) r.insert(r.end(), p->begin(), p->end());
        return r;
    }

    void upload_weights() {
        const std::vector<float> zeros_d(d, 0.0f), zeros_da(da, 0.0f);
        expect("encoder.conv1.weight", (size_t)da * n_mel * 3);
        expect("encoder.conv2.weight", (size_t)da * da * 3);
        expect("encoder.positional_embedding", (size_t)n_ctx * da);
        expect("decoder.token_embedding.weight", (size_t)n_vocab * d);
        expect("decoder.positional_embedding", (size_t)n_tctx * d);
        auto c1 = conv_reorder(W("encoder.conv1.weight"), da, n_mel, K1);
        put_T(conv1w, c1.data(), c1.size());
        put_f32(conv1b, W("encoder.conv1.bias"));
        auto c2 = conv_reorder(W("encoder.conv2.weight"), da, da, 3 * da);
        put_T(conv2w, c2.data(), c2.size());
        put_f32(conv2b, W("encoder.conv2.bias"));
        put_f32(enc_pos, W("encoder.positional_embedding"));
        enc.resize(La);
        for (int i = 0; i < La; i++) {
            const std::string p = "encoder.blocks." + std::to_string(i) + ".";
            EncL& e = enc[i];
            put_f32(e.ln1w, W(p + "attn_ln.weight")); put_f32(e.ln1b, W(p + "attn_ln.bias"));
            auto wqkv = cat({&W(p + "attn.query.weight"), &W(p + "attn.key.weight"), &W(p + "attn.value.weight")});
            if (wqkv.size() != (size_t)3 * da * da) throw Error(SS_ERR_MODEL, "model: bad attention weight shape");
            put_proj(e.wqkv, e.wqkv8, e.sqkv, wqkv.data(), (size_t)3 * da, da);
            put_f32(e.bqkv, cat({&W(p + "attn.query.bias"), &zeros_da, &W(p + "attn.value.bias")}));
            put_proj(e.wo, e.wo8, e.so, W(p + "attn.out.weight").data(), da, da); put_f32(e.bo, W(p + "attn.out.bias"));
            put_f32(e.ln2w, W(p + "mlp_ln.weight")); put_f32(e.ln2b, W(p + "mlp_ln.bias"));
            if (W(p + "mlp.0.weight").size() != (size_t)4 * da * da || W(p + "mlp.2.weight").size() != (size_t)4 * da * da) throw Error(SS_ERR_MODEL, "model: bad mlp weight shape");
            put_proj(e.w1, e.w18, e.s1, W(p + "mlp.0.weight").data(), (size_t)4 * da, da); put_f32(e.b1, W(p + "mlp.0.bias"));
            put_proj(e.w2, e.w28, e.s2, W(p + "mlp.2.weight").data(), da, (size_t)4 * da); put_f32(e.b2, W(p + "mlp.2.bias"));
        }
        put_f32(lnpostw, W("encoder.ln_post.weight")); put_f32(lnpostb, W("encoder.ln_post.bias"));
        put_T_packed(tok_emb, W("decoder.token_embedding.weight").data(), (size_t)n_vocab, (size_t)d, (size_t)n_vocab_pad);
        put_f32(dec_pos, W("decoder.positional_embedding"));
        dec.resize(L);
        std::vector<float> ckw, ckb;
        for (int i = 0; i < L; i++) {
            const std::string p = "decoder.blocks." + std::to_string(i) + ".";
            DecL& e = dec[i];
            put_f32(e.ln1w, W(p + "attn_ln.weight")); put_f32(e.ln1b, W(p + "attn_ln.bias"));
            auto wqkv = cat({&W(p + "attn.query.weight"), &W(p + "attn.key.weight"), &W(p + "attn.value.weight")});
            put_T_packed(e.wqkv, wqkv.data(), (size_t)3 * d, (size_t)d);
            put_f32(e.bqkv, cat({&W(p + "attn.query.bias"), &zeros_d, &W(p + "attn.value.bias")}));
            put_T_packed(e.wo, W(p + "attn.out.weight").data(), (size_t)d, (size_t)d); put_f32(e.bo, W(p + "attn.out.bias"));
            put_f32(e.lncw, W(p + "cross_attn_ln.weight")); put_f32(e.lncb, W(p + "cross_attn_ln.bias"));
            put_T_packed(e.wcq, W(p + "cross_attn.query.weight").data(), (size_t)d, (size_t)d); put_f32(e.bcq, W(p + "cross_attn.query.bias"));
            put_T_packed(e.wco, W(p + "cross_attn.out.weight").data(), (size_t)d, (size_t)d); put_f32(e.bco, W(p + "cross_attn.out.bias"));
            put_f32(e.ln2w, W(p + "mlp_ln.weight")); put_f32(e.ln2b, W(p + "mlp_ln.bias"));
            put_T_packed(e.w1, W(p + "mlp.0.weight").data(), (size_t)4 * d, (size_t)d); put_f32(e.b1, W(p + "mlp.0.bias"));
            put_T_packed(e.w2, W(p + "mlp.2.weight").data(), (size_t)d, (size_t)4 * d); put_f32(e.b2, W(p + "mlp.2.bias"));
            const auto& kw = W(p + "cross_attn.key.weight"); const auto& vw = W(p + "cross_attn.value.weight");
            ckw.insert(ckw.end(), kw.begin(), kw.end()); ckw.insert(ckw.end(), vw.begin(), vw.end());
            ckb.insert(ckb.end(), zeros_d.begin(), zeros_d.end());
            const auto& vb = W(p + "cross_attn.value.bias"); ckb.insert(ckb.end(), vb.begin(), vb.end());
        }
        if (ckw.size() != (size_t)L * 2 * d * da) throw Error(SS_ERR_MODEL, "model: bad cross-attention key/value weight shape");
        put_proj(crosskv_w, crosskv_w8, crosskv_s, ckw.data(), (size_t)L * 2 * d, da); put_f32(crosskv_b, ckb);
        put_f32(lnw, W("decoder.ln.weight")); put_f32(lnb, W("decoder.ln.bias"));
        // mel tables: same libm calls as whisper.cpp's fill_sin_cos_table / hann_window (periodic)
        std::vector<float> sinv(400), cosv(400), hann(400);
        for (int i = 0; i < 400; i++) {
            const double theta = (2 * M_PI * i) / 400;
            sinv[i] = sinf(theta); cosv[i] = cosf(theta);
            hann[i] = 0.5 * (1.0 - cosf((2.0 * M_PI * i) / 400));
        }
        float *dsin, *dcos, *dhann, *dfilt;
        put_f32(dsin, sinv); put_f32(dcos, cosv); put_f32(dhann, hann); put_f32(dfilt, hm.filters);
        w_all.alloc(ar.off + 256, false);
        SS_HIP(hipMemcpyAsync(w_all.p, ar.host.data(), ar.off, hipMemcpyHostToDevice, st));
        SS_HIP(hipStreamSynchronize(st));
        for (auto& f : fix) *f.first = (uint8_t*)w_all.p + f.second;
        mt.sin_t = dsin; mt.cos_t = dcos; mt.hann = dhann; mt.filt = dfilt; mt.n_mel = n_mel;
        std::vector<uint8_t>().swap(ar.host);
        for (auto& kv : hm.t) std::vector<float>().swap(kv.second.f32);  // host copies no longer needed
    }

    // Device memory the engine is about to allocate (speaksense.h documents the formula next to max_batch), against what the device has free: a
    // configuration that cannot fit fails here with the numbers, not as a hipMalloc error somewhere inside the fourth lane's workspaces.
    size_t lane_bytes() const {
        const size_t M = (size_t)B * n_ctx, R = kPartRows;
        size_t enc_ws = ((size_t)B * (2 * n_ctx + 2) * (n_mel + da)) * 2 + M * da * (4 + 2 + (vt_gemm ? 4 : 6) + 2 + 2 + 2 + 4) + (size_t)B * Ha * 64 * Tpad * 2 + M * 4 * da * (fp8_enc ? 1 : 2);
        if (fp8_enc) enc_ws += 2 * M * da + (M + 255) * (6 * da / 64);
        const size_t cross_b = (size_t)L * B * 2 * H * n_ctx * (fp8_enc ? 65 : 128);
        const size_t self_b = (size_t)2 * L * S * n_tctx * d * 2;
        const size_t rows_b = R * ((size_t)2 * n_vocab_pad * 4 + (size_t)(3 + 4) * d * 2 + (size_t)H * 4 * 66 * 4 + 64 * 8 * 4) + (size_t)(2 + 4 * 4) * R * d * 4;
        const size_t pcm_mel = (size_t)B * (2 * 480000 * 4 + (size_t)n_mel * 6000 * 4);       // grown lazily per chunk length: 30 s chunks assumed
        return enc_ws + cross_b + self_b + rows_b + pcm_mel;
    }
    void check_memory_fits(int n_lanes_total) {
        size_t free_b = 0, total_b = 0;
        // The estimate is a courtesy (a readable refusal instead of an out-of-memory error half-way through the allocations), not a gate a
        // deployment can get stuck behind: SS_SKIP_MEM_CHECK=1 turns it off, and it leaves 2 % slack for what it cannot see (allocator rounding).
        if (const char* sk = getenv("SS_SKIP_MEM_CHECK")) if (sk[0] == '1') return;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
        // test hook (tests/test_gpu_lifetime.py), honoured only when SS_TEST_HOOKS=1 is set beside it: a production engine never reads it by accident
        if (const char* th = getenv("SS_TEST_HOOKS")) if (th[0] == '1') if (const char* t = getenv("SS_TEST_FREE_MEM_MIB")) free_b = (size_t)atol(t) << 20;
        const HParams& hp = hm.hp;
        const size_t enc_w = (size_t)La * 12 * da * da + (size_t)da * 3 * n_mel + (size_t)3 * da * da;
        const size_t dec_w = (size_t)L * 16 * d * d + (size_t)hp.n_vocab * d;
        const size_t weights = (enc_w + dec_w) * 2 + (fp8_enc ? (size_t)La * 12 * da * da + (size_t)L * 2 * d * d : 0);
        const size_t need = weights + (size_t)n_lanes_total * lane_bytes();
        if (need > free_b + free_b / 50)
            throw Error(SS_ERR_ARG, "ss_engine_create: max_batch " + std::to_string(B) + " x max_decoders " + std::to_string(ND) + " on " + std::to_string(n_lanes_total) +
                        " lanes needs ~" + std::to_string(need >> 20) + " MiB of device memory (" + std::to_string(lane_bytes() >> 20) + " MiB per lane + " +
                        std::to_string(weights >> 20) + " MiB of weights), the device has " + std::to_string(free_b >> 20) + " MiB free");
    }
    void alloc_workspaces() {
        pcm_d.resize(B); mel_d.resize(B); fmax_d.resize(B); energy_h.resize(B);
        const size_t M = (size_t)B * n_ctx;
        x0.alloc(((size_t)B * (2 * n_ctx + 2) * n_mel + 256) * 2);
        h1.alloc(((size_t)B * (2 * n_ctx + 2) * da + 256) * 2);
        x.alloc(M * da * 4); ln.alloc(M * da * 2); qk.alloc(M * (vt_gemm ? 2 : 3) * da * 2);     // Q | K rows [M][2 da]; Q | K | V [M][3 da] only in the SS_VT_GEMM=0 form
        vT.alloc((size_t)B * Ha * 64 * Tpad * 2);
        att.alloc(M * da * 2); encT.alloc(M * da * 2); encF.alloc(M * da * 4);
        if (fp8_enc) {   // the quantised activations replace the T copy of the MLP hidden state; exponent bytes: one per (row, 64 columns), rows padded to 256
            Mpad = (long)((M + 255) & ~(size_t)255);
            ln8.alloc(M * da); ln_sc.alloc((size_t)Mpad * (da / 64)); att8.alloc(M * da); att_sc.alloc((size_t)Mpad * (da / 64));
            ff8.alloc(M * 4 * da); ff_sc.alloc((size_t)Mpad * (4 * da / 64));
            SS_HIP(hipMemsetAsync(ln_sc.p, 127, (size_t)Mpad * (da / 64), t_alloc_stream)); SS_HIP(hipMemsetAsync(att_sc.p, 127, (size_t)Mpad * (da / 64), t_alloc_stream));
            SS_HIP(hipMemsetAsync(ff_sc.p, 127, (size_t)Mpad * (4 * da / 64), t_alloc_stream));
        } else {
            ff.alloc(M * 4 * da * 2);
        }
        cross.alloc((size_t)L * B * 2 * H * n_ctx * 64 * (fp8_enc ? 1 : 2));   // fp8 engine: e4m3 codes + one exponent byte per (key row, head)
        if (fp8_enc) cross_sc.alloc((size_t)L * B * 2 * H * n_ctx);
        kself.alloc((size_t)L * S * n_tctx * d * 2); vself.alloc((size_t)L * S * n_tctx * d * 2);
        const int R = kPartRows;  // rows per decode launch
        lnd.alloc((size_t)R * d * 2); qd.alloc((size_t)R * d * 2); attd.alloc((size_t)R * d * 2);
        ffd.alloc((size_t)R * 4 * d * 2); logits.alloc((size_t)R * n_vocab_pad * 4); probs.alloc((size_t)R * n_vocab_pad * 4);
        cscratch.alloc((size_t)R * H * 4 * 66 * 4); ctl_d.alloc(2 * R * sizeof(RowCtl));
        samp_d.alloc(R * sizeof(SampleOut)); rowidx_d.alloc(R * 4); rules_scratch.alloc((size_t)R * 64 * 8 * 4);
        {   // ss_params.suppress_non_speech_tokens: which ids of THIS vocabulary are on whisper.cpp's non-speech list, one bit per id
            std::vector<uint32_t> m((n_vocab + 31) / 32, 0u);
            for (int id : non_speech_token_ids(hm.vocab)) if (id < n_vocab) m[id >> 5] |= 1u << (id & 31);
            ns_mask_d.alloc(m.size() * 4);
            SS_HIP(hipMemcpy(ns_mask_d.p, m.data(), m.size() * 4, hipMemcpyHostToDevice));
        }
        SS_HIP(hipHostMalloc((void**)&stage_h, kStageRing * sizeof(Stage), hipHostMallocDefault));
        memset(stage_h, 0, kStageRing * sizeof(Stage));
        for (auto& e : stage_ev) SS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        stage_acquire();
        SS_HIP(hipHostMalloc((void**)&samp_h, 2 * R * sizeof(SampleOut), hipHostMallocDefault));
        samp_hb[0] = samp_h; samp_hb[1] = samp_h + R;
        u_d.alloc((size_t)R * sizeof(double));
    }

    // ------------------------------------------------------------------------------------------
    // device passes
    // ------------------------------------------------------------------------------------------
    GemmDesc gd(const void* A, long lda, const void* Wt, int M, int N, int K, int kind, const float* bias, void* out, long ldo) {
        GemmDesc g{};
        g.A = A; g.lda = lda; g.a_rows_per_batch = 0; g.a_batch_stride = 0;
        g.W = Wt; g.M = M; g.N = N; g.K = K; g.kind = kind; g.bias = bias; g.out = out; g.ldo = ldo;
        g.o_rows_per_batch = 0; g.o_batch_stride = 0; g.scale = 1.0f; g.rows_per_batch = nc; g.d = da; g.Tpad = Tp; g.cache_rows = n_ctx;
        g.n_batch = B; g.gelu_f16_in = dtype_is_f16;
        return g;
    }
    // whisper_full_params.audio_ctx (whisper.cpp exp_n_audio_ctx): the encoder runs over the first `nc` positions of the window -- 2 nc mel frames, nc
    // rows of the positional embedding, nc keys per cross-attention head (RowCtl.n_keys) -- in buffers laid out for nc; the cross-KV CACHE keeps the
    // model's geometry (n_ctx key rows per slot and head), a shortened window fills the first nc of them.  A change of context moves the zero padding
    // rows of the conv stem and the zero key columns of V^T, so both buffers are cleared.
    int nc = 0, Tp = 0;      // context of the encoder pass being issued (set_context); n_ctx / Tpad unless a job shortens it
    void set_context(int a) {
        if (a == nc) return;
        nc = a; Tp = round_up(a, 64);
        SS_HIP(hipMemsetAsync(h1.p, 0, h1.bytes, st));
        SS_HIP(hipMemsetAsync(vT.p, 0, vT.bytes, st));
    }

    // encoder over Wn windows whose time-major inputs are already in x0; leaves ln_post output in encT (and encF if want_f32)
    void encoder_pass(int Wn, bool want_f32) {
        const int T2 = 2 * nc, M = Wn * nc;
        {   // conv1 + GELU: implicit GEMM over overlapping rows of the padded time-major input
            GemmDesc g = gd(x0.p, n_mel, conv1w, Wn * T2, da, K1, EPI_GELU_T, conv1b, h1.as<T>() + da, da);
            g.a_rows_per_batch = T2; g.a_batch_stride = (long)(T2 + 2) * n_mel;
            g.o_rows_per_batch = T2; g.o_batch_stride = (long)(T2 + 2) * da;
            launch_gemm<T>(g, st);
        }
        {   // conv2 (stride 2) + GELU + positional embedding -> f32 residual stream
            GemmDesc g = gd(h1.p, 2 * da, conv2w, M, da, 3 * da, EPI_GELU_POS_F32, conv2b, x.p, da);
            g.a_rows_per_batch = nc; g.a_batch_stride = (long)(T2 + 2) * da;
            g.pos = enc_pos;
            launch_gemm<T>(g, st);
        }
        if (fp8_enc) { encoder_layers_f8(Wn, want_f32); return; }
        for (int il = 0; il < La; il++) {
            const EncL& e = enc[il];
            launch_layernorm<T>(x.as<float>(), e.ln1w, e.ln1b, ln.as<T>(), M, da, nullptr, st);
            const int ldq = vt_gemm ? 2 * da : 3 * da;
            if (vt_gemm) {   // default: the V projection as its own GEMM with the transposing epilogue
                launch_gemm<T>(gd(ln.p, da, e.wqkv, M, 2 * da, da, EPI_STORE_T, e.bqkv, qk.p, 2 * da), st);
                GemmDesc g = gd(ln.p, da, e.wqkv + (size_t)2 * da * da, M, da, da, EPI_VT, e.bqkv + 2 * da, vT.p, 0);
                launch_gemm<T>(g, st);
            } else {         // SS_VT_GEMM=0 (measured slower): one GEMM for Q | K | V as plain rows, then V -> V^T through LDS tiles; the same bits
                launch_gemm<T>(gd(ln.p, da, e.wqkv, M, 3 * da, da, EPI_STORE_T, e.bqkv, qk.p, 3 * da), st);
                launch_v_transpose<T>(qk.as<T>() + 2 * da, 3 * da, vT.as<T>(), Tp, Wn, Ha, nc, st);
            }
            launch_enc_attention<T>(qk.as<T>(), qk.as<T>() + da, ldq, vT.as<T>(), Tp, att.as<T>(), da, Wn, Ha, nc, st);
            {
                GemmDesc g = gd(att.p, da, e.wo, M, da, da, EPI_RES_F32, e.bo, x.p, da);
                g.res = x.as<float>();
                launch_gemm<T>(g, st);
            }
            launch_layernorm<T>(x.as<float>(), e.ln2w, e.ln2b, ln.as<T>(), M, da, nullptr, st);
            launch_gemm<T>(gd(ln.p, da, e.w1, M, 4 * da, da, EPI_GELU_T, e.b1, ff.p, 4 * da), st);
            {
                GemmDesc g = gd(ff.p, 4 * da, e.w2, M, da, 4 * da, EPI_RES_F32, e.b2, x.p, da);
                g.res = x.as<float>();
                launch_gemm<T>(g, st);
            }
        }
        launch_layernorm<T>(x.as<float>(), lnpostw, lnpostb, encT.as<T>(), M, da, nullptr, st);
        if (want_f32) launch_layernorm_f32out<T>(x.as<float>(), lnpostw, lnpostb, encF.as<float>(), M, da, st);
    }
    // fp8 engine: every projection of the encoder blocks on the e4m3 GEMM.  LayerNorm writes e4m3 + exponent bytes directly, FC1's epilogue
    // quantises its GELU output per (row, 64 columns), the attention output (T) is quantised by one small pass; attention itself, the f32
    // residual stream and the conv stem are those of the f16 engine.  ln_post leaves its e4m3 output in ln8 / ln_sc for cross_kv_pass.
    GemmF8Desc gd8(const void* A8, const void* Asc, long lda, const uint8_t* W8, const float* Ws, int M, int N, int K, int kind, const float* bias, void* out, long ldo) {
        GemmF8Desc g{};
        g.A = (const unsigned char*)A8; g.lda = lda; g.a_scale = (const unsigned char*)Asc; g.ldsc = Mpad; g.W = W8; g.w_scale = Ws;
        g.M = M; g.N = N; g.K = K; g.kind = kind; g.bias = bias; g.out = out; g.ldo = ldo; g.scale = 1.0f;
        g.rows_per_batch = nc; g.d = da; g.Tpad = Tp; g.n_batch = B; g.gelu_f16_in = dtype_is_f16; g.cache_rows = n_ctx;
        return g;
    }
    uint8_t *tap8_codes = nullptr, *tap8_sc = nullptr;   // host buffers of the running fp8_first_quant_host call
    void encoder_layers_f8(int Wn, bool want_f32) {
        const int M = Wn * nc;
        for (int il = 0; il < La; il++) {
            const EncL& e = enc[il];
            launch_layernorm_f8(x.as<float>(), e.ln1w, e.ln1b, ln8.as<unsigned char>(), ln_sc.as<unsigned char>(), Mpad, M, da, st);
            if (il == 0 && tap8_codes) {   // test hook fp8_first_quant_host: window 0's codes and exponent bytes at the first quantisation point
                SS_HIP(hipMemcpyAsync(tap8_codes, ln8.p, (size_t)n_ctx * da, hipMemcpyDeviceToHost, st));
                SS_HIP(hipMemcpyAsync(tap8_sc, ln_sc.p, (size_t)(da / 64) * Mpad, hipMemcpyDeviceToHost, st));
            }
            const int ldq = vt_gemm ? 2 * da : 3 * da;
            if (vt_gemm) {
                launch_gemm_f8<T>(gd8(ln8.p, ln_sc.p, da, e.wqkv8, e.sqkv, M, 2 * da, da, F8_STORE_T, e.bqkv, qk.p, 2 * da), st);
                launch_gemm_f8<T>(gd8(ln8.p, ln_sc.p, da, e.wqkv8 + (size_t)2 * da * da, e.sqkv + 2 * da, M, da, da, F8_VT, e.bqkv + 2 * da, vT.p, 0), st);
            } else {
                launch_gemm_f8<T>(gd8(ln8.p, ln_sc.p, da, e.wqkv8, e.sqkv, M, 3 * da, da, F8_STORE_T, e.bqkv, qk.p, 3 * da), st);
                launch_v_transpose<T>(qk.as<T>() + 2 * da, 3 * da, vT.as<T>(), Tp, Wn, Ha, nc, st);
            }
            launch_enc_attention_f8<T>(qk.as<T>(), qk.as<T>() + da, ldq, vT.as<T>(), Tp, att8.as<unsigned char>(), da, att_sc.as<unsigned char>(), Mpad, Wn, Ha, nc, st);
            {
                GemmF8Desc g = gd8(att8.p, att_sc.p, da, e.wo8, e.so, M, da, da, F8_RES_F32, e.bo, x.p, da);
                g.res = x.as<float>();
                launch_gemm_f8<T>(g, st);
            }
            launch_layernorm_f8(x.as<float>(), e.ln2w, e.ln2b, ln8.as<unsigned char>(), ln_sc.as<unsigned char>(), Mpad, M, da, st);
            {
                GemmF8Desc g = gd8(ln8.p, ln_sc.p, da, e.w18, e.s1, M, 4 * da, da, F8_GELU_F8, e.b1, ff8.p, 4 * da);
                g.out_scale = ff_sc.as<unsigned char>(); g.ld_osc = Mpad;
                launch_gemm_f8<T>(g, st);
            }
            {
                GemmF8Desc g = gd8(ff8.p, ff_sc.p, 4 * da, e.w28, e.s2, M, da, 4 * da, F8_RES_F32, e.b2, x.p, da);
                g.res = x.as<float>();
                launch_gemm_f8<T>(g, st);
            }
        }
        launch_layernorm_f8(x.as<float>(), lnpostw, lnpostb, ln8.as<unsigned char>(), ln_sc.as<unsigned char>(), Mpad, M, da, st);
        if (want_f32) launch_layernorm_f32out<T>(x.as<float>(), lnpostw, lnpostb, encF.as<float>(), M, da, st);
    }
    // cross K/V of every decoder layer for Wn windows: one GEMM, N = L*2d, written straight into the cache layout
    // cmap (optional): window k of this pass writes the cross-KV cache slot cmap[k] (windows of a running group keep their slots)
    void cross_kv_pass(int Wn, const int* cmap = nullptr) {
        if (fp8_enc) {
            GemmF8Desc g = gd8(ln8.p, ln_sc.p, da, crosskv_w8, crosskv_s, Wn * nc, L * 2 * d, da, F8_CROSS_KV8, crosskv_b, cross.p, 0);
            g.scale = qscale; g.d = d; g.rows_per_batch = nc; g.n_batch = B; g.out_scale = cross_sc.as<unsigned char>();
            if (cmap) {
                if (Wn > (int)sizeof(g.batch_map)) throw Error(-1, "internal: cross-KV slot map too small");
                g.use_batch_map = 1;
                for (int i = 0; i < Wn; i++) g.batch_map[i] = (unsigned char)cmap[i];
            }
            launch_gemm_f8<T>(g, st);
            return;
        }
        GemmDesc g = gd(encT.p, da, crosskv_w, Wn * nc, L * 2 * d, da, EPI_CROSS_KV, crosskv_b, cross.p, 0);
        g.scale = qscale; g.d = d; g.rows_per_batch = nc; g.n_batch = B;
        if (cmap) {
            if (Wn > (int)sizeof(g.batch_map)) throw Error(-1, "internal: cross-KV slot map too small");
            g.use_batch_map = 1;
            for (int i = 0; i < Wn; i++) g.batch_map[i] = (unsigned char)cmap[i];
        }
        launch_gemm<T>(g, st);
    }

    // ---- the decoder pass (1..128 rows): 11 launches per layer (10 from rows x heads >= direct_pairs on), see kernels_decode.hip / fused_body ----
    struct Plan { int S, NW; };
    Plan pl_qkv, pl_dd, pl_fc1, pl_fc2, pl_logits;
    DBuf xa, xb, p1, pq, p2, p3;
    void plan_decode() {
        dec_gemv_plan(3 * d, d, &pl_qkv.S, &pl_qkv.NW); pl_qkv.S = 1; fix_nw(pl_qkv, d);
        dec_gemv_plan(d, d, &pl_dd.S, &pl_dd.NW, true);
        dec_gemv_plan(4 * d, d, &pl_fc1.S, &pl_fc1.NW); pl_fc1.S = 1; fix_nw(pl_fc1, d);
        dec_gemv_plan(d, 4 * d, &pl_fc2.S, &pl_fc2.NW);
        pl_logits.S = 1; fix_nw(pl_logits, d);
        const size_t pb = (size_t)4 * kPartRows * d * 4;   // <= 4 split-K slots of kPartRows token rows
        xa.alloc((size_t)kPartRows * d * 4); xb.alloc((size_t)kPartRows * d * 4); p1.alloc(pb); pq.alloc(pb); p2.alloc(pb); p3.alloc(pb);
    }
    void fix_nw(Plan& p, int K) {  // direct epilogues need S == 1: pick the widest block whose per-wave k is a multiple of 32 and <= 320
        for (int nw = 4; nw >= 1; nw >>= 1) if (K % nw == 0 && (K / nw) % 32 == 0 && K / nw <= 320) { p.NW = nw; p.S = 1; return; }
        throw Error(SS_ERR_MODEL, "model: width not supported by the fused decode step");
    }
    DecGemvDesc dgd(int pro, int epi, const void* Wt, int M, int N, int K, int S) {
        DecGemvDesc g{};
        g.pro = pro; g.epi = epi; g.W = Wt; g.M = M; g.N = N; g.K = K; g.S = S; g.scale = 1.0f; g.d = d; g.gelu_f16_in = dtype_is_f16;
        return g;
    }
    // The step is a fixed sequence of ~13 L dependent launches whose arguments do not change between steps (tokens, positions and
    // slots travel in ctl_d): after one plain pass per (M, n_samp) shape it is captured into a hipGraph and replayed.
    struct StepGraph { int uses = 0; long last_use = 0; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
    std::map<int, StepGraph> step_graphs;
    long graph_clock = 0;
    // a long-lived service sees up to 128 x 129 (rows, sampled rows) shapes, in practice the diagonal (every row samples) plus the prompt passes:
    // keep the 256 most recently used (an instantiated step graph of large-v3 holds ~0.4 MB of host memory).  An evicted graph's exec may still be
    // replaying: it goes to a retired list that is reaped once the stream has been synchronised anyway (the end of a group), not by stalling the
    // stream in the middle of one.  n_graph_evictions makes thrash visible (ss_engine_lane_counters).
    static constexpr size_t kMaxStepGraphs = 256;
    std::vector<StepGraph> retired_graphs;
    long n_graph_evictions = 0;
    void evict_step_graphs() {
        while (step_graphs.size() > kMaxStepGraphs) {
            auto victim = step_graphs.begin();
            for (auto it = step_graphs.begin(); it != step_graphs.end(); ++it) if (it->second.last_use < victim->second.last_use) victim = it;
            retired_graphs.push_back(victim->second);
            step_graphs.erase(victim);
            n_graph_evictions++;
        }
        // a group under continuous admission may run for hours: bound the retired list by paying the stream synchronisation once per 64 evictions
        if (retired_graphs.size() >= 64) { SS_HIP(hipStreamSynchronize(st)); reap_retired_graphs(); }
    }
    void reap_retired_graphs() {   // caller has synchronised `st`: no replay of a retired graph is executing
        for (StepGraph& g : retired_graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
        retired_graphs.clear();
    }
    // chained = true: the control blocks are already on the device (the previous step's pick kernel advanced them), nothing is uploaded
    void decoder_step_fused(int M, const RuleConsts& rc, const std::vector<int>& samp_rows, bool any_probs, bool chained = false) {
        const int n_samp = (int)samp_rows.size();
        cnt_passes++; cnt_rows += M;
        if (!chained) {
            SS_HIP(hipMemcpyAsync(ctl_d.p, ctl_h, (size_t)(kPartRows + n_samp) * sizeof(RowCtl), hipMemcpyHostToDevice, st));
            if (n_samp) {
                memcpy(rowidx_h, samp_rows.data(), (size_t)n_samp * 4);
                SS_HIP(hipMemcpyAsync(rowidx_d.p, rowidx_h, (size_t)n_samp * 4, hipMemcpyHostToDevice, st));
            }
        }
        if (!use_graph) fused_body(M, n_samp);
        else {
            if (!step_graphs.count(M * 1024 + n_samp)) evict_step_graphs();
            StepGraph& sg = step_graphs[M * 1024 + n_samp];
            sg.last_use = ++graph_clock;
            if (sg.uses++ == 0) fused_body(M, n_samp);
            else {
                if (!sg.exec) {
                    SS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                    try { fused_body(M, n_samp); } catch (...) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); throw; }
                    SS_HIP(hipStreamEndCapture(st, &sg.graph));
                    SS_HIP(hipGraphInstantiate(&sg.exec, sg.graph, nullptr, nullptr, 0));
                }
                SS_HIP(hipGraphLaunch(sg.exec, st));
            }
        }
        if (n_samp == 0) return;
        const RowCtl* ctl = ctl_d.as<RowCtl>();
        step_parity ^= 1;
        launch_logits_rules(logits.as<float>(), n_vocab_pad, ctl + kPartRows, n_samp, rc, samp_d.as<SampleOut>(), any_probs ? probs.as<float>() : nullptr, rules_scratch.as<float>(), st,
                            any_probs ? nullptr : ctl_d.as<RowCtl>(), rowidx_d.as<int>());
        if (any_probs) draw_on_device(ctl + kPartRows, n_samp);
        SS_HIP(hipMemcpyAsync(samp_hb[step_parity], samp_d.p, (size_t)n_samp * sizeof(SampleOut), hipMemcpyDeviceToHost, st));
        SS_HIP(hipEventRecord(ev_step[step_parity], st));
    }
    void fused_body(int M, int n_samp) {
        const RowCtl* ctl = ctl_d.as<RowCtl>();
        const long slot_stride = (long)n_tctx * d, layer_stride = (long)S * slot_stride;
        const long cb_stride = (long)2 * H * n_ctx * 64, cl_stride = (long)B * cb_stride;
        float* xcur = xa.as<float>(); float* xnext = xb.as<float>();
        const float* prev_parts = nullptr; int prev_np = 0; const float* prev_bias = nullptr;
        // Few rows (the latency configuration: one chunk at a time): every residual-update + LayerNorm launch becomes the prologue of the GEMV that
        // consumes it (kernels.h launch_dec_gemv_ln) -- 8 launches per layer instead of 11, one instead of two for the logits.
        const bool lnf = ln_fuse && M <= ln_fuse_rows && n_samp <= ln_fuse_rows;
        for (int il = 0; il < L; il++) {
            const DecL& e = dec[il];
            if (lnf) {
                DecGemvDesc g = dgd(PRO_LN, DEPI_QKV, e.wqkv, M, 3 * d, d, 1);
                if (il == 0) { g.ctl = ctl; g.tok_emb = tok_emb; g.pos_emb = dec_pos; }
                else { g.x_in = xcur; g.parts = prev_parts; g.n_parts = prev_np; g.bias_prev = prev_bias; }
                g.x_out = xnext; g.ln_w = e.ln1w; g.ln_b = e.ln1b;
                g.bias = e.bqkv; g.out = qd.p; g.ldo = d; g.scale = qscale;
                g.ctl_rows = ctl; g.kcache = kself.as<T>() + il * layer_stride; g.vcache = vself.as<T>() + il * layer_stride; g.slot_stride = slot_stride;
                launch_dec_gemv_ln<T>(g, pl_qkv.NW, st);
                std::swap(xcur, xnext);
            } else {   // x = [embed | x + b2 + sum P3]; LN1 (one wave per row) -> QKV GEMV, q scaled, K/V appended to the cache
                DecGemvDesc r = dgd(PRO_LN, DEPI_PART, nullptr, M, d, d, 1);
                if (il == 0) { r.ctl = ctl; r.tok_emb = tok_emb; r.pos_emb = dec_pos; }
                else { r.x_in = xcur; r.parts = prev_parts; r.n_parts = prev_np; r.bias_prev = prev_bias; }
                r.x_out = xnext; r.ln_w = e.ln1w; r.ln_b = e.ln1b;
                launch_dec_reduce_ln<T>(r, lnd.as<T>(), st);
                std::swap(xcur, xnext);
                DecGemvDesc g = dgd(PRO_T, DEPI_QKV, e.wqkv, M, 3 * d, d, 1);
                g.Xt = lnd.p; g.ldx = d; g.bias = e.bqkv; g.out = qd.p; g.ldo = d; g.scale = qscale;
                g.ctl_rows = ctl; g.kcache = kself.as<T>() + il * layer_stride; g.vcache = vself.as<T>() + il * layer_stride; g.slot_stride = slot_stride;
                launch_dec_gemv<T>(g, pl_qkv.NW, st);
            }
            // key splits exist to fill the chip when there are few (row, head) pairs; from M * H >= direct_pairs on, one workgroup per pair
            // streams all 1500 keys and writes the normalised output itself: no partials, no combine launch.  (Folding the row-local projections
            // around the two attention blocks into such (row, head) workgroups -- 7 launches per layer instead of 11 -- was built, is oracle-green
            // and measured SLOWER: 4.3-4.45 ms against 3.65 ms per 32-row pass; tools/experiments/r03_fused_attention_blocks/.)
            const bool direct = M * H >= direct_pairs;
            launch_dec_self_attention<T>(qd.as<T>(), kself.as<T>() + il * layer_stride, vself.as<T>() + il * layer_stride, slot_stride, d, H, ctl, M,
                                         attd.as<T>(), st);
            const int n_qpart = pl_dd.S;   // the same split-K plan whatever the row count: a row's q bits must not depend on the pass it rides in
            {
                {   // attention out-projection, split-K partials
                    DecGemvDesc g = dgd(PRO_T, DEPI_PART, e.wo, M, d, d, pl_dd.S);
                    g.Xt = attd.p; g.ldx = d; g.part_out = p1.as<float>();
                    launch_dec_gemv<T>(g, pl_dd.NW, st);
                }
                if (lnf) {   // x += bo + sum P1; LNc; cross query (the S-way partials of the wide form) in one launch
                    DecGemvDesc g = dgd(PRO_LN, DEPI_PART, e.wcq, M, d, d, pl_dd.S);
                    g.x_in = xcur; g.x_out = xnext; g.parts = p1.as<float>(); g.n_parts = pl_dd.S; g.bias_prev = e.bo; g.ln_w = e.lncw; g.ln_b = e.lncb;
                    g.part_out = pq.as<float>();
                    launch_dec_gemv_ln<T>(g, pl_dd.NW, st);
                    std::swap(xcur, xnext);
                } else {
                // x += bo + sum P1; LNc -> cross query partials (reduced, biased and scaled inside the cross-attention kernel)
                DecGemvDesc r = dgd(PRO_LN, DEPI_PART, nullptr, M, d, d, 1);
                r.x_in = xcur; r.x_out = xnext; r.parts = p1.as<float>(); r.n_parts = pl_dd.S; r.bias_prev = e.bo; r.ln_w = e.lncw; r.ln_b = e.lncb;
                launch_dec_reduce_ln<T>(r, lnd.as<T>(), st);
                std::swap(xcur, xnext);
                DecGemvDesc g = dgd(PRO_T, DEPI_PART, e.wcq, M, d, d, pl_dd.S);
                g.Xt = lnd.p; g.ldx = d; g.part_out = pq.as<float>();
                launch_dec_gemv<T>(g, pl_dd.NW, st);
                }
            }
            const T* kc = cross.as<T>() + il * cl_stride;
            if (fp8_enc) {   // e4m3 cross cache: the same two forms over codes + exponent bytes (half the bytes of the stream that bounds the pass)
                const long sc_b = (long)2 * H * n_ctx;
                launch_dec_cross_attention_f8<T>(pq.as<float>(), n_qpart, e.bcq, qscale, cross.as<unsigned char>() + il * cl_stride,
                                                 cross_sc.as<unsigned char>() + il * (long)B * sc_b, cb_stride, sc_b, d, H, n_ctx, ctl, M,
                                                 direct ? nullptr : cscratch.as<float>(), attd.as<T>(), st);
                if (!direct) launch_dec_cross_combine<T>(cscratch.as<float>(), d, H, M, attd.as<T>(), st);
            } else if (direct) {
                launch_dec_cross_attention_direct<T>(pq.as<float>(), n_qpart, e.bcq, qscale, kc, kc + (long)H * n_ctx * 64, cb_stride, d, H, n_ctx, ctl, M,
                                                     attd.as<T>(), st);
            } else {
                launch_dec_cross_attention_q<T>(pq.as<float>(), n_qpart, e.bcq, qscale, kc, kc + (long)H * n_ctx * 64, cb_stride, d, H, n_ctx, ctl, M,
                                                cscratch.as<float>(), st);
                launch_dec_cross_combine<T>(cscratch.as<float>(), d, H, M, attd.as<T>(), st);
            }
            {   // cross out-projection, split-K partials
                DecGemvDesc g = dgd(PRO_T, DEPI_PART, e.wco, M, d, d, pl_dd.S);
                g.Xt = attd.p; g.ldx = d; g.part_out = p2.as<float>();
                launch_dec_gemv<T>(g, pl_dd.NW, st);
            }
            if (lnf) {   // x += bco + sum P2; LN2; FC1 + GELU in one launch
                DecGemvDesc g = dgd(PRO_LN, DEPI_GELU_T, e.w1, M, 4 * d, d, 1);
                g.x_in = xcur; g.x_out = xnext; g.parts = p2.as<float>(); g.n_parts = pl_dd.S; g.bias_prev = e.bco; g.ln_w = e.ln2w; g.ln_b = e.ln2b;
                g.bias = e.b1; g.out = ffd.p; g.ldo = 4 * d;
                launch_dec_gemv_ln<T>(g, pl_fc1.NW, st);
                std::swap(xcur, xnext);
            } else {   // x += bco + sum P2; LN2 -> FC1 + GELU
                DecGemvDesc r = dgd(PRO_LN, DEPI_PART, nullptr, M, d, d, 1);
                r.x_in = xcur; r.x_out = xnext; r.parts = p2.as<float>(); r.n_parts = pl_dd.S; r.bias_prev = e.bco; r.ln_w = e.ln2w; r.ln_b = e.ln2b;
                launch_dec_reduce_ln<T>(r, lnd.as<T>(), st);
                std::swap(xcur, xnext);
                DecGemvDesc g = dgd(PRO_T, DEPI_GELU_T, e.w1, M, 4 * d, d, 1);
                g.Xt = lnd.p; g.ldx = d; g.bias = e.b1; g.out = ffd.p; g.ldo = 4 * d;
                launch_dec_gemv<T>(g, pl_fc1.NW, st);
            }
            {   // FC2 partials (bias and residual are applied by the next reduce)
                DecGemvDesc g = dgd(PRO_T, DEPI_PART, e.w2, M, d, 4 * d, pl_fc2.S);
                g.Xt = ffd.p; g.ldx = 4 * d; g.part_out = p3.as<float>();
                launch_dec_gemv<T>(g, pl_fc2.NW, st);
            }
            prev_parts = p3.as<float>(); prev_np = pl_fc2.S; prev_bias = e.b2;
        }
        if (n_samp == 0) return;
        if (lnf) {
            DecGemvDesc g = dgd(PRO_LN, DEPI_LOGITS, tok_emb, n_samp, n_vocab_pad, d, 1);
            g.x_in = xcur; g.parts = prev_parts; g.n_parts = prev_np; g.bias_prev = prev_bias; g.ln_w = lnw; g.ln_b = lnb; g.row_idx = rowidx_d.as<int>();
            g.out = logits.p; g.ldo = n_vocab_pad; g.n_valid = n_vocab;
            launch_dec_gemv_ln<T>(g, pl_logits.NW, st);
            return;
        }
        {   // final LayerNorm once (gathering the sampling rows, folding FC2's bias + partials), then logits = x . tok_emb^T
            DecGemvDesc g = dgd(PRO_LN, DEPI_LOGITS, tok_emb, n_samp, n_vocab_pad, d, 1);
            g.x_in = xcur; g.parts = prev_parts; g.n_parts = prev_np; g.bias_prev = prev_bias; g.ln_w = lnw; g.ln_b = lnb; g.row_idx = rowidx_d.as<int>();
            launch_dec_reduce_ln<T>(g, lnd.as<T>(), st);
            DecGemvDesc q = dgd(PRO_T, DEPI_LOGITS, tok_emb, n_samp, n_vocab_pad, d, 1);
            q.Xt = lnd.p; q.ldx = d; q.out = logits.p; q.ldo = n_vocab_pad; q.n_valid = n_vocab;
            launch_dec_gemv<T>(q, pl_logits.NW, st);
        }
    }

    // One decoder launch over M rows described by ctl_h[0..M).  Rows may belong to the same decoder (a multi-token
    // prompt): K/V of every row are written to the cache before the attention kernels run, and each row attends to
    // cache positions <= its own, so causality holds without a mask.  The n_samp rows listed in samp_rows (with their
    // rule state in ctl_h[kPartRows .. kPartRows + n_samp)) get logits + rules; results land in samp_h[0..n_samp).
    // the sampled rows' draws: u_h[k] was filled by round_rows (one generate_canonical per sampled row, in row order)
    void draw_on_device(const RowCtl* ctl_rows, int n_samp) {
        SS_HIP(hipMemcpyAsync(u_d.p, u_h, (size_t)n_samp * sizeof(double), hipMemcpyHostToDevice, st));
        launch_sample_draw(probs.as<float>(), n_vocab_pad, n_vocab, ctl_rows, n_samp, u_d.as<double>(), samp_d.as<SampleOut>(), st);
    }
    // one decoder pass (<= kPartRows rows); returns the parity of the result buffer / event to wait on
    int decoder_step(int M, const RuleConsts& rc, const std::vector<int>& samp_rows, bool any_probs) {
        if (M < 1 || M > kPartRows) throw Error(-1, "internal: decoder pass of " + std::to_string(M) + " rows");
        decoder_step_fused(M, rc, samp_rows, any_probs);
        return step_parity;
    }
    DBuf samp_d, rowidx_d, rules_scratch, ns_mask_d;
    long cnt_passes = 0, cnt_rows = 0, cnt_windows = 0, cnt_admitted = 0, cnt_midstart = 0;   // of the running group: decoder passes (one read of the decoder weights each), rows, windows
    bool use_graph = true, chain_steps = true;
    int ln_fuse_rows = kLnFuseRows;   // (dev: SS_LN_FUSE_ROWS, <= 16, to re-measure where the fusion stops paying)
    bool ln_fuse = true;      // SS_LN_FUSE=0: A/B switch of the LayerNorm-prologue launches for <= kLnFuseRows rows
    bool vt_gemm = true;      // the V projection as its own GEMM with the transposing epilogue.  SS_VT_GEMM=0 (A/B, round 5): Q | K | V in one GEMM as plain rows +
                              // a V -> V^T pass through LDS tiles -- same bits, the V third at 0.41 instead of 0.25 of the MFMA pipe, and 1.4 % SLOWER end to end
                              // (profiles/r05_ah_vt_gemm_ab.txt): the extra 61 MB pass and launch per layer cost more than the epilogue did
    int cb_start_min = 4;     // SS_CB_START_MIN: windows that must be waiting before a running group pauses its decoders for their encoder pass
    static constexpr int direct_pairs = 320;   // (rows x heads) from which the cross-attention runs as one workgroup per (row, head) (large-v3: 16 rows; +2.3 % at 32-row passes, -12 % at 8; same bits either way)

    RuleConsts rule_consts(const ss_params& P) {
        const Vocab& v = hm.vocab;
        RuleConsts rc{};
        rc.n_vocab = n_vocab; rc.eot = v.token_eot; rc.sot = v.token_sot; rc.translate = v.token_translate; rc.transcribe = v.token_transcribe;
        rc.solm = v.token_solm; rc.prev = v.token_prev; rc.nosp = v.token_nosp; rc.not_ = v.token_not; rc.beg = v.token_beg;
        auto it = v.token_to_id.find(" ");
        rc.blank = it == v.token_to_id.end() ? -1 : it->second;
        rc.n_lang = v.num_languages();
        rc.suppress_blank = P.suppress_blank; rc.no_timestamps = P.no_timestamps; rc.tdrz_enable = P.tdrz_enable;
        rc.max_initial_tid = -1;
        if (P.max_initial_ts > 0.0f) {
            const float precision = float(kChunkSec) / hm.hp.n_audio_ctx;
            rc.max_initial_tid = (int)std::round(P.max_initial_ts / precision);
        }
        rc.suppress_eot = P.fixed_steps > 0;
        rc.openai_ts = (compat & SS_COMPAT_OPENAI_TS_RULES) != 0;
        rc.ns_mask = P.suppress_non_speech_tokens ? ns_mask_d.as<uint32_t>() : nullptr;
        return rc;
    }

    // ------------------------------------------------------------------------------------------
    // the batched whisper_full_with_state
    // ------------------------------------------------------------------------------------------
    void run_jobs(std::vector<Job*>& jobs) override {
        std::lock_guard<std::mutex> lk(mu);
        run_jobs_locked(jobs);
    }
    void run_jobs_locked(std::vector<Job*>& jobs) override {
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        for (size_t i = 0; i < jobs.size(); i += B) {
            std::vector<Job*> grp(jobs.begin() + i, jobs.begin() + std::min(jobs.size(), i + (size_t)B));
            try {
                run_group(grp);
            } catch (const Error& e) {
                for (Job* j : grp) if (j && j->status == 0) { j->status = e.code; j->err = e.what(); }   // finished chunks were nulled: theirs to free
                throw;
            }
        }
    }

    // whisper_full_with_state up to the window loop for one chunk: session reset, prompt, parameter checks, language, log-mel on the device.
    // `i` = the batch slot whose device buffers (pcm_d / mel_d / fmax_d) the chunk uses.  A chunk that is refused or has nothing to decode
    // comes back with alive == false (status says why).
    JobState setup_job(Job* j, int i) {
            const Vocab& vocab = hm.vocab;
            Session* s = j->sess;
            s->segments.clear(); s->tokens.clear(); s->sampled.clear(); s->trace.clear(); s->n_encode = s->n_decode = s->n_fail = s->n_windows = 0;   // "clear old results"
            // The carried text context (prompt_past) is only touched once the chunk is known to decode: whisper_full_with_state returns for < 1 s
            // of audio before it reaches "if (params.no_context) prompt_past.clear()" / "prepend the prompt tokens", and a call refused before that
            // point must leave a context-carrying caller's state as it was (touch_context below; the one refusal that comes AFTER it in
            // whisper.cpp, audio_ctx > n_audio_ctx -> -5, applies it first).
            JobState q; q.job = j; q.slot = (int)i;
            j->status = 0;
            const ss_params& P = j->P;
            if (P.token_timestamps) { s->t_beg = 0; s->t_last = 0; s->tid_last = 0; }   // reset at the top of every whisper_full_with_state call
            // What whisper_full_with_state has already done to the state by the time it checks audio_ctx ("overwrite audio_ctx, max allowed is
            // hparams.n_audio_ctx": return -5): decoders 1.. set up (their generators re-seeded), no_context applied, the prompt tokens prepended --
            // all of it only for a chunk of >= 1 s that is not a detect_language call (those return earlier).
            auto touch_context = [&]() {
                if (!(compat & SS_COMPAT_RNG_STATE)) s->rng_dec.assign((size_t)std::max(0, (int)P.best_of - 1), CountingRng());
                if (P.no_context) s->prompt_past.clear();
                if (!j->prompt_tokens.empty()) {   // "prepend the prompt tokens to the prompt_past"
                    s->prompt_past.insert(s->prompt_past.end(), j->prompt_tokens.begin(), j->prompt_tokens.end());
                    std::rotate(s->prompt_past.begin(), s->prompt_past.end() - j->prompt_tokens.size(), s->prompt_past.end());
                }
            };
            if (P.audio_ctx > n_ctx) {
                j->status = SS_ERR_AUDIO_CTX; j->err = "audio_ctx larger than the model's n_audio_ctx";
                const int s0 = P.offset_ms / 10, s1 = P.duration_ms == 0 ? (j->n_samples > 0 ? mel_n_len_org(j->n_samples) : 0) : s0 + P.duration_ms / 10;
                if (P.best_of <= ND && P.offset_ms >= 0 && P.duration_ms >= 0 && j->n_samples > 0 && s1 >= s0 + 100 && !P.detect_language) touch_context();
                return q;
            }
            if (P.audio_ctx != 0 && P.audio_ctx != n_ctx) {   // the reference passes 1500 or 0 (whisper.rs:144,68); a whisper-rs caller may shorten the context
                // the V^T epilogue stores 4 consecutive positions per lane and the cross-attention walks 4 equal key ranges: multiples of 4 only
                if (P.audio_ctx < 0 || P.audio_ctx % 4) { j->status = SS_ERR_UNSUPPORTED; j->err = "audio_ctx must be a positive multiple of 4 (or 0 = the model's n_audio_ctx)"; return q; }
                // whisper.cpp detects the language BEFORE it installs params.audio_ctx in the state (the pass runs on whatever the previous call left there)
                if (P.language[0] == 0 || !strcmp(P.language, "auto") || P.detect_language) {
                    j->status = SS_ERR_UNSUPPORTED; j->err = "language detection together with a shortened audio_ctx"; return q;
                }
            }
            if (P.best_of > ND) { j->status = SS_ERR_ARG; j->err = "best_of exceeds the engine's max_decoders"; return q; }
            if (P.offset_ms < 0 || P.duration_ms < 0) { j->status = SS_ERR_ARG; j->err = "negative offset_ms / duration_ms"; return q; }
            // "auto-detect language if not specified": language nullptr / "" / "auto" or detect_language (whisper_full_with_state)
            const bool auto_lang = P.language[0] == 0 || !strcmp(P.language, "auto") || P.detect_language;
            s->lang_id = -1;
            if (vocab.is_multilingual()) {
                if (auto_lang) {
                    // whisper_lang_auto_detect looks at the window at offset 0; here it shares the first window's encoder pass
                    if (P.offset_ms != 0) { j->status = SS_ERR_UNSUPPORTED; j->err = "language detection together with offset_ms"; return q; }
                    q.need_detect = true;
                } else {
                    const int lid = lang_id(P.language);
                    if (lid < 0 || lid >= vocab.num_languages()) { j->status = SS_ERR_LANG; j->err = std::string("unknown language '") + P.language + "'"; return q; }
                    s->lang_id = lid;
                    set_prompt_init(q, lid);
                }
            } else {
                if (P.detect_language) { j->status = SS_ERR_LANG; j->err = "detect_language on an English-only model"; return q; }
                set_prompt_init(q, 0);   // .en models: [sot] only, whatever the language says
            }
            if (j->n_samples > 0) {
                q.n_len = mel_n_len(j->n_samples); q.n_len_org = mel_n_len_org(j->n_samples);
                const float* dp;
                if (j->pcm_on_device) dp = j->pcm;
                else {
                    if (!pcm_sized) {   // as mel_sized below, for callers that hand over host PCM
                        for (int b = 0; b < B; b++) pcm_d[b].ensure(std::max((size_t)j->n_samples, (size_t)kSampleRate * kChunkSec) * 4);
                        pcm_sized = true;
                    }
                    pcm_d[i].ensure((size_t)j->n_samples * 4);
                    SS_HIP(hipMemcpyAsync(pcm_d[i].p, j->pcm, (size_t)j->n_samples * 4, hipMemcpyHostToDevice, st));
                    dp = pcm_d[i].as<float>();
                }
                if (!mel_sized) {
                    // First chunk on this lane: size EVERY slot's staging buffers for a 30 s chunk now.  Sized slot by slot on first use, each
                    // hipMalloc + zero fill + stream synchronisation (DBuf::alloc) stalls the lane once per slot -- spread over the first
                    // max_batch chunks a lane sees, i.e. inside the timed region of a short benchmark and over the first minutes of a service.
                    const int len30 = mel_n_len(kSampleRate * kChunkSec);
                    for (int b = 0; b < B; b++) { mel_d[b].ensure((size_t)n_mel * std::max(q.n_len, len30) * 4); fmax_d[b].ensure((size_t)std::max(q.n_len, len30) * 4); }
                    mel_sized = true;
                }
                mel_d[i].ensure((size_t)n_mel * q.n_len * 4);
                fmax_d[i].ensure((size_t)q.n_len * 4);
                launch_log_mel(mt, dp, j->n_samples, mel_d[i].as<float>(), q.n_len, fmax_d[i].as<float>(), st);
                if (P.token_timestamps) {   // "state->energy = get_signal_energy(samples, n_samples, 32)": on the device, then to the host for the per-segment pass
                    const size_t eb = (size_t)j->n_samples * 4;
                    if (!energy_sized) {   // first use on this lane: give EVERY slot room for a whole 30 s window now -- pinning host pages costs ~0.1 - 1 ms
                                           // per buffer, and paid slot by slot it would be spread over the first max_batch chunks of a service's life
                        const size_t first = std::max(eb, (size_t)kSampleRate * kChunkSec * 4);
                        for (int b = 0; b < B; b++) energy_h[b].ensure(first);
                        energy_sized = true;
                    }
                    if (eb > energy_h[i].bytes) {     // a longer chunk than this slot has seen: an earlier (refused) chunk's kernel may still be writing the old buffer
                        SS_HIP(hipStreamSynchronize(st));
                        energy_h[i].ensure(eb);
                    }
                    launch_signal_energy(dp, j->n_samples, (float*)energy_h[i].dev, st);
                    q.energy = (const float*)energy_h[i].p; q.n_energy = j->n_samples;
                }
                q.seek_start = P.offset_ms / 10;
                q.seek = q.seek_start; q.seek_end = P.duration_ms == 0 ? q.n_len_org : q.seek_start + P.duration_ms / 10;
                q.alive = q.seek_end >= q.seek_start + 100;  // "if length of spectrogram is less than 1.0s, return"
            }
            if (q.alive && !P.detect_language) touch_context();   // detect_language returns right after the detection, before the context is touched
            return q;
    }

    void run_group(std::vector<Job*>& grp) {
        hook_owner = nullptr;      // this lane's cache slots now belong to the group (the stage hooks' context lived in slot 0)
        const Vocab& vocab = hm.vocab;
        std::vector<JobState> js;
        SS_HIP(hipEventRecord(ev[0], st));
        for (size_t i = 0; i < grp.size(); i++) js.push_back(setup_job(grp[i], (int)i));
        SS_HIP(hipEventRecord(ev[1], st));
        cnt_passes = cnt_rows = cnt_windows = cnt_admitted = cnt_midstart = 0;
        // A chunk is reported the moment its last window is finalised (or it is refused), not when the slowest chunk of the group is done, and
        // a group that keeps running (multi-window chunks, natural-length decodes) takes queued chunks into its free slots at the next window
        // boundary: continuous batching at window granularity.  Finished entries forget their Job (its owner may free it at once).
        auto sweep = [&]() {
            int n_alive = 0;
            for (auto& q : js) {
                if (!q.job) continue;
                if (q.alive && q.job->status == 0 && q.seek + 100 >= q.seek_end) q.alive = false;
                if (!q.alive || q.job->status != 0) {
                    Job* j = q.job;
                    q.job = nullptr; q.alive = false;
                    for (auto& gj : grp) if (gj == j) gj = nullptr;
                    owner->job_finished(j, this);
                } else n_alive++;
            }
            return n_alive;
        };
        // ---- continuous batching at token granularity ------------------------------------------------------------------------------------------
        // Windows are independent state machines (whisper_full's per-window body: temperature ladder, best_of decoders, segmenting).  Every
        // iteration of this loop (1) starts windows for chunks that have none -- the next window of a running chunk, or a chunk admitted from
        // the queue -- with one encoder pass over just those windows, their cross-KV going to free cache slots; (2) advances ALL active
        // decoders of ALL active windows by one round (one decoder pass); (3) closes the attempts that ended: retry at the next temperature,
        // or finalise the window, free its slots and report the chunk if that was its last window.  Rows freed by an early EOT are therefore
        // taken over by new windows after at most `start_min` of them are waiting, not when the slowest window of a group is done.
        std::list<Window> active;
        std::vector<int> free_cross, free_dec;
        for (int i = B - 1; i >= 0; i--) free_cross.push_back(i);
        for (int i = S - 1; i >= 0; i--) free_dec.push_back(i);
        // encoder sections are stamped as they are enqueued; a pair is folded into ms_enc (and its events go back to the lane's pool) as soon as
        // it has completed, so a long-lived group holds a handful of events, and an exception on the way out releases the rest (EncEvents dtor)
        struct EncEvents {
            std::vector<hipEvent_t>& pool; std::deque<std::pair<hipEvent_t, hipEvent_t>> live; float ms = 0.f;
            explicit EncEvents(std::vector<hipEvent_t>& p) : pool(p) {}
            hipEvent_t take() { if (pool.empty()) { hipEvent_t e; SS_HIP(hipEventCreate(&e)); return e; } hipEvent_t e = pool.back(); pool.pop_back(); return e; }
            void fold(bool wait) {
                while (!live.empty()) {
                    if (!wait && hipEventQuery(live.front().second) != hipSuccess) break;
                    float a = 0;
                    if (hipEventElapsedTime(&a, live.front().first, live.front().second) == hipSuccess) ms += a;
                    pool.push_back(live.front().first); pool.push_back(live.front().second);
                    live.pop_front();
                }
            }
            ~EncEvents() { for (auto& p : live) { pool.push_back(p.first); pool.push_back(p.second); } }
        } enc_events(enc_event_pool);
        auto n_jobs_alive = [&]() { int n = 0; for (auto& q : js) n += q.job && q.alive; return n; };
        while (true) {
            sweep();
            // (1) which chunks need a window?
            std::vector<JobState*> need;
            for (auto& q : js) if (q.job && q.alive && !q.has_window) need.push_back(&q);
            if (from_queue && n_jobs_alive() < B) {
                std::vector<Job*> more;
                owner->admit_more(B - n_jobs_alive(), this, more);
                cnt_admitted += (long)more.size();
                if (!more.empty()) {
                    for (Job* j : more) {
                        int slot = -1;
                        std::vector<bool> used(B, false);
                        for (auto& q : js) if (q.job) used[q.slot] = true;
                        for (int i = 0; i < B; i++) if (!used[i]) { slot = i; break; }
                        grp.push_back(j);
                        js.push_back(setup_job(j, slot));
                    }
                    sweep();
                    need.clear();
                    for (auto& q : js) if (q.job && q.alive && !q.has_window) need.push_back(&q);
                }
            }
            if (active.empty() && need.empty()) break;
            // starting windows stalls the decoders that are running for one encoder pass, and an encoder pass over a single window fills
            // less than half the chip: wait until `start_min` windows can share it unless nothing else is running
            if (!need.empty() && (active.empty() || (int)need.size() >= std::min(cb_start_min, std::max(1, (n_jobs_alive() + 1) / 2))) && !free_cross.empty()) {
                spec.valid = false;
                const bool others_running = !active.empty();
                auto ctx_of = [&](const JobState& q) { const int a = q.job->P.audio_ctx; return a > 0 && a < n_ctx ? a : n_ctx; };
                // One context (whisper_full_params.audio_ctx) per encoder pass.  Chunks that ask for another get THEIR pass right behind this one, in
                // the same round, while cross slots last (re-applying the start threshold let a minority context be passed over again and
                // again under load; every skipped pass still paused the running decoders).
                std::vector<JobState*> waiting = need;
                while (!waiting.empty() && !free_cross.empty()) {
                std::vector<Window*> fresh;
                std::vector<int> cmap;
                std::vector<JobState*> rest;
                const int pass_ctx = ctx_of(*waiting.front());
                set_context(pass_ctx);
                for (JobState* qp : waiting) {
                    JobState& q = *qp;
                    if (free_cross.empty() || ctx_of(q) != pass_ctx) { rest.push_back(qp); continue; }
                    active.emplace_back();
                    Window& w = active.back();
                    w.job = q.job; w.cross = free_cross.back(); free_cross.pop_back();
                    w.n_keys = pass_ctx == n_ctx ? 0 : pass_ctx;
                    const ss_params& P = q.job->P;
                    if (P.fixed_steps > 0) w.temperatures = {0.0f};
                    else if (P.temperature_inc > 0.0f) for (float t = P.temperature; t < 1.0f + 1e-6f; t += P.temperature_inc) w.temperatures.push_back(t);
                    else w.temperatures = {P.temperature};
                    // "if there is a very short audio segment left to process, we remove any past prompt"
                    if (q.seek > q.seek_start && q.seek + 500 >= q.seek_end) q.job->sess->prompt_past.clear();
                    launch_mel_window<T>(mel_d[q.slot].template as<float>(), n_mel, q.n_len, q.seek, 2 * nc,
                                         x0.as<T>() + (size_t)fresh.size() * (2 * nc + 2) * n_mel, st);
                    q.has_window = true;
                    cmap.push_back(w.cross);
                    fresh.push_back(&w);
                }
                const int Wn = (int)fresh.size();
                enc_events.fold(false);
                hipEvent_t e0 = enc_events.take(), e1 = enc_events.take();
                enc_events.live.push_back({e0, e1});
                SS_HIP(hipEventRecord(e0, st));
                encoder_pass(Wn, false);
                cross_kv_pass(Wn, cmap.data());
                SS_HIP(hipEventRecord(e1, st));
                for (Window* w : fresh) { w->job->sess->n_encode++; w->job->sess->n_windows++; }
                cnt_windows += Wn;
                if (others_running) cnt_midstart += Wn;
                detect_languages(fresh, js, free_dec);
                for (Window* w : fresh) {
                    if (w->skip) { w->done = true; continue; }
                    w->it = 0;
                    start_attempt(*w, js, free_dec);
                }
                waiting.swap(rest);
                }
            }
            // (2) one round of every active decoder
            decode_round(active, js);
            // (3) attempts that ended
            for (auto it = active.begin(); it != active.end();) {
                Window& w = *it;
                bool running = false;
                for (auto& dq : w.decs) running |= dq.active;
                if (!w.done && !running && !w.skip) {
                    spec.valid = false;
                    for (auto& dq : w.decs) free_dec.push_back(dq.slot);
                    if (evaluate_attempt(w)) w.done = true;
                    else { w.it++; start_attempt(w, js, free_dec); }
                }
                if (w.done) {
                    spec.valid = false;
                    JobState& jq = state_of(js, w.job);
                    if (!w.skip) finalize_window(w, js);
                    jq.has_window = false;
                    free_cross.push_back(w.cross);
                    it = active.erase(it);
                } else ++it;
            }
        }
        sweep();
        SS_HIP(hipEventRecord(ev[2], st));
        SS_HIP(hipEventSynchronize(ev[2]));
        reap_retired_graphs();   // nothing of this lane's stream is executing any more
        float ms_mel = 0, ms_tot = 0;
        SS_HIP(hipEventElapsedTime(&ms_mel, ev[0], ev[1]));
        SS_HIP(hipEventElapsedTime(&ms_tot, ev[0], ev[2]));
        enc_events.fold(true);   // everything has completed by now
        const float ms_enc = enc_events.ms;
        const float ms_dec = std::max(0.0f, ms_tot - ms_mel - ms_enc);   // decoder passes + their host turnarounds: what is left of the group
        {
            std::lock_guard<std::mutex> sl(stat_mu);
            last_ms[0] = ms_mel; last_ms[1] = ms_enc; last_ms[2] = ms_dec; last_ms[3] = ms_tot;
            last_cnt[0] = cnt_passes; last_cnt[1] = cnt_rows; last_cnt[2] = cnt_windows; last_cnt[3] = cnt_admitted;
            for (int i = 0; i < 4; i++) { tot_ms[i] += last_ms[i]; tot_cnt[i] += last_cnt[i]; }
            tot_cnt[4] += cnt_midstart;
            tot_cnt[5] = n_graph_evictions;
        }
        owner->last_lane.store(lane_index);
    }

    // "these tokens determine the task that will be performed": [sot, lang, task] (multilingual) or [sot], + [notimestamps]
    void set_prompt_init(JobState& q, int lid) {
        const Vocab& vocab = hm.vocab;
        const ss_params& P = q.job->P;
        q.prompt_init = {vocab.token_sot};
        if (vocab.is_multilingual()) {
            q.prompt_init.push_back(vocab.token_sot + 1 + lid);
            q.prompt_init.push_back(P.translate ? vocab.token_translate : vocab.token_transcribe);
        }
        if (P.no_timestamps) q.prompt_init.push_back(vocab.token_not);
    }
    // whisper_lang_auto_detect for the windows whose job asked for it: one decoder row [sot] at position 0 per window on the cross-KV just
    // computed, language = argmax of the raw logits over the language tokens sot+1+id, id in [0, 100) (softmax is monotonic)
    void detect_languages(std::vector<Window*>& wins, std::vector<JobState>& js, std::vector<int>& free_dec) {
        const Vocab& vocab = hm.vocab;
        std::vector<Window*> need;
        for (Window* w : wins) if (state_of(js, w->job).need_detect) need.push_back(w);
        if (need.empty()) return;
        spec.valid = false;
        ss_params P0; memset(&P0, 0, sizeof(P0));
        const RuleConsts rc = rule_consts(P0);
        constexpr int kLangs = 100;   // g_lang: 100 entries whatever the model's vocabulary holds
        std::vector<float> lg((size_t)kLangs);
        for (size_t r0 = 0; r0 < need.size(); r0 += 16) {
            const int M = (int)std::min<size_t>(std::min<size_t>(16, free_dec.size()), need.size() - r0);
            if (M < 1) throw Error(-1, "internal: no decoder slot free for language detection");
            stage_acquire();
            std::vector<int> sr;
            for (int m = 0; m < M; m++) {
                RowCtl c{};
                // a decoder slot is borrowed for this one row (position 0 of a free slot: nothing else uses it until the row has run)
                c.token = vocab.token_sot; c.pos = 0; c.slot = free_dec[free_dec.size() - 1 - m]; c.cross = need[r0 + m]->cross; c.n_hist = 1;
                ctl_h[m] = c; ctl_h[kPartRows + m] = c;
                sr.push_back(m);
            }
            decoder_step(M, rc, sr, false);
            stage_release();
            SS_HIP(hipStreamSynchronize(st));
            for (int m = 0; m < M; m++) {
                const int first = vocab.token_sot + 1, n = std::min(kLangs, n_vocab - first);
                SS_HIP(hipMemcpyAsync(lg.data(), logits.as<float>() + (size_t)m * n_vocab_pad + first, (size_t)n * 4, hipMemcpyDeviceToHost, st));
                SS_HIP(hipStreamSynchronize(st));
                int best = 0;
                for (int i = 1; i < n; i++) if (lg[i] > lg[best]) best = i;
                JobState& q = state_of(js, need[r0 + m]->job);
                q.need_detect = false;
                q.job->sess->lang_id = best;
                if (q.job->P.detect_language) { q.alive = false; need[r0 + m]->skip = true; }   // "if (params.detect_language) return 0"
                else set_prompt_init(q, best);
            }
        }
    }

    JobState& state_of(std::vector<JobState>& js, Job* j) { for (auto& q : js) if (q.job == j && j) return q; throw Error(-1, "internal: job state"); }

    // "init prompt and kv cache for the current iteration" of one window at its ladder position w.it: the prompt ([prev] + past text + sot/lang/task)
    // and the decoders (1 at t = 0, best_of at t > 0), each on a free self-KV slot
    void start_attempt(Window& w, std::vector<JobState>& js, std::vector<int>& free_dec) {
        const ss_params& P = w.job->P;
        const float t_cur = w.temperatures[w.it];
        const int nd = (t_cur > 0.0f) ? std::max(1, (int)P.best_of) : 1;
        Session* s = w.job->sess;
        JobState& jq = state_of(js, w.job);
        w.prompt.clear();
        if (!s->prompt_past.empty() && t_cur < 0.5f && P.n_max_text_ctx > 0) {
            const int n_take = std::min(std::min((int)P.n_max_text_ctx, n_tctx / 2 - ((compat & SS_COMPAT_OPENAI_HISTORY) ? 2 : 0)), (int)s->prompt_past.size());
            w.prompt.push_back(hm.vocab.token_prev);
            w.prompt.insert(w.prompt.end(), s->prompt_past.end() - n_take, s->prompt_past.end());
        }
        w.prompt.insert(w.prompt.end(), jq.prompt_init.begin(), jq.prompt_init.end());
        if ((int)free_dec.size() < nd) throw Error(-1, "internal: decoder slots exceeded");
        w.decs.assign(nd, Dec());
        for (int j = 0; j < nd; j++) {
            Dec& q = w.decs[j];
            q.seek_delta = 100 * kChunkSec; q.active = true; q.slot = free_dec.back(); free_dec.pop_back();
        }
    }
    // the end of an attempt: score the decoders, pick the best, decide between accepting and falling back to the next temperature
    bool evaluate_attempt(Window& w) {
        const ss_params& P = w.job->P;
        double best_score = -INFINITY;
        w.best = 0;
        for (size_t j = 0; j < w.decs.size(); j++) {
            Dec& dq = w.decs[j];
            if (dq.failed) continue;
            dq.sampled.resize(dq.tokens.size());
            for (size_t k = 0; k < dq.tokens.size(); k++) dq.sampled[k] = dq.tokens[k].id;
            dq.tokens.resize(dq.result_len);
            sequence_score(P, dq);
            if (P.fixed_steps == 0 && dq.result_len > 32 && dq.entropy < P.entropy_thold) { dq.failed = true; continue; }
            if (best_score < dq.score) { best_score = dq.score; w.best = (int)j; }
        }
        bool success = true;
        if (w.it != (int)w.temperatures.size() - 1) {
            const Dec& dq = w.decs[w.best];
            if (dq.failed || dq.avg_logprobs < P.logprob_thold) { success = false; w.job->sess->n_fail++; }
        }
        return success;
    }
    // One round: every active decoder of every active window advances to its next sampling point (the first round of an attempt feeds the
    // whole prompt, later rounds one token), as rows of the same decoder pass(es); rows are grouped by rule-constant signature (normally one group).
    struct RowRef { Window* w; int j; bool sample; };
    struct Spec { bool valid = false; int parity = 0; RuleConsts rc; std::vector<std::pair<Window*, int>> decs; } spec;   // a chained step in flight
    void decode_round(std::list<Window>& active, std::vector<JobState>& js) {
        std::vector<RowRef> all;
        for (Window& w : active)
            for (int j = 0; j < (int)w.decs.size(); j++) if (w.decs[j].active) all.push_back({&w, j, false});
        size_t pos0 = 0;
        while (pos0 < all.size()) {
            const ss_params& P0 = all[pos0].w->job->P;
            auto same = [&](const ss_params& a) {
                return a.suppress_blank == P0.suppress_blank && a.no_timestamps == P0.no_timestamps && a.tdrz_enable == P0.tdrz_enable &&
                       a.max_initial_ts == P0.max_initial_ts && (a.fixed_steps > 0) == (P0.fixed_steps > 0) &&
                       !a.suppress_non_speech_tokens == !P0.suppress_non_speech_tokens;
            };
            std::vector<RowRef> g;
            while (pos0 < all.size() && same(all[pos0].w->job->P)) g.push_back(all[pos0++]);
            round_rows(g, js, rule_consts(P0));
        }
    }

    void round_rows(std::vector<RowRef>& decs_in, std::vector<JobState>& js, const RuleConsts& rc) {
        const Vocab& vocab = hm.vocab;
        // expand decoders into rows (token, pos), position order within a decoder; split into launches of <= kPartRows rows
        std::vector<RowCtl> rows;
        std::vector<RowRef> refs;
        for (auto& dr : decs_in) {
            Window* w = dr.w; Dec& q = w->decs[dr.j];
            const int n_prompt = (int)w->prompt.size();
            const int n_feed = q.n_fed < n_prompt ? n_prompt - q.n_fed : 1;
            for (int k = 0; k < n_feed; k++) {
                RowCtl c{};
                const int p = q.n_fed + k;
                c.token = p < n_prompt ? w->prompt[p] : q.tokens.back().id;
                c.pos = p; c.slot = q.slot; c.cross = w->cross; c.n_keys = w->n_keys;
                const bool last = k == n_feed - 1;
                if (last) {
                    c.n_hist = (int)q.tokens.size();
                    c.last_ts = !q.tokens.empty() && q.tokens.back().id >= vocab.token_beg;
                    c.penult_ts = q.tokens.size() < 2 || q.tokens[q.tokens.size() - 2].id >= vocab.token_beg;
                    c.has_ts = q.has_ts; c.ts_min = q.seek_delta / 2;
                    if (rc.openai_ts) {   // the rule reads the token history, not the decoder state: last id >= beg sampled in this attempt
                        c.has_ts = 0; c.ts_min = 0;
                        for (const TokenData& t : q.tokens) if (t.id >= vocab.token_beg) { c.has_ts = 1; c.ts_min = t.id - vocab.token_beg; }
                    }
                    c.temperature = w->temperatures[w->it];
                    c.want_probs = c.temperature > 0.0f;
                }
                rows.push_back(c);
                refs.push_back({w, dr.j, last});
            }
            q.n_fed += n_feed;
            if (dr.j == 0) w->job->sess->n_decode++;
        }
        // ---- chained greedy steps -------------------------------------------------------------------------------------------------
        // A round in which every decoder feeds exactly one token and samples greedily is fully determined on the device: the pick kernel
        // of step t has already written the control blocks of step t+1.  Such a step is enqueued BEFORE the host waits for step t, so
        // the GPU never idles through the sample -> host -> upload -> launch turnaround.  The host still accepts every sample with the
        // same rules one step behind; decoders that end simply ignore their row of the step that was already in flight.
        const bool simple = chain_steps && rows.size() == decs_in.size() && rows.size() <= (size_t)kPartRows &&
                            std::all_of(rows.begin(), rows.end(), [](const RowCtl& c) { return c.temperature <= 0.0f; });
        auto may_continue = [&]() {   // is there a decoder that will still be running after the step that is in flight?
            for (auto& dr : decs_in) {
                const ss_params& P = dr.w->job->P;
                const int n_max = P.fixed_steps > 0 ? P.fixed_steps : n_tctx / 2 - 4;
                if (dr.w->decs[dr.j].i + 1 < n_max) return true;
            }
            return false;
        };
        auto launch_chained = [&]() {
            std::vector<int> sr(rows.size());
            for (size_t i = 0; i < sr.size(); i++) sr[i] = (int)i;
            decoder_step_fused((int)rows.size(), rc, sr, false, true);
            spec.valid = true; spec.parity = step_parity; spec.rc = rc;
            spec.decs.clear();
            for (auto& dr : decs_in) spec.decs.push_back({dr.w, dr.j});
        };
        if (spec.valid) {
            Spec cur = spec;
            spec.valid = false;
            std::vector<int> kmap;
            bool ok = simple && memcmp(&cur.rc, &rc, sizeof(RuleConsts)) == 0;
            for (size_t i = 0; ok && i < decs_in.size(); i++) {
                int k = -1;
                for (size_t t = 0; t < cur.decs.size(); t++) if (cur.decs[t].first == decs_in[i].w && cur.decs[t].second == decs_in[i].j) { k = (int)t; break; }
                if (k < 0) ok = false; else kmap.push_back(k);
            }
            if (ok) {   // this round IS the step already in flight
                if (decs_in.size() == cur.decs.size() && may_continue()) launch_chained();
                SS_HIP(hipEventSynchronize(ev_step[cur.parity]));
                for (size_t i = 0; i < decs_in.size(); i++)
                    accept_sample(*decs_in[i].w, decs_in[i].w->decs[decs_in[i].j], state_of(js, decs_in[i].w->job), samp_hb[cur.parity][kmap[i]], nullptr);
                return;
            }
            // not usable (sampled attempt, prompt rows, other rules): it finishes in stream order and is ignored
        }
        for (size_t r0 = 0; r0 < rows.size(); r0 += kPartRows) {
            const int M = (int)std::min<size_t>(kPartRows, rows.size() - r0);
            std::vector<int> samp_rows;
            bool any_probs = false;
            stage_acquire();   // this launch's own pinned block (ctl_h / rowidx_h / u_h)
            for (int m = 0; m < M; m++) {
                ctl_h[m] = rows[r0 + m];
                if (refs[r0 + m].sample) {
                    ctl_h[kPartRows + samp_rows.size()] = rows[r0 + m];
                    if (rows[r0 + m].want_probs) {
                        // whisper_sample_token: dist(decoder.rng) -- decoder 0 draws from the generator the state carries, decoder j >= 1 from its
                        // own (whisper.cpp >= 1.5.0); SS_COMPAT_RNG_STATE: dist(state.rng), one draw per sampled decoder in decoder order
                        any_probs = true;
                        Session* rs = refs[r0 + m].w->job->sess;
                        const int dj = refs[r0 + m].j;
                        CountingRng& g = (dj == 0 || (compat & SS_COMPAT_RNG_STATE)) ? rs->rng : rs->rng_dec.at((size_t)dj - 1);
                        u_h[samp_rows.size()] = std::generate_canonical<double, std::numeric_limits<double>::digits>(g);
                    }
                    samp_rows.push_back(m);
                }
            }
            const int par = decoder_step(M, rc, samp_rows, any_probs);
            stage_release();
            if (samp_rows.empty()) continue;
            if (simple && may_continue()) launch_chained();
            SS_HIP(hipEventSynchronize(ev_step[par]));
            const SampleOut* res = samp_hb[par];
            for (size_t k = 0; k < samp_rows.size(); k++) {
                const RowRef& rr = refs[r0 + samp_rows[k]];
                accept_sample(*rr.w, rr.w->decs[rr.j], state_of(js, rr.w->job), res[k], nullptr);
            }
        }
    }

    // whisper_sample_token + "update the decoder state" of whisper_full_with_state for one decoder
    void accept_sample(Window& w, Dec& q, JobState& jq, const SampleOut& so, const float* pr) {
        const Vocab& vocab = hm.vocab;
        const ss_params& P = w.job->P;
        Session* s = w.job->sess;
        const int n_max = P.fixed_steps > 0 ? P.fixed_steps : n_tctx / 2 - 4;
        const float t_cur = w.temperatures[w.it];
        TokenData tk;
        if (t_cur < 1e-6f) {
            tk.id = so.id; tk.tid = so.tid; tk.p = so.p; tk.plog = so.plog; tk.pt = so.pt; tk.ptsum = so.ptsum;
        } else {
            tk.id = so.id;    // the draw itself ran on the device (sample_draw_kernel) with the uniform this session's generator produced
            tk.p = so.p;
            tk.plog = tk.p > 0.0f ? logf(tk.p) : -INFINITY;
            tk.tid = so.tid; tk.pt = so.pt; tk.ptsum = so.ptsum;
            if (tk.id >= vocab.token_beg) { tk.tid = tk.id; tk.pt = tk.p; }
        }
        q.tokens.push_back(tk);
        w.trace.push_back(tk.id);
        q.sum_logprobs_all += tk.plog;
        const int i = q.i++;
        bool done = false;
        if (tk.id > vocab.token_beg) {
            const int seek_delta_new = 2 * (tk.id - vocab.token_beg);
            if (q.has_ts && q.seek_delta > seek_delta_new && q.result_len < i) { q.failed = true; done = true; }
            else { q.seek_delta = seek_delta_new; q.result_len = i + 1; q.has_ts = true; }
        }
        if (!done && P.fixed_steps > 0) {
            if (i == n_max - 1) { q.result_len = i + 1; q.seek_delta = 100 * kChunkSec; q.completed = true; done = true; }
        } else if (!done) {
            if (tk.id == vocab.token_eot || (P.max_tokens > 0 && i >= P.max_tokens) || (q.has_ts && jq.seek + q.seek_delta + 100 >= jq.seek_end)) {
                if (q.result_len == 0) {
                    if (jq.seek + q.seek_delta + 100 >= jq.seek_end) q.result_len = i + 1;
                    else { q.failed = true; done = true; }
                }
                if (!done) {
                    if (P.single_segment) { q.result_len = i + 1; q.seek_delta = 100 * kChunkSec; }
                    q.completed = true; done = true;
                }
            }
            if (!done && i == n_max - 1 && (q.result_len == 0 || q.seek_delta < 100 * kChunkSec / 2)) { q.failed = true; done = true; }
        }
        if (done || q.i >= n_max) q.active = false;
    }

    // whisper_wrap_segment (whisper.cpp v1.5.x; ss_params.max_len > 0, after the token-level times of a new segment): the LAST segment is cut wherever
    // the next text token would take the piece past max_len bytes -- with split_on_word only before a token that opens a word (' ' first) -- at that
    // token's t0.  Every piece but the last loses speaker_turn_next; ids >= eot carry no text and never start a piece unless they come first.
    static void wrap_last_segment(Session& s, const Vocab& vocab, int max_len, bool split_on_word) {
        size_t from = 0;           // first token of the piece being measured, as an index into the ORIGINAL token list
        const Segment whole = s.segments.back();
        const int n = (int)whole.tokens.size();
        int acc = 0;
        std::string text;
        for (int i = 0; i < n; i++) {
            const TokenData& tok = whole.tokens[i];
            if (tok.id >= vocab.token_eot) continue;
            const std::string& txt = vocab.id_to_token[tok.id];
            const int cur = (int)strlen(txt.c_str());
            const bool opens_word = !split_on_word || (!txt.empty() && txt[0] == ' ');
            if (acc + cur > max_len && i > (int)from && opens_word) {
                Segment& piece = s.segments.back();
                piece.text = text; piece.t1 = tok.t0; piece.speaker_turn_next = false;
                piece.tokens.assign(whole.tokens.begin() + from, whole.tokens.begin() + i);
                s.segments.push_back({tok.t0, whole.t1, std::string(), whole.speaker_turn_next, {}});
                s.segments.back().tokens.assign(whole.tokens.begin() + i, whole.tokens.end());
                from = i; acc = 0; text.clear();
            }
            acc += cur; text += txt;
        }
        s.segments.back().text = text;
    }

    void finalize_window(Window& w, std::vector<JobState>& js) {
        const Vocab& vocab = hm.vocab;
        const ss_params& P = w.job->P;
        Session* s = w.job->sess;
        JobState& jq = state_of(js, w.job);
        const Dec& bd = w.decs[w.best];
        const int seek = jq.seek, seek_delta = bd.seek_delta;
        const std::vector<TokenData>& tk = bd.tokens;
        for (auto& t : tk) s->tokens.push_back(t);
        s->sampled.insert(s->sampled.end(), bd.sampled.begin(), bd.sampled.end());
        s->trace.insert(s->trace.end(), w.trace.begin(), w.trace.end());
        // update prompt_past: the past context that was fed (without [prev] and the task tokens) + this window's text
        {
            std::vector<int> np;
            if (!w.prompt.empty() && w.prompt.front() == vocab.token_prev)
                np.insert(np.end(), w.prompt.begin() + 1, w.prompt.end() - jq.prompt_init.size());
            for (int i = 0; i < bd.result_len && i < (int)tk.size(); i++) np.push_back(tk[i].id);
            // SS_COMPAT_OPENAI_HISTORY: the closing timestamp of the window's last segment belongs to no segment's token slice (OpenAI / HF history)
            if ((compat & SS_COMPAT_OPENAI_HISTORY) && bd.result_len >= 2 && bd.result_len <= (int)tk.size() && tk[bd.result_len - 1].id >= vocab.token_beg &&
                tk[bd.result_len - 2].id >= vocab.token_beg)
                np.pop_back();
            s->prompt_past.swap(np);
        }
        if (!tk.empty()) {
            int64_t t0 = seek + 2 * (tk.front().tid - vocab.token_beg);
            std::string text;
            bool speaker_turn_next = false;
            int i0 = 0;   // first token of the segment being built ("for (int j = i0; j <= i; j++) result_all.back().tokens.push_back(tokens_cur[j])")
            for (int i = 0; i < (int)tk.size(); i++) {
                if (P.print_special || tk[i].id < vocab.token_eot) text += vocab.id_to_token[tk[i].id];
                if (P.tdrz_enable && tk[i].id == vocab.token_solm) speaker_turn_next = true;
                if (tk[i].id > vocab.token_beg && !P.single_segment) {
                    const int64_t t1 = seek + 2 * (tk[i].tid - vocab.token_beg);
                    if (!text.empty()) {
                        s->segments.push_back({t0, t1, text, speaker_turn_next, {}});
                        s->segments.back().tokens.assign(tk.begin() + i0, tk.begin() + i + 1);
                        if (P.token_timestamps) {
                            token_level_times(*s, vocab, s->segments.back(), jq.energy, jq.n_energy, P.thold_pt, P.thold_ptsum);
                            if (P.max_len > 0) wrap_last_segment(*s, vocab, P.max_len, P.split_on_word != 0);
                        }
                    }
                    text.clear();
                    while (i < (int)tk.size() && tk[i].id > vocab.token_beg) i++;
                    i--;
                    t0 = t1;
                    i0 = i + 1;
                    speaker_turn_next = false;
                }
            }
            if (!text.empty()) {
                s->segments.push_back({t0, (int64_t)(seek + seek_delta), text, speaker_turn_next, {}});
                s->segments.back().tokens.assign(tk.begin() + i0, tk.end());
                if (P.token_timestamps) {
                    token_level_times(*s, vocab, s->segments.back(), jq.energy, jq.n_energy, P.thold_pt, P.thold_ptsum);
                    if (P.max_len > 0) wrap_last_segment(*s, vocab, P.max_len, P.split_on_word != 0);
                }
            }
        }
        jq.seek += seek_delta;
    }

    // ------------------------------------------------------------------------------------------
    // stage hooks (host f32 in/out; same kernels as the batch path)
    // ------------------------------------------------------------------------------------------
    void log_mel_host(const float* pcm, int n, float* out, int n_len) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (n_len != mel_n_len(n)) throw Error(SS_ERR_ARG, "log_mel: n_len mismatch");
        pcm_d[0].ensure((size_t)std::max(n, 1) * 4);
        SS_HIP(hipMemcpyAsync(pcm_d[0].p, pcm, (size_t)n * 4, hipMemcpyHostToDevice, st));
        mel_d[0].ensure((size_t)n_mel * n_len * 4); fmax_d[0].ensure((size_t)n_len * 4);
        launch_log_mel(mt, pcm_d[0].as<float>(), n, mel_d[0].as<float>(), n_len, fmax_d[0].as<float>(), st);
        SS_HIP(hipMemcpyAsync(out, mel_d[0].p, (size_t)n_mel * n_len * 4, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
    }
    void signal_energy_host(const float* pcm, int n, float* out) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        pcm_d[0].ensure((size_t)n * 4);
        SS_HIP(hipStreamSynchronize(st));
        energy_h[0].ensure((size_t)n * 4);
        SS_HIP(hipMemcpyAsync(pcm_d[0].p, pcm, (size_t)n * 4, hipMemcpyHostToDevice, st));
        launch_signal_energy(pcm_d[0].as<float>(), n, (float*)energy_h[0].dev, st);
        SS_HIP(hipStreamSynchronize(st));
        memcpy(out, energy_h[0].p, (size_t)n * 4);
    }
    void encode_host(const float* mel, int n_len, int seek, float* enc_out, int audio_ctx = 0) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (audio_ctx > n_ctx) throw Error(SS_ERR_AUDIO_CTX, "encode: audio_ctx larger than the model's n_audio_ctx");
        if (audio_ctx < 0 || audio_ctx % 4) throw Error(SS_ERR_UNSUPPORTED, "encode: audio_ctx must be a positive multiple of 4 (or 0 = the model's n_audio_ctx)");
        set_context(audio_ctx > 0 ? audio_ctx : n_ctx);   // stage hooks run the full context unless told otherwise
        mel_d[0].ensure((size_t)n_mel * n_len * 4);
        SS_HIP(hipMemcpyAsync(mel_d[0].p, mel, (size_t)n_mel * n_len * 4, hipMemcpyHostToDevice, st));
        launch_mel_window<T>(mel_d[0].as<float>(), n_mel, n_len, seek, 2 * nc, x0.as<T>(), st);
        encoder_pass(1, true);
        SS_HIP(hipMemcpyAsync(enc_out, encF.p, (size_t)nc * da * 4, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
    }
    void fp8_first_quant_host(const float* mel, int n_len, int seek, uint8_t* codes, uint8_t* exps) override {
        if (!fp8_enc) throw Error(SS_ERR_UNSUPPORTED, "fp8_first_quant: the engine was not created with SS_DTYPE_FP8");
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        set_context(n_ctx);   // stage hooks always run the full context
        mel_d[0].ensure((size_t)n_mel * n_len * 4);
        SS_HIP(hipMemcpyAsync(mel_d[0].p, mel, (size_t)n_mel * n_len * 4, hipMemcpyHostToDevice, st));
        launch_mel_window<T>(mel_d[0].as<float>(), n_mel, n_len, seek, 2 * n_ctx, x0.as<T>(), st);
        std::vector<uint8_t> sc((size_t)(da / 64) * Mpad);
        tap8_codes = codes; tap8_sc = sc.data();
        // on the way out -- also when a later launch of encoder_pass throws -- the stream is drained first: the two D2H copies encoder_layers_f8
        // enqueued write into `sc` (this frame) and the caller's `codes`
        struct Clear { uint8_t*& a; uint8_t*& b; hipStream_t s; ~Clear() { (void)hipStreamSynchronize(s); a = nullptr; b = nullptr; } } clear{tap8_codes, tap8_sc, st};
        encoder_pass(1, false);
        SS_HIP(hipStreamSynchronize(st));
        for (int m = 0; m < n_ctx; m++)        // exponent bytes out of the GEMM's tile-aware order into [row][64-column block]
            for (int b = 0; b < da / 64; b++) exps[(size_t)m * (da / 64) + b] = sc[f8_scale_index(m, b, Mpad)];
    }
    int hook_n_keys = 0;      // stage hooks: key count of the encoder output last given to set_encoder_host (RowCtl.n_keys of decode_host's rows; 0 = n_ctx)
    // hook_owner (engine.h): the session whose set_encoder_host filled cross slot 0; nullptr once anything else touched lane 0's slot 0
    int hook_kv_len = 0;                // positions of self-KV slot 0 that hold hook_owner's history
    void set_encoder_host(const float* encv, int audio_ctx = 0, const void* owner = nullptr) override {
        std::lock_guard<std::mutex> lk(mu);
        hook_owner = nullptr; hook_kv_len = 0;
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (audio_ctx > n_ctx) throw Error(SS_ERR_AUDIO_CTX, "set_encoder: audio_ctx larger than the model's n_audio_ctx");
        if (audio_ctx < 0 || audio_ctx % 4) throw Error(SS_ERR_UNSUPPORTED, "set_encoder: audio_ctx must be a positive multiple of 4 (or 0 = the model's n_audio_ctx)");
        set_context(audio_ctx > 0 ? audio_ctx : n_ctx);
        hook_n_keys = nc == n_ctx ? 0 : nc;
        SS_HIP(hipMemcpyAsync(encF.p, encv, (size_t)nc * da * 4, hipMemcpyHostToDevice, st));
        launch_f32_to_T<T>(encF.as<float>(), encT.as<T>(), (size_t)nc * da, st);
        if (fp8_enc) launch_quantize_f8<T>(encT.as<T>(), da, ln8.as<unsigned char>(), ln_sc.as<unsigned char>(), Mpad, nc, da, st);
        cross_kv_pass(1);
        SS_HIP(hipStreamSynchronize(st));
        hook_owner = owner;
    }
    void decode_host(const int32_t* tokens, int n, int n_past, float* logits_out, const void* owner = nullptr) override {
        std::lock_guard<std::mutex> lk(mu);
        if (!owner || owner != hook_owner.load())
            throw Error(SS_ERR_ARG, "decode: this session's encoder output is no longer on the device (another session's set_encoder / transcription took the "
                                    "stage-hook slot): call set_encoder (whisper_encode) again and decode from n_past = 0");
        if (n_past > hook_kv_len)
            throw Error(SS_ERR_ARG, "decode: n_past " + std::to_string(n_past) + " is beyond the " + std::to_string(hook_kv_len) + " positions decoded since the last set_encoder");
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        ss_params P; ss_default_params(&P);
        const RuleConsts rc = rule_consts(P);
        for (int i = 0; i < n; i++) {
            RowCtl c{};
            c.token = tokens[i]; c.pos = n_past + i; c.slot = 0; c.cross = 0; c.n_hist = 1; c.n_keys = hook_n_keys;
            stage_acquire();
            ctl_h[0] = c; ctl_h[kPartRows] = c;
            std::vector<int> sr;
            if (i == n - 1) sr.push_back(0);
            decoder_step(1, rc, sr, false);
            stage_release();
            SS_HIP(hipStreamSynchronize(st));
        }
        SS_HIP(hipMemcpyAsync(logits_out, logits.p, (size_t)n_vocab * 4, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
        hook_kv_len = n_past + n;
    }
    // stage hook: cross-KV of one window from a given encoder output (cache slot `window` of lane 0)
    void set_encoder_window_host(const float* encv, int window) override {
        std::lock_guard<std::mutex> lk(mu);
        hook_owner = nullptr;
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        set_context(n_ctx);   // stage hooks always run the full context
        if (window < 0 || window >= B) throw Error(SS_ERR_ARG, "set_encoder_window: window outside the engine's batch");
        SS_HIP(hipMemcpyAsync(encF.p, encv, (size_t)n_ctx * da * 4, hipMemcpyHostToDevice, st));
        launch_f32_to_T<T>(encF.as<float>(), encT.as<T>(), (size_t)n_ctx * da, st);
        if (fp8_enc) launch_quantize_f8<T>(encT.as<T>(), da, ln8.as<unsigned char>(), ln_sc.as<unsigned char>(), Mpad, n_ctx, da, st);
        cross_kv_pass(1, &window);
        SS_HIP(hipStreamSynchronize(st));
    }
    // stage hook: ONE decoder launch over n rows (the pass the batched engine runs: rows of different windows and slots side by side); raw logits
    // of the rows listed in samp_rows.  The kernels are chosen exactly as in run_group (decoder_step): <= 16 rows the fused step, 17..64 the
    // multi-tile GEMVs, rows x heads >= direct_pairs the one-workgroup cross-attention.
    void decode_rows_host(const int32_t* token, const int32_t* pos, const int32_t* slot, const int32_t* crossw, int n, const int32_t* samp_rows, int n_samp,
                          float* logits_out) override {
        std::lock_guard<std::mutex> lk(mu);
        hook_owner = nullptr;
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (n < 1 || n > kPartRows || n_samp < 1 || n_samp > n) throw Error(SS_ERR_ARG, "decode_rows: 1..128 rows, 1..n sampling rows");
        for (int i = 0; i < n; i++)
            if (token[i] < 0 || token[i] >= n_vocab || pos[i] < 0 || pos[i] >= n_tctx || slot[i] < 0 || slot[i] >= S || crossw[i] < 0 || crossw[i] >= B)
                throw Error(SS_ERR_ARG, "decode_rows: row " + std::to_string(i) + " out of range");
        ss_params P; ss_default_params(&P);
        const RuleConsts rc = rule_consts(P);
        stage_acquire();
        for (int i = 0; i < n; i++) {
            RowCtl c{};
            c.token = token[i]; c.pos = pos[i]; c.slot = slot[i]; c.cross = crossw[i]; c.n_hist = 1;
            ctl_h[i] = c;
        }
        std::vector<int> sr;
        for (int k = 0; k < n_samp; k++) {
            if (samp_rows[k] < 0 || samp_rows[k] >= n) { stage_release(); throw Error(SS_ERR_ARG, "decode_rows: sampling row out of range"); }
            ctl_h[kPartRows + k] = ctl_h[samp_rows[k]];
            sr.push_back(samp_rows[k]);
        }
        decoder_step(n, rc, sr, false);
        stage_release();
        SS_HIP(hipMemcpy2DAsync(logits_out, (size_t)n_vocab * 4, logits.p, (size_t)n_vocab_pad * 4, (size_t)n_vocab * 4, n_samp, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
    }
    void process_logits_host(const float* raw, const int32_t* hist, int n_hist, int has_ts, int seek_delta, const ss_params& P, float out6[6],
                             float* logprobs_out) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        const Vocab& vocab = hm.vocab;
        SS_HIP(hipMemcpyAsync(logits.p, raw, (size_t)n_vocab * 4, hipMemcpyHostToDevice, st));
        RowCtl c{};
        c.n_hist = n_hist;
        c.last_ts = n_hist > 0 && hist[n_hist - 1] >= vocab.token_beg;
        c.penult_ts = n_hist < 2 || hist[n_hist - 2] >= vocab.token_beg;
        c.has_ts = has_ts; c.ts_min = seek_delta / 2; c.temperature = 0.0f;
        if (compat & SS_COMPAT_OPENAI_TS_RULES) {   // the rule reads the history itself (see round_rows)
            c.has_ts = 0; c.ts_min = 0;
            for (int i = 0; i < n_hist; i++) if (hist[i] >= vocab.token_beg) { c.has_ts = 1; c.ts_min = hist[i] - vocab.token_beg; }
        }
        stage_acquire();
        ctl_h[0] = c;
        SS_HIP(hipMemcpyAsync(ctl_d.p, ctl_h, sizeof(RowCtl), hipMemcpyHostToDevice, st));
        stage_release();
        launch_logits_rules(logits.as<float>(), n_vocab_pad, ctl_d.as<RowCtl>(), 1, rule_consts(P), samp_d.as<SampleOut>(), nullptr, rules_scratch.as<float>(), st);
        SS_HIP(hipMemcpyAsync(samp_h, samp_d.p, sizeof(SampleOut), hipMemcpyDeviceToHost, st));
        if (logprobs_out) {   // the processed row itself (tests hold every mask bit to the golden vectors)
            launch_logits_logprob_rows(logits.as<float>(), n_vocab_pad, ctl_d.as<RowCtl>(), 1, rule_consts(P), samp_d.as<SampleOut>(), probs.as<float>(), st);
            SS_HIP(hipMemcpyAsync(logprobs_out, probs.p, (size_t)n_vocab * 4, hipMemcpyDeviceToHost, st));
        }
        SS_HIP(hipStreamSynchronize(st));
        out6[0] = (float)samp_h[0].id; out6[1] = samp_h[0].p; out6[2] = samp_h[0].plog; out6[3] = (float)samp_h[0].tid;
        out6[4] = samp_h[0].pt; out6[5] = samp_h[0].ptsum;
    }
    // ------------------------------------------------------------------------------------------
    // STFT denoiser: denoise_audio of /root/reference/src/audio/mod.rs:507-523
    // ------------------------------------------------------------------------------------------
    DBuf dn_in, dn_mid, dn_out, dn_power, dn_noise, dn_signal, dn_var, dn_frames, dn_tw, pp_small, rs_sincs, rs_idx, rs_chunk;
    int rs_rate = 0;
    void dn_estimates(const float* x, int n, std::vector<float>* var_h) {
        const int n_chunks = n / 2048;
        dn_power.ensure((size_t)std::max(1, n_chunks) * 2048 * 4); dn_noise.ensure(2048 * 4); dn_signal.ensure(2048 * 4);
        dn_var.ensure((size_t)std::max(1, n_chunks) * 4);
        launch_dn_chunk_power(x, n_chunks, dn_tw.as<float2>(), dn_power.as<float>(), st);
        launch_dn_spectra(dn_power.as<float>(), n_chunks, dn_noise.as<float>(), dn_signal.as<float>(), dn_var.as<float>(), st);
        if (var_h) {
            var_h->assign(std::max(0, n_chunks - 1), 0.f);
            if (n_chunks > 1) SS_HIP(hipMemcpyAsync(var_h->data(), dn_var.p, (size_t)(n_chunks - 1) * 4, hipMemcpyDeviceToHost, st));
            SS_HIP(hipStreamSynchronize(st));
        }
    }
    void dn_pass(int mode, const float* x, int n, int step, float strength, float* y) {
        const int n_frames = (n - 2048) / step + 1;
        dn_frames.ensure((size_t)n_frames * 2048 * 4);
        launch_dn_frames(mode, x, n_frames, step, dn_tw.as<float2>(), dn_noise.as<float>(), dn_signal.as<float>(), strength, dn_frames.as<float>(), st);
        launch_dn_overlap_add(dn_frames.as<float>(), n_frames, step, n, y, st);
    }
    void denoise_host(const float* pcm, int n, const ss_denoise_config& cfg, int force_type, float* out, int* noise_type, float* norm_var, float* ms) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (cfg.frame_size != 2048) throw Error(SS_ERR_UNSUPPORTED, "denoise: only frame_size 2048 (the reference default) is implemented");
        if (n < 2048) throw Error(SS_ERR_ARG, "denoise: fewer samples than one frame (the reference panics in overlap_add)");
        const int step = (int)(2048.0f * (1.0f - cfg.overlap));
        if (step < 1 || step > 2048) throw Error(SS_ERR_ARG, "denoise: bad overlap");
        dn_twiddles();
        dn_in.ensure((size_t)n * 4); dn_mid.ensure((size_t)n * 4); dn_out.ensure((size_t)n * 4);
        SS_HIP(hipMemcpyAsync(dn_in.p, pcm, (size_t)n * 4, hipMemcpyHostToDevice, st));
        SS_HIP(hipEventRecord(ev[0], st));
        // analyze_noise_characteristics (mod.rs:533-579): sum over consecutive chunk pairs of mean squared spectral difference / n
        std::vector<float> var_h;
        dn_estimates(dn_in.as<float>(), n, &var_h);
        float sv = 0.f;
        for (float v : var_h) sv += v;
        const float nv = sv / (float)n;
        int nt = nv < 0.1f ? 0 : (nv > 0.5f ? 1 : 2);
        if (force_type >= 0 && force_type <= 2) nt = force_type;
        if (nt == 0) dn_pass(0, dn_in.as<float>(), n, step, cfg.strength, dn_out.as<float>());
        else if (nt == 1) dn_pass(1, dn_in.as<float>(), n, step, cfg.strength, dn_out.as<float>());
        else {
            dn_pass(0, dn_in.as<float>(), n, step, cfg.strength, dn_mid.as<float>());
            dn_estimates(dn_mid.as<float>(), n, nullptr);   // the Wiener stage estimates its spectra from ITS input
            dn_pass(1, dn_mid.as<float>(), n, step, cfg.strength, dn_out.as<float>());
        }
        SS_HIP(hipEventRecord(ev[1], st));
        SS_HIP(hipMemcpyAsync(out, dn_out.p, (size_t)n * 4, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
        if (noise_type) *noise_type = nt;
        if (norm_var) *norm_var = nv;
        if (ms) SS_HIP(hipEventElapsedTime(ms, ev[0], ev[1]));
    }

    void dn_twiddles() {
        if (dn_tw.p) return;
        std::vector<float> tw(2048);
        for (int k = 0; k < 1024; k++) { tw[2 * k] = (float)cos(-2.0 * M_PI * k / 2048.0); tw[2 * k + 1] = (float)sin(-2.0 * M_PI * k / 2048.0); }
        dn_tw.alloc(2048 * 4);
        SS_HIP(hipMemcpyAsync(dn_tw.p, tw.data(), 2048 * 4, hipMemcpyHostToDevice, st));
        SS_HIP(hipStreamSynchronize(st));
    }
    // StreamAudioProcessor over a whole stream (src/audio/mod.rs:67-155): see kernels_denoise.hip
    void preprocess_stream_host(const float* pcm, int64_t n, const int32_t* chunk_lens, int n_chunks, int chunk_len, const ss_denoise_config& cfg, float* out,
                                float* gains_out, float* ms) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (cfg.frame_size != 2048) throw Error(SS_ERR_UNSUPPORTED, "preprocess: only frame_size 2048 (the reference default) is implemented");
        dn_twiddles();
        std::vector<long> off;
        if (chunk_lens) {
            off.resize(n_chunks + 1);
            off[0] = 0;
            for (int i = 0; i < n_chunks; i++) {
                if (chunk_lens[i] <= 0) throw Error(SS_ERR_ARG, "preprocess: empty read chunk");
                off[i + 1] = off[i] + chunk_lens[i];
            }
            if (off[n_chunks] != n) throw Error(SS_ERR_ARG, "preprocess: chunk lengths do not add up to n");
        } else {
            if (chunk_len <= 0) throw Error(SS_ERR_ARG, "preprocess: chunk_len must be positive");
            n_chunks = (int)((n + chunk_len - 1) / chunk_len);
        }
        const int n_frames = (int)((n + 2047) / 2048);
        dn_in.ensure((size_t)n * 4); dn_mid.ensure((size_t)n * 4); dn_out.ensure((size_t)n_frames * 2048 * 4);
        pp_small.ensure((size_t)n_frames * 4 * 4 + (off.size() + 1) * 8);
        float* energy = pp_small.as<float>();
        float* sub = energy + n_frames;
        float* gain = sub + 2 * n_frames;
        long* d_off = nullptr;
        if (!off.empty()) {
            d_off = (long*)(gain + n_frames);   // 16 * n_frames bytes into an allocation aligned to 256: 8-byte aligned
            SS_HIP(hipMemcpyAsync(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice, st));
        }
        SS_HIP(hipMemcpyAsync(dn_in.p, pcm, (size_t)n * 4, hipMemcpyHostToDevice, st));
        SS_HIP(hipEventRecord(ev[0], st));
        launch_pp_stream(dn_in.as<float>(), (long)n, d_off, n_chunks, chunk_len, n_frames, dn_tw.as<float2>(), cfg.strength, cfg.noise_gate,
                         cfg.enable_noise_reduction, dn_mid.as<float>(), energy, sub, gain, dn_out.as<float>(), st);
        SS_HIP(hipEventRecord(ev[1], st));
        SS_HIP(hipMemcpyAsync(out, dn_out.p, (size_t)n_frames * 2048 * 4, hipMemcpyDeviceToHost, st));
        if (gains_out) SS_HIP(hipMemcpyAsync(gains_out, gain, (size_t)n_frames * 4, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
        if (ms) SS_HIP(hipEventElapsedTime(ms, ev[0], ev[1]));
    }

    // rubato SincFixedIn as create_resampler configures it (src/audio/mod.rs:235-250); see kernels_resample.hip
    static void rs_make_sincs(float f_cutoff, std::vector<float>& sincs) {
        const int npoints = 256, factor = 256, tot = npoints * factor;
        std::vector<float> y(tot);
        float sum = 0.0f;
        for (int x = 0; x < tot; x++) {
            const float x_pi = (float)x * (float)(2.0 * M_PI) / (float)tot;
            float w = 0.35875f - 0.48829f * cosf(x_pi) + 0.14128f * cosf(2.0f * x_pi) - 0.01168f * cosf(3.0f * x_pi);
            w = w * w;
            const float v = ((float)x - (float)(tot / 2)) * f_cutoff / (float)factor;
            const float a = v * (float)M_PI;
            const float sc = v == 0.0f ? 1.0f : sinf(a) / a;
            y[x] = w * sc;
            sum += y[x];
        }
        sum /= (float)factor;
        sincs.assign((size_t)tot, 0.f);
        for (int p = 0; p < npoints; p++)
            for (int n = 0; n < factor; n++) sincs[(size_t)(factor - n - 1) * npoints + p] = y[factor * p + n] / sum;
    }
    void resample_stream_host(const float* pcm, int64_t n, int from_rate, float* out, int64_t out_cap, int64_t* n_out, int32_t* chunk_lens, float* ms) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (from_rate <= 0 || from_rate == 16000) throw Error(SS_ERR_ARG, "resample: from_rate must be positive and different from 16000");
        const double ratio = 16000.0 / (double)from_rate;
        if (rs_rate != from_rate) {
            std::vector<float> sincs;
            rs_make_sincs(ratio >= 1.0 ? 0.95f : 0.95f * (float)ratio, sincs);
            rs_sincs.ensure(sincs.size() * 4);
            SS_HIP(hipMemcpyAsync(rs_sincs.p, sincs.data(), sincs.size() * 4, hipMemcpyHostToDevice, st));
            SS_HIP(hipStreamSynchronize(st));
            rs_rate = from_rate;
        }
        // the output instants: process_into_buffer's `while idx < end_idx { idx += t_ratio; ... }`, last_index carried between chunks
        const int chunk = 4096, sinc_len = 256;
        const int64_t n_chunks = n / chunk;
        const double t_ratio = 1.0 / ratio;
        const double end_idx = (double)(chunk - (sinc_len + 1) - (long)ceil(t_ratio));
        std::vector<double> idx_rel;
        std::vector<int> chunk_of;
        idx_rel.reserve((size_t)((double)n * ratio) + 16);
        chunk_of.reserve(idx_rel.capacity());
        double last_index = -(double)(sinc_len / 2);
        for (int64_t c = 0; c < n_chunks; c++) {
            double idx = last_index;
            int cnt = 0;
            while (idx < end_idx) {
                idx += t_ratio;
                idx_rel.push_back(idx);
                chunk_of.push_back((int)c);
                cnt++;
            }
            last_index = idx - (double)chunk;
            if (chunk_lens) chunk_lens[c] = cnt;
        }
        const int64_t total = (int64_t)idx_rel.size();
        if (n_out) *n_out = total;
        if (total > out_cap) throw Error(SS_ERR_ARG, "resample: output buffer too small (ss_resample_max_out)");
        if (total == 0) { if (ms) *ms = 0.f; return; }
        dn_in.ensure((size_t)n_chunks * chunk * 4); dn_out.ensure((size_t)total * 4);
        rs_idx.ensure((size_t)total * 8); rs_chunk.ensure((size_t)total * 4);
        SS_HIP(hipMemcpyAsync(dn_in.p, pcm, (size_t)n_chunks * chunk * 4, hipMemcpyHostToDevice, st));
        SS_HIP(hipMemcpyAsync(rs_idx.p, idx_rel.data(), (size_t)total * 8, hipMemcpyHostToDevice, st));
        SS_HIP(hipMemcpyAsync(rs_chunk.p, chunk_of.data(), (size_t)total * 4, hipMemcpyHostToDevice, st));
        SS_HIP(hipEventRecord(ev[0], st));
        launch_resample(dn_in.as<float>(), rs_idx.as<double>(), rs_chunk.as<int>(), (long)total, rs_sincs.as<float>(), dn_out.as<float>(), st);
        SS_HIP(hipEventRecord(ev[1], st));
        SS_HIP(hipMemcpyAsync(out, dn_out.p, (size_t)total * 4, hipMemcpyDeviceToHost, st));
        SS_HIP(hipStreamSynchronize(st));
        if (ms) SS_HIP(hipEventElapsedTime(ms, ev[0], ev[1]));
    }

    void selftest_gemm(int M, int N, int K, int kind, float* max_err, float* max_ref) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (M < 1 || N % 128 || K % 64 || N < 128 || K < 64) throw Error(SS_ERR_ARG, "selftest_gemm: N must be a multiple of 128 and K of 64");
        gemm_selftest<T>(M, N, K, kind, max_err, max_ref, st);
    }

    void selftest_gemm_ex(int M, int N, int K, int kind, int fp8, int reps, float* max_err, float* max_ref, float* avg_ms) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        if (avg_ms) *avg_ms = 0.f;
        if (fp8) {
            if (M < 1 || N < 256 || N % 256 || K < 256 || K % 256) throw Error(SS_ERR_ARG, "selftest_gemm (fp8): N and K must be multiples of 256");
            gemm_f8_selftest<T>(M, N, K, kind, max_err, max_ref, st, reps, avg_ms);
        } else {
            if (M < 1 || N % 128 || K % 64 || N < 128 || K < 64) throw Error(SS_ERR_ARG, "selftest_gemm: N must be a multiple of 128 and K of 64");
            gemm_selftest<T>(M, N, K, kind, max_err, max_ref, st, reps, avg_ms);
        }
    }

    void probe_gemm(int batch, int reps, float* avg_ms, double* flops) override {
        std::lock_guard<std::mutex> lk(mu);
        SS_HIP(hipSetDevice(opts.device));
        AllocStreamScope alloc_scope(st);
        set_context(n_ctx);
        if (batch < 1 || batch > B) throw Error(SS_ERR_ARG, "probe_gemm: batch out of range");
        const int M = batch * n_ctx;
        if (fp8_enc) {   // the same projection on the e4m3 kernel (GELU output quantised in the epilogue)
            GemmF8Desc g = gd8(ln8.p, ln_sc.p, da, enc[0].w18, enc[0].s1, M, 4 * da, da, F8_GELU_F8, enc[0].b1, ff8.p, 4 * da);
            g.out_scale = ff_sc.as<unsigned char>(); g.ld_osc = Mpad;
            for (int i = 0; i < reps; i++) launch_gemm_f8<T>(g, st);      // untimed: see below
            SS_HIP(hipEventRecord(ev[0], st));
            for (int i = 0; i < reps; i++) launch_gemm_f8<T>(g, st);
        } else {
            GemmDesc g = gd(ln.p, da, enc[0].w1, M, 4 * da, da, EPI_GELU_T, enc[0].b1, ff.p, 4 * da);
            // `reps` untimed launches first: the chip's clock needs tens of milliseconds of this load to settle (the same binary measures 889 TF/s over
            // its first 10 launches and 990 over 60, profiles/r06_e_yardstick_equal_load.txt); in the engine these GEMMs run back to back for the
            // whole encoder phase, so the settled rate is the one that describes them
            for (int i = 0; i < reps; i++) launch_gemm<T>(g, st);
            SS_HIP(hipEventRecord(ev[0], st));
            for (int i = 0; i < reps; i++) launch_gemm<T>(g, st);
        }
        SS_HIP(hipEventRecord(ev[1], st));
        SS_HIP(hipEventSynchronize(ev[1]));
        float ms = 0;
        SS_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
        *avg_ms = ms / reps;
        *flops = 2.0 * M * (4.0 * da) * da;
    }
};

// ------------------------------------------------------------------------------------------------
// async batch former
// ------------------------------------------------------------------------------------------------
void EngineBase::run_jobs_any(std::vector<Job*>& jobs) {
    const int n = n_lanes();
    for (int i = 0; i < n; i++) {
        EngineBase* l = lane(i);
        if (l->mu.try_lock()) {
            std::lock_guard<std::mutex> lk(l->mu, std::adopt_lock);
            l->run_jobs_locked(jobs);
            return;
        }
    }
    lane((int)(rr.fetch_add(1) % (unsigned)n))->run_jobs(jobs);
}

void EngineBase::run_jobs_parallel(std::vector<Job*>& jobs) {
    const size_t B = (size_t)(opts.max_batch > 0 ? opts.max_batch : 8);
    const int nl = n_lanes();
    if (jobs.size() <= B || nl == 1) { run_jobs_any(jobs); return; }
    std::vector<std::vector<Job*>> groups;
    for (size_t i = 0; i < jobs.size(); i += B) groups.emplace_back(jobs.begin() + i, jobs.begin() + std::min(jobs.size(), i + B));
    std::atomic<size_t> next{0};
    std::mutex emu;
    std::exception_ptr first;
    auto body = [&](int li) {
        for (size_t g; (g = next.fetch_add(1)) < groups.size();) {
            try { lane(li)->run_jobs(groups[g]); }
            catch (...) { std::lock_guard<std::mutex> lk(emu); if (!first) first = std::current_exception(); }
        }
    };
    std::vector<std::thread> th;
    const int nt = (int)std::min<size_t>(nl, groups.size());
    for (int i = 1; i < nt; i++) th.emplace_back(body, i);
    body(0);
    for (auto& t : th) t.join();
    if (first) std::rethrow_exception(first);
}

void EngineBase::job_finished(Job* j, EngineBase* lane_) {
    bool q = false;
    {
        std::lock_guard<std::mutex> lk(qmu);
        q = j->queued;
        if (q) {
            auto it = std::find(running.begin(), running.end(), std::make_pair(j, lane_));
            if (it != running.end()) running.erase(it);
            complete_locked(j);
        } else j->done.store(true, std::memory_order_release);
    }
    if (q) { donecv.notify_all(); qcv.notify_all(); }
}
// The one place a queued chunk becomes "done" (caller holds qmu).  Order matters: the session is released first (ss_session_free may delete it as
// soon as in_flight reads 0, and it waits under qmu), `done` last -- after it the waiter may delete the ticket, so `j` must not be touched again.
void EngineBase::complete_locked(Job* j) {
    load.fetch_sub(1);
    if (j->sess) j->sess->in_flight.fetch_sub(1);
    j->done.store(true, std::memory_order_release);
}
int EngineBase::admit_more(int n_max, EngineBase* lane_, std::vector<Job*>& out) {
    if (workers_free.load() > 0) return 0;   // a lane with nothing to do starts a chunk sooner than a running group reaches its next window start
    std::lock_guard<std::mutex> lk(qmu);
    if (stop) return 0;                      // the engine is going away: what is queued is failed by stop_worker, not started
    for (auto it = queue.begin(); it != queue.end() && (int)out.size() < n_max;) {
        bool dup = false;
        for (Job* b : out) if (b->sess == (*it)->sess) { dup = true; break; }
        if (!dup) for (auto& r : running) if (r.first->sess == (*it)->sess) { dup = true; break; }
        if (dup) { ++it; continue; }
        out.push_back(*it);
        running.push_back({*it, lane_});
        it = queue.erase(it);
    }
    return (int)out.size();
}

void EngineBase::start_worker() {
    const int n = n_lanes();
    for (int li = 0; li < n; li++) {
        workers.emplace_back([this, li] {
            EngineBase* L = lane(li);
            while (true) {
                std::vector<Job*> batch;
                workers_free.fetch_add(1);
                struct FreeGuard { std::atomic<int>& c; bool on = true; void off() { if (on) { c.fetch_sub(1); on = false; } } ~FreeGuard() { off(); } } free_guard{workers_free};
                {
                    // one worker at a time forms a batch (otherwise two idle workers would split a trickle of chunks between them); the
                    // others queue up behind form_mu and form the NEXT batch while this one runs
                    std::lock_guard<std::mutex> fl(form_mu);
                    std::unique_lock<std::mutex> lk(qmu);
                    qcv.wait(lk, [this] { return stop || !queue.empty(); });
                    if (stop && queue.empty()) return;
                    const int maxb = opts.max_batch > 0 ? opts.max_batch : 8;
                    if ((int)queue.size() < maxb && opts.batch_wait_us > 0)
                        qcv.wait_for(lk, std::chrono::microseconds(opts.batch_wait_us), [&] { return stop || (int)queue.size() >= maxb; });
                    // Level the lanes when the queue is short: with more than one batch but fewer than (idle lanes x max_batch) chunks queued,
                    // filling this lane to max_batch would leave the other idle lanes with little or nothing (64 chunks on three idle lanes:
                    // 32 / 32 / 0); take an even share instead (22 / 21 / 21).  Up to max_batch chunks stay together: a pass streams the decoder
                    // weights once whatever its row count, so splitting ONE batch over lanes buys nothing (measured: r04_a).  workers_free counts
                    // this worker and the ones queued up behind form_mu, i.e. the lanes with nothing to run.
                    const int idle = std::max(1, workers_free.load());
                    // chunks often arrive as a burst that is still being submitted when the first max_batch of them are here: with other lanes idle,
                    // linger while the queue keeps growing (300 us without a new chunk ends it; bounded by batch_wait_us) so that the burst is seen whole
                    if (idle > 1 && opts.batch_wait_us > 0) {
                        const auto t_stop = std::chrono::steady_clock::now() + std::chrono::microseconds(opts.batch_wait_us);
                        while (!stop && (int)queue.size() >= maxb && (int)queue.size() < idle * maxb && std::chrono::steady_clock::now() < t_stop) {
                            const size_t before = queue.size();
                            qcv.wait_for(lk, std::chrono::microseconds(300), [&] { return stop || queue.size() > before; });
                            if (queue.size() == before) break;
                        }
                    }
                    const int take = (int)queue.size() <= maxb ? maxb : std::min(maxb, ((int)queue.size() + idle - 1) / idle);
                    // one chunk per session per batch (run_group writes the session's results): a second ticket of a session already in this
                    // batch stays queued, in order, for the next one
                    for (auto it = queue.begin(); it != queue.end() && (int)batch.size() < take;) {
                        bool dup = false;
                        for (Job* b : batch) if (b->sess == (*it)->sess) { dup = true; break; }
                        if (!dup) for (auto& r : running) if (r.first->sess == (*it)->sess) { dup = true; break; }   // ... or still running on another lane
                        if (dup) { ++it; continue; }
                        batch.push_back(*it);
                        it = queue.erase(it);
                    }
                    for (Job* b : batch) running.push_back({b, L});
                    if (batch.empty()) {   // only tickets of busy sessions are queued: wait for a completion instead of spinning
                        donecv.wait_for(lk, std::chrono::milliseconds(2));
                        continue;
                    }
                }
                // nothing may escape this thread (std::terminate would take the host service down).  Chunks normally complete one by one from
                // inside the group (job_finished); whatever this lane still has registered afterwards -- the group threw -- is failed and released.
                free_guard.off();
                int fail_code = 0;
                std::string fail_what;
                {
                    std::lock_guard<std::mutex> lane_lk(L->mu);      // a blocking caller (ss_transcribe_batch) may be using this lane
                    L->from_queue = true;
                    try { L->run_jobs_locked(batch); }
                    catch (const Error& e) { fail_code = e.code; fail_what = e.what(); }
                    catch (const std::exception& e) { fail_code = SS_ERR_DEVICE; fail_what = e.what(); }
                    catch (...) { fail_code = SS_ERR_DEVICE; fail_what = "unknown exception in the batch former"; }
                    L->from_queue = false;
                }
                {
                    std::lock_guard<std::mutex> lk(qmu);
                    for (auto it = running.begin(); it != running.end();) {
                        if (it->second != L) { ++it; continue; }
                        Job* j = it->first;
                        if (j->status == 0) { j->status = fail_code ? fail_code : SS_ERR_DEVICE; j->err = fail_code ? fail_what : "chunk left unfinished by its device group"; }
                        it = running.erase(it);
                        complete_locked(j);
                    }
                }
                donecv.notify_all();
                qcv.notify_all();
            }
        });
    }
}
// Engine teardown with work outstanding (ss_engine_free): chunks still QUEUED fail with SS_ERR_DEVICE at once, chunks a lane is running finish
// normally (a group admits nothing more once `stop` is set), threads blocked in wait() are woken and have left before this returns.  Every
// ticket is `done` afterwards, so a later ss_wait returns its recorded status without touching the engine.
void EngineBase::stop_worker() {
    {
        std::lock_guard<std::mutex> lk(qmu);
        stop = true;
        for (Job* j : queue) {
            j->status = SS_ERR_DEVICE; j->err = "engine freed while the chunk was queued";
            complete_locked(j);
        }
        queue.clear();
    }
    qcv.notify_all(); donecv.notify_all();
    for (auto& w : workers) if (w.joinable()) w.join();
    workers.clear();
    std::unique_lock<std::mutex> lk(qmu);
    donecv.notify_all();
    donecv.wait(lk, [this] { return n_waiters == 0; });
}
void EngineBase::submit(Job* j) {
    j->queued = true;
    load.fetch_add(1);
    {
        std::lock_guard<std::mutex> lk(qmu);
        if (stop) {   // the engine is being freed: refuse instead of queueing behind workers that have gone
            j->status = SS_ERR_DEVICE; j->err = "engine is shutting down";
            complete_locked(j);
            return;
        }
        queue.push_back(j);
    }
    qcv.notify_all();
}
void EngineBase::wait(Job* j) {
    std::unique_lock<std::mutex> lk(qmu);
    n_waiters++;
    donecv.wait(lk, [&] { return j->done.load(std::memory_order_acquire); });
    if (--n_waiters == 0 && stop) donecv.notify_all();   // stop_worker waits for the last waiter to leave
}

void EngineBase::wait_session_idle(Session& s, std::unique_lock<std::mutex>& registry) {
    std::unique_lock<std::mutex> lk(qmu);
    n_waiters++;
    registry.unlock();
    donecv.wait(lk, [&] { return s.in_flight.load() == 0; });
    if (--n_waiters == 0 && stop) donecv.notify_all();
}
EngineBase* make_engine_bf16(const char* path, const ss_engine_opts& o) { return new EngineT<bf16>(path, o, nullptr); }
EngineBase* make_engine_f16(const char* path, const ss_engine_opts& o) { return new EngineT<f16>(path, o, nullptr); }

}  // namespace ss
