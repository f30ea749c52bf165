// LayerNorm, embedding, conversions, and the fused logits-rules + log-softmax + greedy-pick kernel.
// Replaces ggml's norm/mul/add, get_rows, cpy nodes (/root/reference/resources/ggml-metal.metal:571-621 kernel_norm,
// :3743-3855 kernel_get_rows, :1962-2126 kernel_cpy) and whisper.cpp's host-side whisper_process_logits +
// whisper_sample_token(best) loops over the vocabulary, which here never leave the GPU:
// only {id, p, plog, tid, pt, ptsum} per sequence cross PCIe each step.
#include "kernels.h"
#include "wave_ops.h"

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (d <= 2048), eps 1e-5
// ---------------------------------------------------------------------------------------------
template <typename TO, int NI>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        TO* __restrict__ y, int rows, int d, const int* __restrict__ row_idx) {
    // NI float4 per lane (d <= NI*256); out-of-range lanes load a clamped address and are masked, so every load of the
    // row is issued back to back (a lane-dependent guard around the loads costs one memory round trip per group)
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)(row_idx ? row_idx[row] : row) * d;
    f32x4 v[NI];
    int cc[NI];
    bool ok[NI];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) { const int c = (i * 64 + lane) * 4; ok[i] = c < d; cc[i] = ok[i] ? c : 0; }
#pragma unroll
    for (int i = 0; i < NI; i++) v[i] = *(const f32x4*)(xr + cc[i]);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        if (!ok[i]) v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    sum = wave_sum(sum);
    const float mean = sum / d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
#pragma unroll
        for (int e = 0; e < 4; e++) { v[i][e] = ok[i] ? v[i][e] - mean : 0.f; sq += v[i][e] * v[i][e]; }
    }
    sq = wave_sum(sq);
    const float scale = 1.0f / sqrtf(sq / d + 1e-5f);
    TO* yr = y + (long)row * d;
    f32x4 ww[NI], bb[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) { ww[i] = *(const f32x4*)(w + cc[i]); bb[i] = *(const f32x4*)(b + cc[i]); }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        if (ok[i]) {
            typedef TO TO4 __attribute__((ext_vector_type(4)));
            TO4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = (TO)(v[i][e] * scale * ww[i][e] + bb[i][e]);
            *(TO4*)(yr + cc[i]) = o;
        }
    }
}

template <typename TO>
static void launch_ln_any(const float* x, const float* w, const float* b, TO* y, int rows, int d, const int* row_idx, hipStream_t st) {
    if (d > 2048 || d % 4) throw Error(-1, "layernorm: d must be <= 2048 and a multiple of 4");
    const int grid = (rows + 3) / 4;
    if (d <= 512) { layernorm_kernel<TO, 2><<<grid, 256, 0, st>>>(x, w, b, y, rows, d, row_idx); SS_LAUNCH_CHECK(); }
    else if (d <= 1280) { layernorm_kernel<TO, 5><<<grid, 256, 0, st>>>(x, w, b, y, rows, d, row_idx); SS_LAUNCH_CHECK(); }
    else { layernorm_kernel<TO, 8><<<grid, 256, 0, st>>>(x, w, b, y, rows, d, row_idx); SS_LAUNCH_CHECK(); }
}

template <typename T>
void launch_layernorm(const float* x, const float* w, const float* b, T* y, int rows, int d, const int* row_idx, hipStream_t st) {
    launch_ln_any<T>(x, w, b, y, rows, d, row_idx, st);
}
template void launch_layernorm<bf16>(const float*, const float*, const float*, bf16*, int, int, const int*, hipStream_t);
template void launch_layernorm<f16>(const float*, const float*, const float*, f16*, int, int, const int*, hipStream_t);
template <typename T>
void launch_layernorm_f32out(const float* x, const float* w, const float* b, float* y, int rows, int d, hipStream_t st) {
    launch_ln_any<float>(x, w, b, y, rows, d, nullptr, st);
}
template void launch_layernorm_f32out<bf16>(const float*, const float*, const float*, float*, int, int, hipStream_t);
template void launch_layernorm_f32out<f16>(const float*, const float*, const float*, float*, int, int, hipStream_t);


template <typename TI, typename TO>
__global__ void convert_kernel(const TI* __restrict__ in, TO* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (TO)(float)in[i];
}
template <typename T>
void launch_f32_to_T(const float* in, T* out, size_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    convert_kernel<float, T><<<blocks, 256, 0, st>>>(in, out, n); SS_LAUNCH_CHECK();
}
template <typename T>
void launch_T_to_f32(const T* in, float* out, size_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    convert_kernel<T, float><<<blocks, 256, 0, st>>>(in, out, n); SS_LAUNCH_CHECK();
}
template void launch_f32_to_T<bf16>(const float*, bf16*, size_t, hipStream_t);
template void launch_f32_to_T<f16>(const float*, f16*, size_t, hipStream_t);
template void launch_T_to_f32<bf16>(const bf16*, float*, size_t, hipStream_t);
template void launch_T_to_f32<f16>(const f16*, float*, size_t, hipStream_t);

// ---------------------------------------------------------------------------------------------
// whisper_process_logits + whisper_sample_token(best), one 1024-thread workgroup per sequence
// ---------------------------------------------------------------------------------------------
namespace {
constexpr float kNegInf = -__builtin_huge_valf();

__device__ __forceinline__ float masked_logit(int i, float v, const RowCtl& c, const RuleConsts& rc) {
    if (c.temperature > 0.0f) v /= c.temperature;
    const bool is_initial = c.n_hist == 0;
    bool kill = false;
    if (rc.suppress_blank && is_initial && (i == rc.eot || i == rc.blank)) kill = true;
    if (i == rc.not_ || i == rc.sot || i == rc.nosp || i == rc.translate || i == rc.transcribe || i == rc.prev) kill = true;
    if (!rc.tdrz_enable && i == rc.solm) kill = true;
    if (rc.no_timestamps && i >= rc.beg) kill = true;
    if (i > rc.sot && i <= rc.sot + rc.n_lang) kill = true;
    if (rc.suppress_eot && i == rc.eot) kill = true;
    if (rc.ns_mask && ((rc.ns_mask[i >> 5] >> (i & 31)) & 1u)) kill = true;
    if (c.last_ts) {
        if (c.penult_ts) { if (i >= rc.beg) kill = true; }
        else if (i < rc.eot) kill = true;
    }
    if (is_initial && rc.max_initial_tid >= 0 && i > rc.beg + rc.max_initial_tid) kill = true;
    if (rc.openai_ts) {   // OpenAI's ApplyTimestampRules: forced first timestamp; ids up to the last timestamp are gone, the last itself unless a pair is open
        if (is_initial && !rc.no_timestamps && i < rc.beg) kill = true;
        if (c.has_ts && i >= rc.beg && i < rc.beg + c.ts_min + ((c.last_ts && !c.penult_ts) ? 0 : 1)) kill = true;
    } else if (c.has_ts && i >= rc.beg && i < rc.beg + c.ts_min) kill = true;
    return kill ? kNegInf : v;
}

struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {  // larger value, then smaller index ("first max wins")
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ MaxIdx wave_max(MaxIdx a) {
    for (int o = 32; o > 0; o >>= 1) {
        MaxIdx b;
        b.v = __shfl_xor(a.v, o);
        b.i = __shfl_xor(a.i, o);
        a = better(a, b);
    }
    return a;
}

// Stage A: grid (kRuleSlices, M), 256 threads.  Each workgroup scans one slice of the vocabulary and emits
//   [0] max over all  [1] sum exp(v - max_all)  [2] max over text  [3] argmax text (as float bits)
//   [4] max over ts   [5] argmax ts             [6] sum_ts exp(v - max_ts)
// (a 1024-thread workgroup per row spent 73 us per step on 8 CUs; sliced, the same scan uses 64 x M workgroups)
constexpr int kRuleSlices = 64, kRuleRec = 8;

__global__ __launch_bounds__(256) void logits_rules_scan_kernel(const float* __restrict__ logits, long ld, const RowCtl* __restrict__ ctl, RuleConsts rc,
                                                                float* __restrict__ scratch) {
    __builtin_amdgcn_s_setprio(3);   // chain kernels outrank co-resident streaming waves (kernels_decode.hip SS_CHAIN_PRIO_STMT)
    __shared__ MaxIdx s_t[4], s_s[4];
    __shared__ float s_f[2][4];
    const int m = blockIdx.y, sl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RowCtl c = ctl[m];
    const float* raw = logits + (long)m * ld;
    const int n = rc.n_vocab, per = (n + kRuleSlices - 1) / kRuleSlices, i0 = sl * per, i1 = min(n, i0 + per);
    const int big = 0x7fffffff;
    MaxIdx mt{kNegInf, big}, ms{kNegInf, big};   // text / timestamp maxima with first-index tie-break
    float vals[4];
    // pass 1 (values kept in registers: <= 4 per thread for 16 slices x 256 threads over ~52k tokens)
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = i0 + u * 256 + tid;
        float v = kNegInf;
        if (i < i1) v = masked_logit(i, raw[i], c, rc);
        vals[u] = v;
        if (i < i1) {
            MaxIdx cand{v, i};
            if (i < rc.beg) mt = better(mt, cand); else ms = better(ms, cand);
        }
    }
    mt = wave_max(mt); ms = wave_max(ms);
    if (lane == 0) { s_t[wave] = mt; s_s[wave] = ms; }
    __syncthreads();
    mt = better(better(s_t[0], s_t[1]), better(s_t[2], s_t[3]));
    ms = better(better(s_s[0], s_s[1]), better(s_s[2], s_s[3]));
    const float mall = fmaxf(mt.v, ms.v);
    float sall = 0.f, sts = 0.f;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = i0 + u * 256 + tid;
        const float v = vals[u];
        if (i < i1 && v > kNegInf) {
            sall += expf(v - mall);
            if (i >= rc.beg) sts += expf(v - ms.v);
        }
    }
    sall = wave_sum(sall); sts = wave_sum(sts);
    if (lane == 0) { s_f[0][wave] = sall; s_f[1][wave] = sts; }
    __syncthreads();
    if (tid == 0) {
        float* o = scratch + ((long)m * kRuleSlices + sl) * kRuleRec;
        o[0] = mall; o[1] = (s_f[0][0] + s_f[0][1]) + (s_f[0][2] + s_f[0][3]);
        o[2] = mt.v; o[3] = __int_as_float(mt.i); o[4] = ms.v; o[5] = __int_as_float(ms.i);
        o[6] = (s_f[1][0] + s_f[1][1]) + (s_f[1][2] + s_f[1][3]);
    }
}

// Stage B: one wave per row combines the slices (fixed order) and applies whisper_sample_token(best)
__global__ __launch_bounds__(64) void logits_rules_pick_kernel(const float* __restrict__ scratch, RuleConsts rc, SampleOut* __restrict__ out,
                                                               RowCtl* __restrict__ ctl_upd, const int* __restrict__ row_of) {
    const int m = blockIdx.x;
    if (threadIdx.x != 0) return;
    __builtin_amdgcn_s_setprio(3);
    const float* s = scratch + (long)m * kRuleSlices * kRuleRec;
    const int big = 0x7fffffff;
    float mall = kNegInf;
    MaxIdx mt{kNegInf, big}, ms{kNegInf, big};
    for (int k = 0; k < kRuleSlices; k++) {
        const float* o = s + k * kRuleRec;
        mall = fmaxf(mall, o[0]);
        mt = better(mt, MaxIdx{o[2], __float_as_int(o[3])});
        ms = better(ms, MaxIdx{o[4], __float_as_int(o[5])});
    }
    float sall = 0.f, sts = 0.f;
    for (int k = 0; k < kRuleSlices; k++) {
        const float* o = s + k * kRuleRec;
        if (o[0] > kNegInf) sall += o[1] * expf(o[0] - mall);
        if (o[4] > kNegInf) sts += o[6] * expf(o[4] - ms.v);
    }
    const float lse = logf(sall) + mall;
    const float ts_logprob = sts > 0.0f ? logf(sts) + (ms.v - lse) : kNegInf;   // logsumexp over timestamp logprobs
    const float max_text_logprob = mt.v - lse;
    const bool force_ts = ts_logprob > max_text_logprob;                          // "sample timestamp" rule
    const MaxIdx best = force_ts ? ms : better(mt, ms);                           // first max wins
    SampleOut r;
    const float p_best = best.v > kNegInf ? expf(best.v - lse) : 0.0f;
    r.id = p_best > 0.0f ? best.i : 0;               // whisper_sample_token: id stays 0 if no prob exceeds 0
    r.p = p_best;
    r.plog = p_best > 0.0f ? best.v - lse : 0.0f;
    const float p_ts = ms.v > kNegInf ? expf(ms.v - lse) : 0.0f;
    const float sum_ts = sts > 0.0f ? expf(ts_logprob) : 0.0f;
    r.tid = p_ts > 0.0f ? ms.i : 0;                  // stays 0 if every timestamp prob is 0
    r.pt = p_ts / (sum_ts + 1e-10f);
    r.ptsum = sum_ts;
    if (r.id >= rc.beg) { r.tid = r.id; r.pt = r.p; }
    r.pad[0] = force_ts; r.pad[1] = __float_as_int(lse);
    out[m] = r;
    if (ctl_upd) {
        // the row's control block for the NEXT step, exactly as the host derives it from this sample when it is accepted greedily
        // (engine.cpp round_rows / accept_sample): lets step t+1 be enqueued before the host has seen step t
        RowCtl c = ctl_upd[kPartRows + m];
        const int nh = c.n_hist + 1;
        c.penult_ts = nh < 2 ? 1 : c.last_ts;
        c.last_ts = r.id >= rc.beg;
        c.n_hist = nh;
        c.token = r.id;
        c.pos += 1;
        if (rc.openai_ts ? r.id >= rc.beg : r.id > rc.beg) { c.has_ts = 1; c.ts_min = r.id - rc.beg; }
        ctl_upd[kPartRows + m] = c;
        ctl_upd[row_of[m]] = c;
    }
}

// t > 0 only: the full probability row for host-side sampling (probs = 0 where masked, text masked when a timestamp is forced)
// LOGP (stage hook ss_process_logits_row): whisper_process_logits' `logprobs` row instead -- -inf where a rule masks the id (the "timestamp mass
// beats every text token" rule included: it masks the text ids without renormalising), v - lse elsewhere
template <bool LOGP>
__global__ __launch_bounds__(256) void logits_probs_kernel(const float* __restrict__ logits, long ld, const RowCtl* __restrict__ ctl, RuleConsts rc,
                                                           const SampleOut* __restrict__ picked, float* __restrict__ probs) {
    const int m = blockIdx.y;
    const RowCtl c = ctl[m];
    const float lse = __int_as_float(picked[m].pad[1]);
    const bool force_ts = picked[m].pad[0] != 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rc.n_vocab; i += gridDim.x * 256) {
        float v = masked_logit(i, logits[(long)m * ld + i], c, rc);
        if (force_ts && i < rc.beg) v = kNegInf;
        if constexpr (LOGP) { probs[(long)m * ld + i] = v == kNegInf ? kNegInf : v - lse; continue; }
        probs[(long)m * ld + i] = v == kNegInf ? 0.0f : expf(v - lse);
    }
}
// t > 0: std::discrete_distribution's draw on the device.  libstdc++ normalises the weights by their sum, forms the running sums and returns
// lower_bound(cumulative, u) for u = generate_canonical<double, 53>(rng); the host draws u (so the generator is consumed exactly as
// whisper_sample_token consumes it) and this kernel finds the index: 256 contiguous segments summed in double, a scan of the 256 partial
// sums, then the owning segment is walked again.  Only the association of the double additions differs from the sequential scan (a draw
// would have to land within ~1e-16 of a boundary to notice); it replaces a 200 KB device-to-host copy and three passes over the
// vocabulary per decoder per step on the host.
__global__ __launch_bounds__(256) void sample_draw_kernel(const float* __restrict__ probs, long ld, int n_vocab, const RowCtl* __restrict__ ctl,
                                                          const double* __restrict__ u, SampleOut* __restrict__ out) {
    __shared__ double s_sum[256];
    __shared__ double s_pre[257];
    const int m = blockIdx.x, t = threadIdx.x;
    if (!ctl[m].want_probs) return;
    const float* p = probs + (long)m * ld;
    const int seg = (n_vocab + 255) / 256, b = t * seg, e = min(n_vocab, b + seg);
    double loc = 0.0;
    for (int i = b; i < e; i++) loc += (double)p[i];
    s_sum[t] = loc;
    __syncthreads();
    if (t == 0) {
        double run = 0.0;
        for (int i = 0; i < 256; i++) { s_pre[i] = run; run += s_sum[i]; }
        s_pre[256] = run;
    }
    __syncthreads();
    const double total = s_pre[256];
    const double target = u[m];
    const double lo = s_pre[t] / total, hi = t == 255 ? 1.0 : s_pre[t + 1] / total;   // the last cumulative value is forced to 1
    if (b < e && lo < target && target <= hi) {   // lower_bound: the first index whose cumulative value is >= u
        double run = s_pre[t];
        int id = e - 1;
        for (int i = b; i < e; i++) {
            run += (double)p[i];
            if (run / total >= target) { id = i; break; }
        }
        out[m].id = id;
        out[m].p = p[id];
    } else if (t == 0 && !(target > 0.0)) {       // u == 0: the first element
        out[m].id = 0;
        out[m].p = p[0];
    }
}
}  // namespace

void launch_sample_draw(const float* probs, long ld, int n_vocab, const RowCtl* ctl, int M, const double* u, SampleOut* out, hipStream_t st) {
    sample_draw_kernel<<<M, 256, 0, st>>>(probs, ld, n_vocab, ctl, u, out); SS_LAUNCH_CHECK();
}

void launch_logits_rules(const float* logits, long ld, const RowCtl* ctl, int M, const RuleConsts& rc, SampleOut* out, float* probs, float* scratch,
                         hipStream_t st, RowCtl* ctl_upd, const int* row_of) {
    if ((rc.n_vocab + kRuleSlices - 1) / kRuleSlices > 4 * 256) throw Error(-1, "logits rules: vocabulary too large for the slice plan");
    logits_rules_scan_kernel<<<dim3(kRuleSlices, M), 256, 0, st>>>(logits, ld, ctl, rc, scratch); SS_LAUNCH_CHECK();
    logits_rules_pick_kernel<<<M, 64, 0, st>>>(scratch, rc, out, ctl_upd, row_of); SS_LAUNCH_CHECK();
    if (probs) { logits_probs_kernel<false><<<dim3(32, M), 256, 0, st>>>(logits, ld, ctl, rc, out, probs); SS_LAUNCH_CHECK(); }
}
void launch_logits_logprob_rows(const float* logits, long ld, const RowCtl* ctl, int M, const RuleConsts& rc, const SampleOut* picked, float* logprobs, hipStream_t st) {
    logits_probs_kernel<true><<<dim3(32, M), 256, 0, st>>>(logits, ld, ctl, rc, picked, logprobs); SS_LAUNCH_CHECK();
}

}  // namespace ss
