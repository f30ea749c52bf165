// LayerNorm, embedding, conversions, and the fused logits-rules + log-softmax + greedy-pick kernel.
// Replaces ggml's norm/mul/add, get_rows, cpy nodes (/root/reference/resources/ggml-metal.metal:571-621 kernel_norm,
// :3743-3855 kernel_get_rows, :1962-2126 kernel_cpy) and whisper.cpp's host-side whisper_process_logits +
// whisper_sample_token(best) loops over the vocabulary, which here never leave the GPU:
// only {id, p, plog, tid, pt, ptsum} per sequence cross PCIe each step.
#include "kernels.h"

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (d <= 2048), eps 1e-5
// ---------------------------------------------------------------------------------------------
template <typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                        TO* __restrict__ y, int rows, int d, const int* __restrict__ row_idx) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)(row_idx ? row_idx[row] : row) * d;
    f32x4 v[8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            v[i] = *(const f32x4*)(xr + c);
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
#pragma unroll
            for (int e = 0; e < 4; e++) { v[i][e] -= mean; sq += v[i][e] * v[i][e]; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float scale = 1.0f / sqrtf(sq / d + 1e-5f);
    TO* yr = y + (long)row * d;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int c = (i * 64 + lane) * 4;
        if (c < d) {
            const f32x4 ww = *(const f32x4*)(w + c), bb = *(const f32x4*)(b + c);
#pragma unroll
            for (int e = 0; e < 4; e++) yr[c + e] = (TO)(v[i][e] * scale * ww[e] + bb[e]);
        }
    }
}

template <typename T>
void launch_layernorm(const float* x, const float* w, const float* b, T* y, int rows, int d, const int* row_idx, hipStream_t st) {
    if (d > 2048 || d % 4) throw Error(-1, "layernorm: d must be <= 2048 and a multiple of 4");
    layernorm_kernel<T><<<(rows + 3) / 4, 256, 0, st>>>(x, w, b, y, rows, d, row_idx);
}
template void launch_layernorm<bf16>(const float*, const float*, const float*, bf16*, int, int, const int*, hipStream_t);
template void launch_layernorm<f16>(const float*, const float*, const float*, f16*, int, int, const int*, hipStream_t);
template <typename T>
void launch_layernorm_f32out(const float* x, const float* w, const float* b, float* y, int rows, int d, hipStream_t st) {
    if (d > 2048 || d % 4) throw Error(-1, "layernorm: d must be <= 2048 and a multiple of 4");
    layernorm_kernel<float><<<(rows + 3) / 4, 256, 0, st>>>(x, w, b, y, rows, d, nullptr);
}
template void launch_layernorm_f32out<bf16>(const float*, const float*, const float*, float*, int, int, hipStream_t);
template void launch_layernorm_f32out<f16>(const float*, const float*, const float*, float*, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// token + positional embedding
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_kernel(const T* __restrict__ te, const float* __restrict__ pe, const RowCtl* __restrict__ ctl, int d, float* __restrict__ x) {
    const int m = blockIdx.x;
    const RowCtl c = ctl[m];
    for (int i = threadIdx.x; i < d; i += blockDim.x) x[(long)m * d + i] = (float)te[(long)c.token * d + i] + pe[(long)c.pos * d + i];
}
template <typename T>
void launch_embed(const T* te, const float* pe, const RowCtl* ctl, int M, int d, float* x, hipStream_t st) {
    embed_kernel<T><<<M, 256, 0, st>>>(te, pe, ctl, d, x);
}
template void launch_embed<bf16>(const bf16*, const float*, const RowCtl*, int, int, float*, hipStream_t);
template void launch_embed<f16>(const f16*, const float*, const RowCtl*, int, int, float*, hipStream_t);

template <typename TI, typename TO>
__global__ void convert_kernel(const TI* __restrict__ in, TO* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (TO)(float)in[i];
}
template <typename T>
void launch_f32_to_T(const float* in, T* out, size_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    convert_kernel<float, T><<<blocks, 256, 0, st>>>(in, out, n);
}
template <typename T>
void launch_T_to_f32(const T* in, float* out, size_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    convert_kernel<T, float><<<blocks, 256, 0, st>>>(in, out, n);
}
template void launch_f32_to_T<bf16>(const float*, bf16*, size_t, hipStream_t);
template void launch_f32_to_T<f16>(const float*, f16*, size_t, hipStream_t);
template void launch_T_to_f32<bf16>(const bf16*, float*, size_t, hipStream_t);
template void launch_T_to_f32<f16>(const f16*, float*, size_t, hipStream_t);

// ---------------------------------------------------------------------------------------------
// whisper_process_logits + whisper_sample_token(best), one 1024-thread workgroup per sequence
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int kRuleThreads = 1024;
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
    if (c.last_ts) {
        if (c.penult_ts) { if (i >= rc.beg) kill = true; }
        else if (i < rc.eot) kill = true;
    }
    if (is_initial && rc.max_initial_tid >= 0 && i > rc.beg + rc.max_initial_tid) kill = true;
    if (c.has_ts && i >= rc.beg && i < rc.beg + c.ts_min) kill = true;
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

__global__ __launch_bounds__(kRuleThreads) void logits_rules_kernel(const float* __restrict__ logits, long ld, const RowCtl* __restrict__ ctl,
                                                                    RuleConsts rc, SampleOut* __restrict__ out, float* __restrict__ probs) {
    __shared__ float s_f[3][16];
    __shared__ MaxIdx s_mi[2][16];
    __shared__ float s_b[8];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RowCtl c = ctl[m];
    const float* raw = logits + (long)m * ld;
    const int n = rc.n_vocab;
    const int big = 0x7fffffff;

    // pass 1: maxima over all / text / timestamp tokens
    float mall = kNegInf, mtext = kNegInf, mts = kNegInf;
    for (int i = tid; i < n; i += kRuleThreads) {
        const float v = masked_logit(i, raw[i], c, rc);
        mall = fmaxf(mall, v);
        if (i < rc.beg) mtext = fmaxf(mtext, v); else mts = fmaxf(mts, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mall = fmaxf(mall, __shfl_xor(mall, o));
        mtext = fmaxf(mtext, __shfl_xor(mtext, o));
        mts = fmaxf(mts, __shfl_xor(mts, o));
    }
    if (lane == 0) { s_f[0][wave] = mall; s_f[1][wave] = mtext; s_f[2][wave] = mts; }
    __syncthreads();
    if (tid == 0) {
        float a = kNegInf, b = kNegInf, d = kNegInf;
        for (int w = 0; w < 16; w++) { a = fmaxf(a, s_f[0][w]); b = fmaxf(b, s_f[1][w]); d = fmaxf(d, s_f[2][w]); }
        s_b[0] = a; s_b[1] = b; s_b[2] = d;
    }
    __syncthreads();
    mall = s_b[0]; mtext = s_b[1]; mts = s_b[2];
    __syncthreads();
    // pass 2: sum exp over all (rel. mall) and over timestamps (rel. mts)
    float sall = 0.f, sts = 0.f;
    for (int i = tid; i < n; i += kRuleThreads) {
        const float v = masked_logit(i, raw[i], c, rc);
        if (v > kNegInf) {
            sall += expf(v - mall);
            if (i >= rc.beg) sts += expf(v - mts);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { sall += __shfl_xor(sall, o); sts += __shfl_xor(sts, o); }
    if (lane == 0) { s_f[0][wave] = sall; s_f[1][wave] = sts; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; w++) { a += s_f[0][w]; b += s_f[1][w]; }
        const float lse = logf(a) + mall;
        // timestamp_logprob = logsumexp over timestamp logprobs; logprob_max over ts = mts - lse
        const float ts_logprob = b > 0.0f ? logf(b) + (mts - lse) : kNegInf;
        const float max_text_logprob = mtext - lse;
        s_b[3] = lse;
        s_b[4] = ts_logprob > max_text_logprob ? 1.0f : 0.0f;
    }
    __syncthreads();
    const float lse = s_b[3];
    const bool force_ts = s_b[4] != 0.0f;
    // pass 3: greedy pick over probabilities (first max wins) + timestamp statistics
    MaxIdx best{-1.0f, big}, best_ts{0.0f, big};
    float sum_ts = 0.f;
    for (int i = tid; i < n; i += kRuleThreads) {
        float v = masked_logit(i, raw[i], c, rc);
        if (force_ts && i < rc.beg) v = kNegInf;
        const float p = v == kNegInf ? 0.0f : expf(v - lse);
        if (probs) probs[(long)m * ld + i] = p;
        if (p > best.v) { best.v = p; best.i = i; }
        if (i >= rc.beg) {
            sum_ts += p;
            if (p > best_ts.v) { best_ts.v = p; best_ts.i = i; }
        }
    }
    best = wave_max(best);
    best_ts = wave_max(best_ts);
    for (int o = 32; o > 0; o >>= 1) sum_ts += __shfl_xor(sum_ts, o);
    if (lane == 0) { s_mi[0][wave] = best; s_mi[1][wave] = best_ts; s_f[2][wave] = sum_ts; }
    __syncthreads();
    if (tid == 0) {
        MaxIdx a = s_mi[0][0], t = s_mi[1][0];
        float st = s_f[2][0];
        for (int w = 1; w < 16; w++) { a = better(a, s_mi[0][w]); t = better(t, s_mi[1][w]); st += s_f[2][w]; }
        SampleOut r;
        r.id = a.v > 0.0f ? a.i : 0;           // whisper_sample_token: id stays 0 if no prob exceeds 0
        r.p = a.v > 0.0f ? a.v : 0.0f;
        {
            float v = masked_logit(r.id, raw[r.id], c, rc);
            if (force_ts && r.id < rc.beg) v = kNegInf;
            r.plog = a.v > 0.0f ? v - lse : 0.0f;
        }
        r.tid = t.v > 0.0f ? t.i : 0;          // stays 0 if every timestamp prob is 0
        r.pt = t.v / (st + 1e-10f);
        r.ptsum = st;
        if (r.id >= rc.beg) { r.tid = r.id; r.pt = r.p; }
        r.pad[0] = force_ts; r.pad[1] = 0;
        out[m] = r;
    }
}
}  // namespace

void launch_logits_rules(const float* logits, long ld, const RowCtl* ctl, int M, const RuleConsts& rc, SampleOut* out, float* probs, hipStream_t st) {
    logits_rules_kernel<<<M, kRuleThreads, 0, st>>>(logits, ld, ctl, rc, out, probs);
}

}  // namespace ss
