// Fused decode-step GEMVs for gfx950 (M <= 16 rows: one token for each of up to 16 sequences).
//
// A decoder step is HBM-bound (1.8 GB of weights + B x 246 MB of cross-KV per step for large-v3) but, launched as one
// small kernel per graph node the way ggml does (/root/reference/resources/ggml-metal.metal:1307-1363 kernel_mul_mv_f16_f32,
// :571-621 kernel_norm, :54-151 add/mul), it is launch/latency-bound: ~350 dependent launches per step.  Here every
// projection is ONE launch that fills the chip and carries its neighbours with it:
//   prologue : residual add + bias of the previous projection + deterministic reduction of its split-K partials
//              (+ token/positional embedding for layer 0) -> LayerNorm -> f16/bf16 operand tile in LDS;
//              or the flash-decoding combine of the cross-attention partials
//   body     : weight fragments are prefetched into VGPRs BEFORE the prologue (both HBM latencies overlap), then
//              16x16x32 MFMAs with the 16-row weight tile as A operand and the (<=16) token rows as B operand
//   epilogue : q/k/v scaling + KV-cache append, GELU, logits, or raw split-K partials for the next prologue
// Split-K goes across workgroups (grid = N/16 x S) so that N = d projections still launch >= 256 workgroups; partials
// are summed in a fixed order by the consumer, so results are run-to-run identical (no float atomics).
#include "kernels.h"
#include "wave_ops.h"

// streamed-once operands (decoder weights, cross K/V).  Non-temporal loads (MI355X_MICROARCH.md "nt-weights") measured 1.2 % SLOWER here
// (A/B/A/B on one box: 2.28 vs 2.25 ms per step), so plain loads are the default; -DSS_NT builds the nt variant.
#ifdef SS_NT
#define SS_LDW(p) __builtin_nontemporal_load(p)
#else
#define SS_LDW(p) (*(p))
#endif


namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct MfmaD;
template <> struct MfmaD<bf16> {
    typedef bf16x8 V8;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaD<f16> {
    typedef f16x8 V8;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ float gelu_tanh_d(float x) {
    const float u = 0.79788456080286535588f * x * (1.0f + 0.044715f * x * x);
    const float e = __expf(2.0f * u);
    return 0.5f * x * (1.0f + (1.0f - 2.0f / (e + 1.0f)));
}
template <typename T> __device__ __forceinline__ float gelu_in_round_d(float x, int on);
template <> __device__ __forceinline__ float gelu_in_round_d<bf16>(float x, int) { return x; }
template <> __device__ __forceinline__ float gelu_in_round_d<f16>(float x, int on) { return on ? (float)(f16)x : x; }

constexpr int kMaxFrag = 10;      // <= 320 k per wave
constexpr int kXsPad = 8;         // LDS row padding (elements)
constexpr int kCrossSplitD = 4, kCrossPartD = 66;

// x = [x_in | tok+pos embedding] + bias_prev + sum_p parts[p]  (fixed order, branch-free: up to 4 partial slots, unused
// slots re-read slot 0 with weight 0 so that every load is independent and in flight together), optional write-back,
// LayerNorm over the full row, normalised columns [kbeg, kbeg+kslice) written to dst as T.  One wave per row.
template <typename T, int NI, bool PLAIN = false>   // PLAIN: x = x_in only (no embedding, partials, bias or write-back): the fused-prologue form
__device__ __forceinline__ void ln_row(const DecGemvDesc& g, int m, int lane, bool write_x, int kbeg, int kslice, T* dst) {
    // NI float4 per lane cover the row (d <= NI*256); lanes past the end load a clamped (valid) address and are masked,
    // so no load sits behind a divergent branch: all of them are in flight together.
    const int d = g.K;
    const int r = g.row_idx ? g.row_idx[m] : m;
    f32x4 v[NI];
    int cc[NI];
    bool ok[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) { const int c = (i * 64 + lane) * 4; ok[i] = c < d; cc[i] = ok[i] ? c : 0; }
    f32x4 ww[NI], bb[NI];   // issued with the row loads: the LayerNorm affine must not cost its own memory round trip
#pragma unroll
    for (int i = 0; i < NI; i++) { ww[i] = *(const f32x4*)(g.ln_w + cc[i]); bb[i] = *(const f32x4*)(g.ln_b + cc[i]); }
    if constexpr (PLAIN) {
        const float* xr = g.x_in + (long)r * d;
#pragma unroll
        for (int i = 0; i < NI; i++) v[i] = *(const f32x4*)(xr + cc[i]);
    } else if (g.ctl) {  // layer 0: token + positional embedding (replaces ggml get_rows + add)
        const RowCtl rc = g.ctl[r];
        const T* te = (const T*)g.tok_emb + (long)rc.token * d;
        const float* pe = g.pos_emb + (long)rc.pos * d;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const f32x4 p4 = *(const f32x4*)(pe + cc[i]);
            const T* t = te + cc[i];
            v[i] = (f32x4){(float)t[0] + p4[0], (float)t[1] + p4[1], (float)t[2] + p4[2], (float)t[3] + p4[3]};
        }
    } else {
        const float* xr = g.x_in + (long)r * d;
        const float wgt[4] = {g.n_parts > 0 ? 1.f : 0.f, g.n_parts > 1 ? 1.f : 0.f, g.n_parts > 2 ? 1.f : 0.f, g.n_parts > 3 ? 1.f : 0.f};
        const float* pp[4];
#pragma unroll
        for (int p = 0; p < 4; p++) pp[p] = g.n_parts > 0 ? g.parts + ((long)(p < g.n_parts ? p : 0) * kPartRows + r) * d : xr;
        const float bw = g.bias_prev ? 1.f : 0.f;
        const float* bp = g.bias_prev ? g.bias_prev : xr;
        f32x4 a[NI], b[NI], q0[NI], q1[NI];
#pragma unroll
        for (int i = 0; i < NI; i++) { a[i] = *(const f32x4*)(xr + cc[i]); b[i] = *(const f32x4*)(bp + cc[i]); }
#pragma unroll
        for (int i = 0; i < NI; i++) { q0[i] = *(const f32x4*)(pp[0] + cc[i]); q1[i] = *(const f32x4*)(pp[1] + cc[i]); }
#pragma unroll
        for (int i = 0; i < NI; i++) v[i] = ((a[i] + b[i] * bw) + q0[i] * wgt[0]) + q1[i] * wgt[1];
#pragma unroll
        for (int i = 0; i < NI; i++) { q0[i] = *(const f32x4*)(pp[2] + cc[i]); q1[i] = *(const f32x4*)(pp[3] + cc[i]); }
#pragma unroll
        for (int i = 0; i < NI; i++) v[i] = (v[i] + q0[i] * wgt[2]) + q1[i] * wgt[3];
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NI; i++) {
        if (!ok[i]) v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    if constexpr (!PLAIN) {
        if (write_x && g.x_out) {
#pragma unroll
            for (int i = 0; i < NI; i++) if (ok[i]) *(f32x4*)(g.x_out + (long)r * d + cc[i]) = v[i];
        }
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
    const float rstd = 1.0f / sqrtf(sq / d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NI; i++) {
        const int c = cc[i];
        if (ok[i] && c >= kbeg && c < kbeg + kslice) {
#pragma unroll
            for (int e = 0; e < 4; e++) dst[(c - kbeg) + e] = (T)(v[i][e] * rstd * ww[i][e] + bb[i][e]);
        }
    }
}

// stand-alone form (one wave per row): used in front of GEMVs with thousands of workgroups (logits), where a per-workgroup
// prologue would repeat the reduction too often
template <typename T, int NI>
__global__ __launch_bounds__(64) void dec_reduce_ln_kernel(DecGemvDesc g, T* out) {
    ln_row<T, NI>(g, blockIdx.x, threadIdx.x, true, 0, g.K, out + (long)blockIdx.x * g.K);
}

// one output (token row m, output column n, whole-slice sum v) of a decode GEMV
template <typename T, int EPI>
__device__ __forceinline__ void dec_epilogue(const DecGemvDesc& g, int s, int m, int n, float v) {
    if constexpr (EPI == DEPI_PART) {
        g.part_out[((long)s * kPartRows + m) * g.N + n] = v;
    } else if constexpr (EPI == DEPI_RES) {   // residual stream: x_out = x_in + bias + W a  (S == 1: the whole K sum is here)
        g.x_out[(long)m * g.N + n] = (g.x_in[(long)m * g.N + n] + g.bias[n]) + v;
    } else {
        if (g.bias) v += g.bias[n];
        if constexpr (EPI == DEPI_GELU_T) {
            ((T*)g.out)[(long)m * g.ldo + n] = (T)gelu_tanh_d(gelu_in_round_d<T>(v, g.gelu_f16_in));
        } else if constexpr (EPI == DEPI_LOGITS) {
            if (n < g.n_valid) ((float*)g.out)[(long)m * g.ldo + n] = v;
        } else if constexpr (EPI == DEPI_QKV) {
            const int d = g.d;
            if (n < d) ((T*)g.out)[(long)m * g.ldo + n] = (T)(v * g.scale);
            else {
                const RowCtl c = g.ctl_rows[m];
                const long off = (long)c.slot * g.slot_stride + (long)c.pos * d;
                if (n < 2 * d) ((T*)g.kcache)[off + (n - d)] = (T)(v * g.scale);
                else ((T*)g.vcache)[off + (n - 2 * d)] = (T)v;
            }
        }
    }
}

// NT = g.NT output columns per workgroup (1..16; the MFMA tile is 16 wide, rows >= NT are never loaded nor stored).  Narrow tiles are how a
// projection with few outputs (N = d) still fills the chip WITHOUT split-K: N / NT >= 256 workgroups each own the whole K sum, so the epilogue
// can be the real one (residual update, q scaling, GELU) and no partials / reduce launch exists.  MAXT = 1024 carries K = 4d in 16 waves.
template <typename T, int PRO, int EPI, int NI, int MAXT>
__global__ __launch_bounds__(MAXT) void dec_gemv_kernel(DecGemvDesc g) {
    typedef typename MfmaD<T>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int NT = g.NT;
    const int n0 = blockIdx.x * NT, s = blockIdx.y;
    const int kslice = g.K / g.S, kbeg = s * kslice, kw = kslice / NW, kwb = wave * kw;   // kw % 32 == 0, kw <= 320
    const int nfr = kw / 32, npair = nfr / 2;
    T* xs = (T*)smem_d;                                   // [16][kslice + pad]
    const int xld = kslice + kXsPad;
    float* red = (float*)(smem_d + (PRO == PRO_T ? (size_t)0 : (size_t)16 * xld * sizeof(T)));  // [NW][16][17]

    // ---- weight prefetch: lane loads 32 contiguous bytes of its row per MFMA pair ----
    const bool wrow = frow < NT && n0 + frow < g.N;          // this lane's weight row exists
    const T* wp = (const T*)g.W + (long)(wrow ? n0 + frow : 0) * g.K + kbeg + kwb;
    V8 wf[kMaxFrag];
#pragma unroll
    for (int j = 0; j < kMaxFrag / 2; j++) {
        wf[2 * j] = V8{}; wf[2 * j + 1] = V8{};
        if (j < npair && wrow) {
            wf[2 * j] = SS_LDW((const V8*)(wp + j * 64 + fg * 16));
            wf[2 * j + 1] = SS_LDW((const V8*)(wp + j * 64 + fg * 16 + 8));
        }
    }
    V8 wtail = {};
    if ((nfr & 1) && wrow) wtail = SS_LDW((const V8*)(wp + npair * 64 + fg * 8));

    // ---- prologue: build the operand tile xs[m][0..kslice) ----
    if constexpr (PRO == PRO_LN) {
        for (int m = wave; m < 16; m += NW) {
            if (m >= g.M) {
                for (int c = lane; c < kslice; c += 64) xs[m * xld + c] = (T)0.0f;
                continue;
            }
            if (g.n_parts == 0 && !g.bias_prev && !g.ctl) ln_row<T, NI, true>(g, m, lane, false, kbeg, kslice, xs + m * xld);
            else ln_row<T, NI>(g, m, lane, blockIdx.x == 0 && s == 0, kbeg, kslice, xs + m * xld);
        }
    } else if constexpr (PRO == PRO_COMBINE) {
        // flash-decoding combine of the cross-attention partials for the columns of this K slice
        const int H = g.K / 64, h0 = kbeg >> 6, hs = kslice >> 6;   // kslice is a multiple of 64 (checked on the host)
        float* wtab = red;                                           // [16][hs][5] split weights + denominator (red is reused after the sync)
        for (int idx = tid; idx < g.M * hs; idx += blockDim.x) {
            const int m = idx / hs, hh = idx % hs;
            const float* part = g.cross_parts + (long)(m * H + h0 + hh) * kCrossSplitD * kCrossPartD;
            const float m0 = part[0], m1 = part[kCrossPartD], m2 = part[2 * kCrossPartD], m3 = part[3 * kCrossPartD];
            const float l0 = part[1], l1 = part[kCrossPartD + 1], l2 = part[2 * kCrossPartD + 1], l3 = part[3 * kCrossPartD + 1];
            const float mx = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
            const float w0 = __expf(m0 - mx), w1 = __expf(m1 - mx), w2 = __expf(m2 - mx), w3 = __expf(m3 - mx);
            float den = 0.f;                     // same order as dec_cross_combine_kernel: num / den
            den += w0 * l0; den += w1 * l1; den += w2 * l2; den += w3 * l3;
            float* wt = wtab + idx * 5;
            wt[0] = w0; wt[1] = w1; wt[2] = w2; wt[3] = w3; wt[4] = den;
        }
        for (int idx = tid + g.M * kslice; idx < 16 * kslice; idx += blockDim.x) xs[(idx / kslice) * xld + idx % kslice] = (T)0.0f;
        __syncthreads();
#pragma unroll 4
        for (int idx = tid; idx < g.M * kslice; idx += blockDim.x) {
            const int m = idx / kslice, cc = idx % kslice, hh = cc >> 6, j = cc & 63;
            const float* part = g.cross_parts + (long)(m * H + h0 + hh) * kCrossSplitD * kCrossPartD + 2 + j;
            const float* wt = wtab + (m * hs + hh) * 5;
            float num = 0.f;
            num += wt[0] * part[0]; num += wt[1] * part[kCrossPartD]; num += wt[2] * part[2 * kCrossPartD]; num += wt[3] * part[3 * kCrossPartD];
            xs[m * xld + cc] = (T)(num / wt[4]);
        }
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PRO == PRO_T) {
        // activations: B fragments straight from L2 into VGPRs, issued together with the weight prefetch (no LDS staging,
        // no barrier before the MFMAs).  Token rows >= M read row 0: MFMA columns are independent and never stored.
        const T* xg = (const T*)g.Xt + (long)(frow < g.M ? frow : 0) * g.ldx + kbeg + kwb;
        V8 xf[kMaxFrag];
#pragma unroll
        for (int j = 0; j < kMaxFrag / 2; j++) {
            if (j < npair) {
                xf[2 * j] = *(const V8*)(xg + j * 64 + fg * 16);
                xf[2 * j + 1] = *(const V8*)(xg + j * 64 + fg * 16 + 8);
            }
        }
        V8 xtail = {};
        if (nfr & 1) xtail = *(const V8*)(xg + npair * 64 + fg * 8);
#pragma unroll
        for (int j = 0; j < kMaxFrag / 2; j++) {
            if (j < npair) {
                acc = MfmaD<T>::mma(wf[2 * j], xf[2 * j], acc);
                acc = MfmaD<T>::mma(wf[2 * j + 1], xf[2 * j + 1], acc);
            }
        }
        if (nfr & 1) acc = MfmaD<T>::mma(wtail, xtail, acc);
    } else {
        __syncthreads();
        const T* xr = xs + frow * xld + kwb;
#pragma unroll
        for (int j = 0; j < kMaxFrag / 2; j++) {
            if (j < npair) {
                const V8 x0 = *(const V8*)(xr + j * 64 + fg * 16), x1 = *(const V8*)(xr + j * 64 + fg * 16 + 8);
                acc = MfmaD<T>::mma(wf[2 * j], x0, acc);
                acc = MfmaD<T>::mma(wf[2 * j + 1], x1, acc);
            }
        }
        if (nfr & 1) {
            const V8 x0 = *(const V8*)(xr + npair * 64 + fg * 8);
            acc = MfmaD<T>::mma(wtail, x0, acc);
        }
    }
    // D[n][m]: lane holds n = fg*4 + r, m = frow
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 16 + frow) * 17 + fg * 4 + r] = acc[r];
    __syncthreads();

    // ---- epilogue: 16 m x 16 n outputs ----
    for (int idx = tid; idx < 256; idx += blockDim.x) {
        const int m = idx >> 4, nn = idx & 15, n = n0 + nn;
        if (m < g.M && nn < NT && n < g.N) {
            float v = 0.f;
            for (int w = 0; w < NW; w++) v += red[(w * 16 + m) * 17 + nn];
            dec_epilogue<T, EPI>(g, s, m, n, v);
        }
    }
}

template <typename T, int PRO, int EPI, int NI, int MAXT>
static void launch_dg3(const DecGemvDesc& g, int NW, hipStream_t st) {
    const int kslice = g.K / g.S;
    size_t red_f = (size_t)NW * 16 * 17, wtab_f = (size_t)16 * (kslice / 64 + 1) * 5;
    const size_t lds = (PRO == PRO_T ? 0 : (size_t)16 * (kslice + kXsPad) * sizeof(T)) + (red_f > wtab_f ? red_f : wtab_f) * 4;
    static std::atomic<uint64_t> attr{0};
    once_per_device(attr, [] { SS_HIP(hipFuncSetAttribute((const void*)dec_gemv_kernel<T, PRO, EPI, NI, MAXT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); });
    dim3 grid((g.N + g.NT - 1) / g.NT, g.S);
    dec_gemv_kernel<T, PRO, EPI, NI, MAXT><<<grid, NW * 64, lds, st>>>(g); SS_LAUNCH_CHECK();
}
template <typename T, int PRO, int EPI, int NI>
static void launch_dg2(const DecGemvDesc& g, int NW, hipStream_t st) {
    if (NW <= 4) launch_dg3<T, PRO, EPI, NI, 256>(g, NW, st);
    else launch_dg3<T, PRO, EPI, NI, 1024>(g, NW, st);
}
template <typename T, int PRO, int EPI>
static void launch_dg(const DecGemvDesc& g, int NW, hipStream_t st) {
    if constexpr (PRO == PRO_LN) {
        if (g.K <= 512) launch_dg2<T, PRO, EPI, 2>(g, NW, st);
        else if (g.K <= 1280) launch_dg2<T, PRO, EPI, 5>(g, NW, st);
        else launch_dg2<T, PRO, EPI, 8>(g, NW, st);
    } else {
        launch_dg2<T, PRO, EPI, 1>(g, NW, st);
    }
}


// ---------------------------------------------------------------------------------------------
// The same GEMV for 17..64 token rows (CT = 2 or 4 column tiles of 16): the weight fragments are fetched ONCE and used by CT MFMAs each, so
// a decoder pass over up to 64 rows (several device batches merged, best_of = 5 sampled decoders, long prompts) still streams the decoder
// weights once.  Activations straight from L2 (PRO_T); epilogues as above.
// ---------------------------------------------------------------------------------------------
template <typename T, int EPI, int CT>
__global__ __launch_bounds__(256) void dec_gemv_wide_kernel(DecGemvDesc g) {
    typedef typename MfmaD<T>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16, s = blockIdx.y;
    const int kslice = g.K / g.S, kbeg = s * kslice, kw = kslice / NW, kwb = wave * kw;
    const int nfr = kw / 32, npair = nfr / 2;
    float* red = (float*)smem_d;   // [NW][CT*16][17]
    const T* wp = (const T*)g.W + (long)(n0 + frow) * g.K + kbeg + kwb;
    V8 wf[kMaxFrag];
#pragma unroll
    for (int j = 0; j < kMaxFrag / 2; j++) {
        if (j < npair) {
            wf[2 * j] = SS_LDW((const V8*)(wp + j * 64 + fg * 16));
            wf[2 * j + 1] = SS_LDW((const V8*)(wp + j * 64 + fg * 16 + 8));
        }
    }
    V8 wtail = {};
    if (nfr & 1) wtail = SS_LDW((const V8*)(wp + npair * 64 + fg * 8));
    f32x4 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ct++) {
        acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int m = ct * 16 + frow;
        const T* xg = (const T*)g.Xt + (long)(m < g.M ? m : 0) * g.ldx + kbeg + kwb;
        V8 xf[kMaxFrag];
#pragma unroll
        for (int j = 0; j < kMaxFrag / 2; j++) {
            if (j < npair) {
                xf[2 * j] = *(const V8*)(xg + j * 64 + fg * 16);
                xf[2 * j + 1] = *(const V8*)(xg + j * 64 + fg * 16 + 8);
            }
        }
        V8 xtail = {};
        if (nfr & 1) xtail = *(const V8*)(xg + npair * 64 + fg * 8);
#pragma unroll
        for (int j = 0; j < kMaxFrag / 2; j++) {
            if (j < npair) {
                acc[ct] = MfmaD<T>::mma(wf[2 * j], xf[2 * j], acc[ct]);
                acc[ct] = MfmaD<T>::mma(wf[2 * j + 1], xf[2 * j + 1], acc[ct]);
            }
        }
        if (nfr & 1) acc[ct] = MfmaD<T>::mma(wtail, xtail, acc[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((wave * CT + ct) * 16 + frow) * 17 + fg * 4 + r] = acc[ct][r];
    __syncthreads();
    for (int idx = tid; idx < CT * 256; idx += blockDim.x) {
        const int m = idx >> 4, nn = idx & 15, n = n0 + nn;
        if (m < g.M && n < g.N) {
            float v = 0.f;
            for (int w = 0; w < NW; w++) v += red[((w * CT + (m >> 4)) * 16 + (m & 15)) * 17 + nn];
            dec_epilogue<T, EPI>(g, s, m, n, v);
        }
    }
}
template <typename T, int EPI, int CT>
static void launch_dgw2(const DecGemvDesc& g, int NW, hipStream_t st) {
    const size_t lds = (size_t)NW * CT * 16 * 17 * 4;
    dim3 grid((g.N + 15) / 16, g.S);
    dec_gemv_wide_kernel<T, EPI, CT><<<grid, NW * 64, lds, st>>>(g); SS_LAUNCH_CHECK();
}
template <typename T, int EPI>
static void launch_dgw(const DecGemvDesc& g, int NW, hipStream_t st) {
    if (g.M <= 32) launch_dgw2<T, EPI, 2>(g, NW, st);
    else launch_dgw2<T, EPI, 4>(g, NW, st);
}

// choose split-K so the grid has >= ~256 workgroups; per-wave k must be a multiple of 32 and <= 320
void dec_gemv_plan(int N, int K, int* S_out, int* NW_out, bool whole_heads) {
    int bestS = 0, bestNW = 0, bestScore = -1;
    for (int S = 1; S <= 4; S++) {
        if (K % S) continue;
        for (int NW = 4; NW >= 1; NW >>= 1) {
            const int ks = K / S;
            if (ks % NW || (whole_heads && ks % 64)) continue;
            const int kw = ks / NW;
            if (kw % 32 || kw > 320 || kw < 32) continue;
            const int blocks = ((N + 15) / 16) * S;
            // prefer >= 256 blocks, then fewer splits (less partial traffic), then more waves per block
            int score = (blocks >= 256 ? 1000 : blocks * 3) - S * 8 + NW;
            if (score > bestScore) { bestScore = score; bestS = S; bestNW = NW; }
        }
    }
    if (!bestS) throw Error(-1, "dec_gemv: no split plan for K=" + std::to_string(K));
    *S_out = bestS; *NW_out = bestNW;
}

template <typename T>
void launch_dec_reduce_ln(const DecGemvDesc& g, T* out, hipStream_t st) {
    if (g.K > 2048 || g.n_parts > 4) throw Error(-1, "dec_reduce_ln: bad shape");
    if (g.K <= 512) { dec_reduce_ln_kernel<T, 2><<<g.M, 64, 0, st>>>(g, out); SS_LAUNCH_CHECK(); }
    else if (g.K <= 1280) { dec_reduce_ln_kernel<T, 5><<<g.M, 64, 0, st>>>(g, out); SS_LAUNCH_CHECK(); }
    else { dec_reduce_ln_kernel<T, 8><<<g.M, 64, 0, st>>>(g, out); SS_LAUNCH_CHECK(); }
}
template void launch_dec_reduce_ln<bf16>(const DecGemvDesc&, bf16*, hipStream_t);
template void launch_dec_reduce_ln<f16>(const DecGemvDesc&, f16*, hipStream_t);

template <typename T>
void launch_dec_gemv(const DecGemvDesc& g0, int NW, hipStream_t st) {
    DecGemvDesc g = g0;
    if (g.NT <= 0) g.NT = 16;
    if (g.NT > 16 || NW < 1 || NW > 16 || (NW & (NW - 1))) throw Error(-1, "dec_gemv: bad tile / wave count");
    if (g.M > 16) {   // 17..64 rows: the multi-tile kernel
        if (g.M > kPartRows || g.pro != PRO_T || NW > 4 || g.NT != 16 || g.K % g.S || (g.K / g.S) % NW || ((g.K / g.S) / NW) % 32 || (g.K / g.S) / NW > 320)
            throw Error(-1, "dec_gemv: bad shape for the 17..64-row kernel");
        if (g.epi != DEPI_PART && g.S != 1) throw Error(-1, "dec_gemv: direct epilogues need S == 1");
        switch (g.epi) {
            case DEPI_PART: launch_dgw<T, DEPI_PART>(g, NW, st); break;
            case DEPI_QKV: launch_dgw<T, DEPI_QKV>(g, NW, st); break;
            case DEPI_GELU_T: launch_dgw<T, DEPI_GELU_T>(g, NW, st); break;
            case DEPI_LOGITS: launch_dgw<T, DEPI_LOGITS>(g, NW, st); break;
            default: throw Error(-1, "dec_gemv: unsupported epilogue for the 17..64-row kernel");
        }
        return;
    }
    if (g.n_parts > 4) throw Error(-1, "dec_gemv: at most 4 split-K partials");
    if (g.pro == PRO_COMBINE && (g.K / g.S) % 64) throw Error(-1, "dec_gemv: combine prologue needs K slices of whole heads");
    if (g.M < 1 || g.M > 16 || g.K % g.S || (g.K / g.S) % NW || ((g.K / g.S) / NW) % 32 || (g.K / g.S) / NW > 320)
        throw Error(-1, "dec_gemv: bad shape");
    if (g.epi != DEPI_PART && g.S != 1) throw Error(-1, "dec_gemv: direct epilogues need S == 1");
    if (g.pro == PRO_LN && g.K > 2048) throw Error(-1, "dec_gemv: LayerNorm prologue needs K <= 2048");
#define DG(P, E) launch_dg<T, P, E>(g, NW, st)
    switch (g.pro * 8 + g.epi) {
        case PRO_LN * 8 + DEPI_QKV: DG(PRO_LN, DEPI_QKV); break;
        case PRO_LN * 8 + DEPI_PART: DG(PRO_LN, DEPI_PART); break;
        case PRO_LN * 8 + DEPI_GELU_T: DG(PRO_LN, DEPI_GELU_T); break;
        case PRO_T * 8 + DEPI_LOGITS: DG(PRO_T, DEPI_LOGITS); break;
        case PRO_T * 8 + DEPI_PART: DG(PRO_T, DEPI_PART); break;
        case PRO_T * 8 + DEPI_RES: DG(PRO_T, DEPI_RES); break;
        case PRO_T * 8 + DEPI_QKV: DG(PRO_T, DEPI_QKV); break;
        case PRO_T * 8 + DEPI_GELU_T: DG(PRO_T, DEPI_GELU_T); break;
        case PRO_COMBINE * 8 + DEPI_PART: DG(PRO_COMBINE, DEPI_PART); break;
        default: throw Error(-1, "dec_gemv: unsupported prologue/epilogue pair");
    }
#undef DG
}
template void launch_dec_gemv<bf16>(const DecGemvDesc&, int, hipStream_t);
template void launch_dec_gemv<f16>(const DecGemvDesc&, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// cross-attention with the q projection's split-K reduction in its prologue
// grid (4 key splits, H, M), 256 threads.  q = round_T((sum_s qpart[s][m][:] + bias) * scale)
// ---------------------------------------------------------------------------------------------
template <typename T, int NSPLIT>
__global__ __launch_bounds__(256) void dec_cross_attn_q_kernel(const float* __restrict__ qpart, int n_qpart, const float* __restrict__ qbias, float qscale,
                                                               const T* __restrict__ kc, const T* __restrict__ vc, long b_stride, int d, int H, int Tn,
                                                               const RowCtl* __restrict__ ctl, float* __restrict__ scratch, T* __restrict__ out_direct) {
    typedef typename MfmaD<T>::V8 V8;
    __shared__ float s_sc[(NSPLIT == 1 ? 1536 : 512) + 128];
    __shared__ float s_red[8];
    __shared__ float s_o[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, c = lane & 7;
    const int sp = blockIdx.x, h = blockIdx.y, m = blockIdx.z;
    const int per = (Tn + NSPLIT - 1) / NSPLIT;
    const int k_beg = sp * per, k_end = min(Tn, k_beg + per), nk = k_end - k_beg;
    const RowCtl rc = ctl[m];
    const T* K = kc + (long)rc.cross * b_stride + (long)h * Tn * 64;
    const T* V = vc + (long)rc.cross * b_stride + (long)h * Tn * 64;
    float qv[8];
    {
        const int col = h * 64 + c * 8;
        f32x4 a0 = *(const f32x4*)(qbias + col), a1 = *(const f32x4*)(qbias + col + 4);
        f32x4 t0[4], t1[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {   // up to 4 partial slots, unused ones re-read slot 0 with weight 0
            const float* pp = qpart + ((long)(p < n_qpart ? p : 0) * kPartRows + m) * d + col;
            t0[p] = *(const f32x4*)pp; t1[p] = *(const f32x4*)(pp + 4);
        }
#pragma unroll
        for (int p = 0; p < 4; p++) { const float w = p < n_qpart ? 1.f : 0.f; a0 += t0[p] * w; a1 += t1[p] * w; }
#pragma unroll
        for (int e = 0; e < 4; e++) { qv[e] = (float)(T)(a0[e] * qscale); qv[4 + e] = (float)(T)(a1[e] * qscale); }
    }
    // phase 1: scores.  One wave-instruction reads 8 key rows x 128 B; 4 independent loads in flight per lane
    float mx = -1e30f;
    const int nit = (nk + 31) / 32;
    for (int it = 0; it < nit; it += 4) {
        V8 kv[4];
        int ii[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ii[u] = (it + u) * 32 + wave * 8 + r;
            kv[u] = SS_LDW((const V8*)(K + (long)(k_beg + (ii[u] < nk ? ii[u] : 0)) * 64 + c * 8));
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) a += qv[e] * (float)kv[u][e];
            a = sum_lanes8(a);
            if (ii[u] < nk) {
                if (c == 0) s_sc[ii[u]] = a;
                mx = fmaxf(mx, a);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sum = 0.f;
    for (int i = tid; i < nk; i += 256) {
        const float p = (float)(T)__expf(s_sc[i] - mx);
        s_sc[i] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    // phase 2: o[c*8+e] += p[key] V[key][c*8+e]
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < nit; it += 4) {
        V8 vv[4];
        float pw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = (it + u) * 32 + wave * 8 + r;
            const bool okk = i < nk;
            vv[u] = SS_LDW((const V8*)(V + (long)(k_beg + (okk ? i : 0)) * 64 + c * 8));
            pw[u] = okk ? s_sc[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pw[u] * (float)vv[u][e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
        acc[e] = sum_stride8(acc[e]);
    }
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) s_o[wave][c * 8 + e] = acc[e];
    }
    __syncthreads();
    if constexpr (NSPLIT == 1) {
        if (tid < 64) out_direct[(long)m * d + h * 64 + tid] = (T)((s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid]) / sum);
    } else {
        float* part = scratch + ((long)(m * H + h) * kCrossSplitD + sp) * kCrossPartD;
        if (tid < 64) part[2 + tid] = s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid];
        if (tid == 0) { part[0] = mx; part[1] = sum; }
    }
}

template <typename T>
void launch_dec_cross_attention_direct(const float* qpart, int n_qpart, const float* qbias, float qscale, const T* kc, const T* vc, long b_stride, int d,
                                       int H, int Tn, const RowCtl* ctl, int M, T* out, hipStream_t st) {
    if (Tn > 1536) throw Error(-1, "cross attention: n_audio_ctx too large");
    dim3 grid(1, H, M);
    dec_cross_attn_q_kernel<T, 1><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, vc, b_stride, d, H, Tn, ctl, nullptr, out); SS_LAUNCH_CHECK();
}
template void launch_dec_cross_attention_direct<bf16>(const float*, int, const float*, float, const bf16*, const bf16*, long, int, int, int, const RowCtl*, int,
                                                      bf16*, hipStream_t);
template void launch_dec_cross_attention_direct<f16>(const float*, int, const float*, float, const f16*, const f16*, long, int, int, int, const RowCtl*, int,
                                                     f16*, hipStream_t);

template <typename T>
void launch_dec_cross_attention_q(const float* qpart, int n_qpart, const float* qbias, float qscale, const T* kc, const T* vc, long b_stride, int d, int H,
                                  int Tn, const RowCtl* ctl, int M, float* scratch, hipStream_t st) {
    if ((Tn + kCrossSplitD - 1) / kCrossSplitD > 512) throw Error(-1, "cross attention: n_audio_ctx too large");
    dim3 grid(kCrossSplitD, H, M);
    dec_cross_attn_q_kernel<T, kCrossSplitD><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, vc, b_stride, d, H, Tn, ctl, scratch, nullptr); SS_LAUNCH_CHECK();
}
template void launch_dec_cross_attention_q<bf16>(const float*, int, const float*, float, const bf16*, const bf16*, long, int, int, int, const RowCtl*, int,
                                                 float*, hipStream_t);
template void launch_dec_cross_attention_q<f16>(const float*, int, const float*, float, const f16*, const f16*, long, int, int, int, const RowCtl*, int,
                                                float*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// the same cross-attention over an e4m3 cross cache (fp8 engine): a key row of one head is 64 codes + one exponent byte, so the stream that
// bounds a decoder pass (245.8 MB per sequence per step in f16) is halved.  4 lanes x 16 codes per row, 16 key rows per wave-instruction;
// the exponent of a K row scales its score, the exponent of a V row is folded into its probability.  Arithmetic as the f16 kernel: q rounded
// to T, scores and P.V accumulated in f32, p rounded to T.
// ---------------------------------------------------------------------------------------------
template <typename T, int NSPLIT>
__global__ __launch_bounds__(256) void dec_cross_attn_q8_kernel(const float* __restrict__ qpart, int n_qpart, const float* __restrict__ qbias, float qscale,
                                                                const unsigned char* __restrict__ kc, const unsigned char* __restrict__ ksc, long b_stride,
                                                                long sc_stride, int d, int H, int Tn, const RowCtl* __restrict__ ctl,
                                                                float* __restrict__ scratch, T* __restrict__ out_direct) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float s_sc[(NSPLIT == 1 ? 1536 : 512) + 128];
    __shared__ float s_red[8];
    __shared__ float s_o[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 2, c = lane & 3;            // 16 key rows x 4 chunks of 16 codes per wave-instruction
    const int sp = blockIdx.x, h = blockIdx.y, m = blockIdx.z;
    const int per = (Tn + NSPLIT - 1) / NSPLIT;
    const int k_beg = sp * per, k_end = min(Tn, k_beg + per), nk = k_end - k_beg;
    const RowCtl rc = ctl[m];
    // window layout: codes [kv][h][t][64], exponent bytes [kv][h][t]
    const unsigned char* K = kc + (long)rc.cross * b_stride + (long)h * Tn * 64;
    const unsigned char* V = K + (long)H * Tn * 64;
    const unsigned char* KS = ksc + (long)rc.cross * sc_stride + (long)h * Tn;
    const unsigned char* VS = KS + (long)H * Tn;
    float qv[16];
    {
        const int col = h * 64 + c * 16;
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {
            f32x4 a = *(const f32x4*)(qbias + col + q4 * 4);
            f32x4 t[4];
#pragma unroll
            for (int p = 0; p < 4; p++) t[p] = *(const f32x4*)(qpart + ((long)(p < n_qpart ? p : 0) * kPartRows + m) * d + col + q4 * 4);
#pragma unroll
            for (int p = 0; p < 4; p++) a += t[p] * (p < n_qpart ? 1.f : 0.f);
#pragma unroll
            for (int e = 0; e < 4; e++) qv[q4 * 4 + e] = (float)(T)(a[e] * qscale);
        }
    }
    auto dot16 = [&](const u32x4& w, const float* x) {
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            a += x[4 * j + 0] * __builtin_amdgcn_cvt_f32_fp8((int)w[j], 0);
            a += x[4 * j + 1] * __builtin_amdgcn_cvt_f32_fp8((int)w[j], 1);
            a += x[4 * j + 2] * __builtin_amdgcn_cvt_f32_fp8((int)w[j], 2);
            a += x[4 * j + 3] * __builtin_amdgcn_cvt_f32_fp8((int)w[j], 3);
        }
        return a;
    };
    // phase 1: scores
    float mx = -1e30f;
    const int nit = (nk + 63) / 64;                   // 64 keys per block iteration (4 waves x 16 rows)
    for (int it = 0; it < nit; it += 4) {
        u32x4 kw[4];
        int ii[4];
        unsigned char eb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ii[u] = (it + u) * 64 + wave * 16 + r;
            const int kk = k_beg + (ii[u] < nk ? ii[u] : 0);
            kw[u] = SS_LDW((const u32x4*)(K + (long)kk * 64 + c * 16));
            eb[u] = KS[kk];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float a = dot16(kw[u], qv);
            a += dpp_mov<kDppXor1>(a);
            a += dpp_mov<kDppXor2>(a);
            a *= __builtin_bit_cast(float, (unsigned)eb[u] << 23);
            if (ii[u] < nk) {
                if (c == 0) s_sc[ii[u]] = a;
                mx = fmaxf(mx, a);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sum = 0.f;
    for (int i = tid; i < nk; i += 256) {
        const float p = (float)(T)__expf(s_sc[i] - mx);
        s_sc[i] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    // phase 2: o[c*16 + e] += p[key] * 2^(ev - 127) * code
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.f;
    for (int it = 0; it < nit; it += 4) {
        u32x4 vw[4];
        float pw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = (it + u) * 64 + wave * 16 + r;
            const bool okk = i < nk;
            const int kk = k_beg + (okk ? i : 0);
            vw[u] = SS_LDW((const u32x4*)(V + (long)kk * 64 + c * 16));
            pw[u] = okk ? s_sc[i] * __builtin_bit_cast(float, (unsigned)VS[kk] << 23) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[4 * j + 0] += pw[u] * __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 0);
                acc[4 * j + 1] += pw[u] * __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 1);
                acc[4 * j + 2] += pw[u] * __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 2);
                acc[4 * j + 3] += pw[u] * __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 3);
            }
    }
    // sum over the 16 lanes that share lane & 3 (strides 4, 8, 16, 32)
#pragma unroll
    for (int e = 0; e < 16; e++) {
        float a = acc[e];
        a += dpp_mov<kDppRor4>(a);
        a += dpp_mov<kDppRor8>(a);
        a = rows_sum(a);
        acc[e] = a;
    }
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 16; e++) s_o[wave][c * 16 + e] = acc[e];
    }
    __syncthreads();
    if constexpr (NSPLIT == 1) {
        if (tid < 64) out_direct[(long)m * d + h * 64 + tid] = (T)((s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid]) / sum);
    } else {
        float* part = scratch + ((long)(m * H + h) * kCrossSplitD + sp) * kCrossPartD;
        if (tid < 64) part[2 + tid] = s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid];
        if (tid == 0) { part[0] = mx; part[1] = sum; }
    }
}

template <typename T>
void launch_dec_cross_attention_f8(const float* qpart, int n_qpart, const float* qbias, float qscale, const unsigned char* kc, const unsigned char* ksc,
                                   long b_stride, long sc_stride, int d, int H, int Tn, const RowCtl* ctl, int M, float* scratch, T* out, hipStream_t st) {
    if (scratch) {
        if ((Tn + kCrossSplitD - 1) / kCrossSplitD > 512) throw Error(-1, "cross attention: n_audio_ctx too large");
        dim3 grid(kCrossSplitD, H, M);
        dec_cross_attn_q8_kernel<T, kCrossSplitD><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, ksc, b_stride, sc_stride, d, H, Tn, ctl, scratch, nullptr);
    } else {
        if (Tn > 1536) throw Error(-1, "cross attention: n_audio_ctx too large");
        dim3 grid(1, H, M);
        dec_cross_attn_q8_kernel<T, 1><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, ksc, b_stride, sc_stride, d, H, Tn, ctl, nullptr, out);
    }
    SS_LAUNCH_CHECK();
}
template void launch_dec_cross_attention_f8<bf16>(const float*, int, const float*, float, const unsigned char*, const unsigned char*, long, long, int, int, int,
                                                  const RowCtl*, int, float*, bf16*, hipStream_t);
template void launch_dec_cross_attention_f8<f16>(const float*, int, const float*, float, const unsigned char*, const unsigned char*, long, long, int, int, int,
                                                 const RowCtl*, int, float*, f16*, hipStream_t);

}  // namespace ss
