// Decode-step kernels for gfx950: one token row for each of up to 64 sequences (a decoder PASS of the batched engine).
//
// A decoder pass is HBM-bound (1.6 GB of decoder weights + rows x 246 MB of cross-KV for large-v3) but, launched as one small kernel per
// graph node the way ggml does (/root/reference/resources/ggml-metal.metal:1307-1363 kernel_mul_mv_f16_f32, :571-621 kernel_norm, :54-151 add/mul),
// it is launch/latency-bound: ~350 dependent launches per step.  Here a layer is 11-12 launches:
//   dec_reduce_ln_kernel   residual add + bias of the previous projection + deterministic reduction of its split-K partials (+ token/positional
//                          embedding for layer 0) -> LayerNorm -> f16/bf16 rows                                           (one wave per row)
//   dec_gemv_kernel        weight fragments AND activation fragments go straight from HBM/L2 into VGPRs, all loads issued before the first
//                          16x16x32 MFMA (16-row weight tile = A operand, 1 or 2 column tiles of 16 token rows = B operands fed by ONE fetch
//                          of the weight fragments); epilogues: q/k/v scaling + KV-cache append, GELU, logits, or raw split-K partials
//   dec_cross_attn_q(8)_kernel   cross-attention over the 1500 encoder positions with the q projection's reduction in its prologue
// Split-K goes across workgroups (grid = N/16 x S) so that N = d projections still launch >= 256 workgroups; partials are summed in a fixed
// order by the consumer, so results are run-to-run identical (no float atomics).
// Variants measured slower and archived (tools/experiments/r02_variants/r02_variants.diff): LayerNorm / flash-decoding-combine prologues inside the
// GEMV, narrow output tiles without split-K, a residual-update epilogue, non-temporal weight loads (-1.2 %).
#include <cstdlib>
#include "kernels.h"
#include "wave_ops.h"

#ifdef SS_WEIGHTS_NT           // dev A/B (tools/build_variant.sh kernels_decode.hip -DSS_WEIGHTS_NT): decoder weight fragments as non-temporal loads
#define SS_LDW(p) __builtin_nontemporal_load(p)
#else
#define SS_LDW(p) (*(p))   // decoder weight fragments: plain loads
#endif
// The latency-bound chain kernels (GEMVs, reduce + LayerNorm, self-attention) raise their waves' issue priority: with several lanes in flight their
// waves share SIMDs with another lane's streaming cross-attention (or encoder GEMM) waves, which have plenty of independent work to issue; the
// chain wave's handful of instructions are on some lane's critical path.  A/B/A/B on one box (profiles/r04_l_chain_prio_ab.txt): 3 lanes x 32 rows
// 3091 / 3095x -> 3133 / 3138x, pass 6.70 -> 6.57 ms; one lane alone: no change.  (The cross-attention one level above the encoder GEMMs, s_setprio 1:
// 3127 - 3137x against 3134 - 3149x, no gain, profiles/r04_m_cross_prio_ab.txt.)
#define SS_CHAIN_PRIO_STMT __builtin_amdgcn_s_setprio(3);
// The cross-K/V stream (rows x 245.8 MB per pass, every byte used once) can be marked non-temporal (`global_load ... nt`): its lines then leave L2
// first instead of evicting the chain kernels' weights / activations and the encoder GEMMs' operand tiles of the other lanes.  Same bytes, same
// arithmetic.  A/B/A/B on one box (profiles/r06_a_nt_zsplit_yardstick.txt): 3 lanes x 32 rows 3392 / 3400x -> 3461 / 3496x, pass 5.96 / 5.89 -> 5.74 / 5.66 ms;
// one lane x 8 rows: no change (nothing to evict).  On by default; SS_CROSS_NT=0 restores plain loads.
template <bool NT, typename V> __device__ __forceinline__ V ld_stream(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}


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

constexpr int kCrossSplitD = 4, kCrossPartD = 66;

// x = [x_in | tok+pos embedding] + bias_prev + sum_p parts[p]  (fixed order, branch-free: up to 4 partial slots, unused
// slots re-read slot 0 with weight 0 so that every load is independent and in flight together), optional write-back,
// LayerNorm over the full row, normalised columns [kbeg, kbeg+kslice) written to dst as T.  One wave per row.
// pack_row >= 0: dst is the base of a fragment-major activation buffer (kernels.h dec_wpack_off) and this row is row pack_row of it;
// pack_row < 0: dst points at a plain row (the LDS staging of dec_gemv_ln_kernel)
template <typename T, int NI>
__device__ __forceinline__ void ln_row(const DecGemvDesc& g, int m, int lane, bool write_x, int kbeg, int kslice, T* dst, int pack_row = -1) {
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
    if (g.ctl) {  // layer 0: token + positional embedding (replaces ggml get_rows + add)
        const RowCtl rc = g.ctl[r];
        const float* pe = g.pos_emb + (long)rc.pos * d;
#pragma unroll
        for (int i = 0; i < NI; i++) {
            const f32x4 p4 = *(const f32x4*)(pe + cc[i]);
            const T* t = (const T*)g.tok_emb + dec_wpack_off(rc.token, cc[i], d);   // 4 consecutive k of a row stay contiguous in the fragment-major layout
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
    if (write_x && g.x_out) {
#pragma unroll
        for (int i = 0; i < NI; i++) if (ok[i]) *(f32x4*)(g.x_out + (long)r * d + cc[i]) = v[i];
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
            T* q = pack_row >= 0 ? dst + dec_wpack_off(pack_row, c, d) : dst + (c - kbeg);   // c is a multiple of 4: the four stay contiguous either way
#pragma unroll
            for (int e = 0; e < 4; e++) q[e] = (T)(v[i][e] * rstd * ww[i][e] + bb[i][e]);
        }
    }
}

// one wave per row
template <typename T, int NI>
__global__ __launch_bounds__(64) void dec_reduce_ln_kernel(DecGemvDesc g, T* out) {
    SS_CHAIN_PRIO_STMT
    ln_row<T, NI>(g, blockIdx.x, threadIdx.x, true, 0, g.K, out, blockIdx.x);
}

// one output (token row m, output column n, whole-slice sum v) of a decode GEMV
template <typename T, int EPI>
__device__ __forceinline__ void dec_epilogue(const DecGemvDesc& g, int s, int m, int n, float v) {
    if constexpr (EPI == DEPI_PART) {
        g.part_out[((long)s * kPartRows + m) * g.N + n] = v;
    } else {
        if (g.bias) v += g.bias[n];
        if constexpr (EPI == DEPI_GELU_T) {
            ((T*)g.out)[dec_wpack_off(m, n, g.ldo)] = (T)gelu_tanh_d(gelu_in_round_d<T>(v, g.gelu_f16_in));   // FC2's B operand: fragment-major
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

// Fragment f of a wave covers k = kbeg + 32 f .. + 31, lane group fg its 8 consecutive k at 8 fg: the weights are stored fragment-major
// (kernels.h dec_wpack_off: one contiguous kilobyte per fragment), the activations row-major.

// NFR = 32-k fragments per wave (k per wave = 32 NFR <= 320), a template parameter: with a run-time count the loads sat behind branches and
// hipcc put `s_waitcnt vmcnt(0)` between the weight loads, the activation loads of the first column tile and those of the second -- three
// dependent memory round trips per GEMV (r03: 10.9 us per 32-row projection in the pipeline).  Now every load of the workgroup -- NFR weight
// fragments from HBM, CT x NFR activation fragments from L2 -- is issued before the first MFMA.  CT = column tiles of 16 token rows (1, 2); grid.z = groups of CT tiles.
template <typename T, int EPI, int CT, int NFR>
__global__ __launch_bounds__(256) void dec_gemv_kernel(DecGemvDesc g) {
    SS_CHAIN_PRIO_STMT
    typedef typename MfmaD<T>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16, s = blockIdx.y;
    const int kbeg = s * (g.K / g.S) + wave * (32 * NFR);
    float* red = (float*)smem_d;   // [NW][CT*16][17]
    const T* wp = (const T*)g.W + ((long)blockIdx.x * (g.K >> 5) + (kbeg >> 5)) * 512 + lane * 8;      // N is a multiple of 16 (checked on the host): every weight tile exists
    V8 wf[NFR];
#pragma unroll
    for (int f = 0; f < NFR; f++) wf[f] = SS_LDW((const V8*)(wp + f * 512));
    // activations: B fragments straight from L2 into VGPRs (no LDS staging, no barrier before the MFMAs), fragment-major like the weights:
    // one contiguous kilobyte per load.  Token rows >= M hold stale finite values: MFMA columns are independent and never stored.
    f32x4 acc[CT];
#pragma unroll
    for (int c0 = 0; c0 < CT; c0 += 2) {
        constexpr int CB = CT >= 2 ? 2 : 1;
        V8 xf[CB][NFR];
#pragma unroll
        for (int ct = 0; ct < CB; ct++) {
            const int tz = blockIdx.z * CT + c0 + ct;                   // blockIdx.z: which group of CT column tiles (passes of more than 32 rows, launch_dg)
            const int mt = tz * 16 < g.M ? tz : 0;                      // a column tile wholly beyond M reads tile 0: its MFMA columns are never stored
            const T* xg = (const T*)g.Xt + (((long)mt * (g.ldx >> 5) + (kbeg >> 5)) * 64 + lane) * 8;
#pragma unroll
            for (int f = 0; f < NFR; f++) xf[ct][f] = *(const V8*)(xg + f * 512);
        }
        __builtin_amdgcn_sched_barrier(0);     // every load above is issued before the first MFMA below (left alone, hipcc trickles them in between the MFMAs to save registers)
#pragma unroll
        for (int ct = 0; ct < CB; ct++) {
            acc[c0 + ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int f = 0; f < NFR; f++) acc[c0 + ct] = MfmaD<T>::mma(wf[f], xf[ct][f], acc[c0 + ct]);
        }
    }
    // D[n][m]: lane holds n = fg*4 + r, m = frow
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((wave * CT + ct) * 16 + frow) * 17 + fg * 4 + r] = acc[ct][r];
    __syncthreads();
    // ---- epilogue: (CT * 16) m x 16 n outputs ----
    for (int idx = tid; idx < CT * 256; idx += blockDim.x) {
        const int ml = idx >> 4, nn = idx & 15, n = n0 + nn, m = blockIdx.z * (CT * 16) + ml;
        if (m < g.M && n < g.N) {
            float v = 0.f;
            for (int w = 0; w < NW; w++) v += red[((w * CT + (ml >> 4)) * 16 + (ml & 15)) * 17 + nn];
            dec_epilogue<T, EPI>(g, s, m, n, v);
        }
    }
}

// LayerNorm-prologue form for M <= 16 rows (dispatched for M <= kLnFuseRows): the weight fragments are requested FIRST, then the waves normalise
// the rows into LDS (wave per row, ln_row) while those loads are in flight, then the B fragments come from LDS.  Workgroup (0, 0) also writes the
// updated residual stream (x_out is the other half of a ping-pong pair, so the other workgroups still read the old rows).
// The k ranges of the split-K slices and of the waves, and the order in which a column's wave sums are added, are those of dec_gemv_kernel for the
// same (S, NW) plan: a row computed here (<= kLnFuseRows rows in the pass) has the bits it gets from the dec_reduce_ln + dec_gemv pair in a
// wider pass (batch invariance, round 5; the engine therefore launches the cross-query projection with the S-way plan in both forms).
template <typename T, int EPI, int NFR, int NI>
__global__ __launch_bounds__(256) void dec_gemv_ln_kernel(DecGemvDesc g) {
    SS_CHAIN_PRIO_STMT
    typedef typename MfmaD<T>::V8 V8;
    extern __shared__ __attribute__((aligned(16))) char smem_d[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = blockDim.x >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16, s = blockIdx.y;            // s: split-K slice (DEPI_PART only), the same slices and wave ranges as dec_gemv_kernel
    const int kbeg = s * (g.K / g.S) + wave * (32 * NFR);
    const int xld = g.K + 8;                                   // row stride in T elements: 16-B aligned rows, banks spread
    T* xs = (T*)smem_d;                                        // [16][xld]
    float* red = (float*)(smem_d + (size_t)16 * xld * sizeof(T));   // [NW][16][17]
    const T* wp = (const T*)g.W + ((long)blockIdx.x * (g.K >> 5) + (kbeg >> 5)) * 512 + lane * 8;
    V8 wf[NFR];
#pragma unroll
    for (int f = 0; f < NFR; f++) wf[f] = SS_LDW((const V8*)(wp + f * 512));
    for (int m = wave; m < 16; m += NW) {
        if (m < g.M) ln_row<T, NI>(g, m, lane, blockIdx.x == 0 && blockIdx.y == 0, 0, g.K, xs + (long)m * xld);
        else for (int c = lane * 8; c < g.K; c += 512) *(V8*)(xs + (long)m * xld + c) = V8{};      // unused MFMA columns: finite values
    }
    __syncthreads();
    V8 xf[NFR];
#pragma unroll
    for (int f = 0; f < NFR; f++) xf[f] = *(const V8*)(xs + (long)frow * xld + kbeg + f * 32 + fg * 8);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int f = 0; f < NFR; f++) acc = MfmaD<T>::mma(wf[f], xf[f], acc);
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 16 + frow) * 17 + fg * 4 + r] = acc[r];
    __syncthreads();
    for (int idx = tid; idx < 256; idx += blockDim.x) {
        const int m = idx >> 4, nn = idx & 15, n = n0 + nn;
        if (m < g.M && n < g.N) {
            float v = 0.f;
            for (int w = 0; w < NW; w++) v += red[(w * 16 + m) * 17 + nn];
            dec_epilogue<T, EPI>(g, s, m, n, v);
        }
    }
}
template <typename T, int EPI, int NFR>
static void launch_dgl2(const DecGemvDesc& g, int NW, hipStream_t st) {
    const size_t lds = (size_t)16 * (g.K + 8) * sizeof(T) + (size_t)NW * 16 * 17 * 4;
    dim3 grid(g.N / 16, g.S);
    if (g.K <= 512) dec_gemv_ln_kernel<T, EPI, NFR, 2><<<grid, NW * 64, lds, st>>>(g);
    else if (g.K <= 1280) dec_gemv_ln_kernel<T, EPI, NFR, 5><<<grid, NW * 64, lds, st>>>(g);
    else dec_gemv_ln_kernel<T, EPI, NFR, 8><<<grid, NW * 64, lds, st>>>(g);
    SS_LAUNCH_CHECK();
}
template <typename T, int EPI>
static void launch_dgl(const DecGemvDesc& g, int NW, hipStream_t st) {
    switch (((g.K / g.S) / NW) / 32) {
        case 1: launch_dgl2<T, EPI, 1>(g, NW, st); break;
        case 2: launch_dgl2<T, EPI, 2>(g, NW, st); break;
        case 3: launch_dgl2<T, EPI, 3>(g, NW, st); break;
        case 4: launch_dgl2<T, EPI, 4>(g, NW, st); break;
        case 5: launch_dgl2<T, EPI, 5>(g, NW, st); break;
        case 6: launch_dgl2<T, EPI, 6>(g, NW, st); break;
        case 8: launch_dgl2<T, EPI, 8>(g, NW, st); break;
        case 10: launch_dgl2<T, EPI, 10>(g, NW, st); break;
        default: throw Error(-1, "dec_gemv_ln: k per wave must be 32..320 (1-6, 8, 10 fragments)");
    }
}

template <typename T, int EPI, int CT, int NFR>
static void launch_dg4(const DecGemvDesc& g, int NW, hipStream_t st) {
    const size_t lds = (size_t)NW * CT * 16 * 17 * 4;
    dim3 grid(g.N / 16, g.S, (g.M + CT * 16 - 1) / (CT * 16));
    dec_gemv_kernel<T, EPI, CT, NFR><<<grid, NW * 64, lds, st>>>(g); SS_LAUNCH_CHECK();
}
template <typename T, int EPI, int CT>
static void launch_dg3(const DecGemvDesc& g, int NW, hipStream_t st) {
    switch (((g.K / g.S) / NW) / 32) {
        case 1: launch_dg4<T, EPI, CT, 1>(g, NW, st); break;
        case 2: launch_dg4<T, EPI, CT, 2>(g, NW, st); break;
        case 3: launch_dg4<T, EPI, CT, 3>(g, NW, st); break;
        case 4: launch_dg4<T, EPI, CT, 4>(g, NW, st); break;
        case 5: launch_dg4<T, EPI, CT, 5>(g, NW, st); break;
        case 6: launch_dg4<T, EPI, CT, 6>(g, NW, st); break;
        case 7: launch_dg4<T, EPI, CT, 7>(g, NW, st); break;
        case 8: launch_dg4<T, EPI, CT, 8>(g, NW, st); break;
        case 9: launch_dg4<T, EPI, CT, 9>(g, NW, st); break;
        case 10: launch_dg4<T, EPI, CT, 10>(g, NW, st); break;
        default: throw Error(-1, "dec_gemv: k per wave must be 32..320");
    }
}
// Passes of more than 32 rows: grid.z = groups of 32 rows, every workgroup IS the 32-row kernel.  The groups of one weight tile are
// gridDim.x * gridDim.y (a multiple of 8) apart in dispatch order, i.e. on the same XCD: the tile comes from HBM once and from that XCD's L2 for the
// other groups.  MFMA columns are independent, so a row's bits do not depend on the form (tests/test_gpu_batch_invariance.py).  Rounds 4-5 gave ONE
// workgroup all the rows (4 / 8 column tiles, two at a time): 2 - 4 dependent rounds of activation loads, 21 - 37 us per 128-row projection against
// 7 - 10 us at 32 rows; A/B on one box (profiles/r06_a_nt_zsplit_yardstick.txt): 1 lane x 128 rows 3156 -> 3283x, 2 lanes x 64 rows 3337 -> 3390x.
template <typename T, int EPI>
static void launch_dg(const DecGemvDesc& g, int NW, hipStream_t st) {
    if (g.M <= 16) launch_dg3<T, EPI, 1>(g, NW, st);
    else launch_dg3<T, EPI, 2>(g, NW, st);
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
void launch_dec_gemv(const DecGemvDesc& g, int NW, hipStream_t st) {
    if (NW < 1 || NW > 4 || (NW & (NW - 1))) throw Error(-1, "dec_gemv: bad wave count");
    if (g.M < 1 || g.M > kPartRows || g.N % 16 || g.K % g.S || (g.K / g.S) % NW || ((g.K / g.S) / NW) % 32 || (g.K / g.S) / NW > 320)
        throw Error(-1, "dec_gemv: bad shape");
    if (g.epi != DEPI_PART && g.S != 1) throw Error(-1, "dec_gemv: direct epilogues need S == 1");
    switch (g.epi) {
        case DEPI_PART: launch_dg<T, DEPI_PART>(g, NW, st); break;
        case DEPI_QKV: launch_dg<T, DEPI_QKV>(g, NW, st); break;
        case DEPI_GELU_T: launch_dg<T, DEPI_GELU_T>(g, NW, st); break;
        case DEPI_LOGITS: launch_dg<T, DEPI_LOGITS>(g, NW, st); break;
        default: throw Error(-1, "dec_gemv: unsupported epilogue");
    }
}
template <typename T>
void launch_dec_gemv_ln(const DecGemvDesc& g, int NW, hipStream_t st) {
    if (NW < 1 || NW > 4 || (NW & (NW - 1))) throw Error(-1, "dec_gemv_ln: bad wave count");
    if (g.M < 1 || g.M > 16 || g.N % 16 || g.S < 1 || g.K % g.S || (g.K / g.S) % NW || ((g.K / g.S) / NW) % 32 || (g.K / g.S) / NW > 320 || g.K > 2048 ||
        g.K % 8 || g.n_parts > 4)
        throw Error(-1, "dec_gemv_ln: bad shape");
    if (g.epi != DEPI_PART && g.S != 1) throw Error(-1, "dec_gemv_ln: direct epilogues need S == 1");
    switch (g.epi) {
        case DEPI_PART: launch_dgl<T, DEPI_PART>(g, NW, st); break;
        case DEPI_QKV: launch_dgl<T, DEPI_QKV>(g, NW, st); break;
        case DEPI_GELU_T: launch_dgl<T, DEPI_GELU_T>(g, NW, st); break;
        case DEPI_LOGITS: launch_dgl<T, DEPI_LOGITS>(g, NW, st); break;
        default: throw Error(-1, "dec_gemv_ln: unsupported epilogue");
    }
}
template void launch_dec_gemv_ln<bf16>(const DecGemvDesc&, int, hipStream_t);
template void launch_dec_gemv_ln<f16>(const DecGemvDesc&, int, hipStream_t);
template void launch_dec_gemv<bf16>(const DecGemvDesc&, int, hipStream_t);
template void launch_dec_gemv<f16>(const DecGemvDesc&, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// cross-attention with the q projection's split-K reduction in its prologue.  q = round_T((sum_s qpart[s][m][:] + bias) * scale)
//
// BATCH INVARIANCE (round 5): the 1500 keys are ALWAYS walked as the same kCrossSplitD = 4 ranges, each with its own max / sum / o[64] computed
// in one fixed operand order, and the four partials are ALWAYS merged by cross_combine() below.  NR = 1: grid (4, H, M), one range per
// workgroup, partials through scratch, dec_cross_combine_kernel merges (few rows: the splits fill the chip).  NR = 4: grid (1, H, M), one
// workgroup walks all four ranges and merges them itself (rows x heads >= 320: no partials, no combine launch).  Both forms execute the same
// floating-point operations on the same operands in the same order, so a row's output bits do not depend on how many other rows share its
// pass -- what state.full() guarantees per (state, audio) (/root/reference/src/asr/whisper.rs:75).  Rounds 2-4 ran the NR = 4 case as ONE
// range (one max over 1500 keys): a different rounding of every p, i.e. results that depended on the batch.
// ---------------------------------------------------------------------------------------------
static int g_cross_nt = -1;
static bool cross_nt() {
    if (g_cross_nt < 0) { const char* e = getenv("SS_CROSS_NT"); g_cross_nt = e ? atoi(e) != 0 : 1; }
    return g_cross_nt != 0;
}
constexpr int kCrossRangeMax = 512;   // keys per range the LDS score buffer holds (n_audio_ctx <= 2048)

__device__ __forceinline__ float cross_combine(const float mxs[kCrossSplitD], const float sums[kCrossSplitD], const float os[kCrossSplitD]) {
    float mx = -1e30f;
#pragma unroll
    for (int s = 0; s < kCrossSplitD; s++) mx = fmaxf(mx, mxs[s]);
    float num = 0.f, den = 0.f;
#pragma unroll
    for (int s = 0; s < kCrossSplitD; s++) {
        const float w = __expf(mxs[s] - mx);
        num = __builtin_fmaf(w, os[s], num);
        den = __builtin_fmaf(w, sums[s], den);
    }
    return num / den;
}

template <typename T, int NR, bool NT>
__global__ __launch_bounds__(256) void dec_cross_attn_q_kernel(const float* __restrict__ qpart, int n_qpart, const float* __restrict__ qbias, float qscale,
                                                               const T* __restrict__ kc, const T* __restrict__ vc, long b_stride, int d, int H, int Tn,
                                                               const RowCtl* __restrict__ ctl, float* __restrict__ scratch, T* __restrict__ out_direct) {
    typedef typename MfmaD<T>::V8 V8;
    __shared__ float s_sc[NR * kCrossRangeMax];
    __shared__ float s_red[2][NR][4];
    __shared__ float s_o[NR][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, c = lane & 7;
    const int h = blockIdx.y, m = blockIdx.z;
    const RowCtl rc = ctl[m];
    const int nkeys = rc.n_keys > 0 ? rc.n_keys : Tn;      // whisper_full_params.audio_ctx: the window's keys are the first nkeys rows of its cache slot
    const int per = (nkeys + kCrossSplitD - 1) / kCrossSplitD;
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
    float mx[NR];
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int k_beg = (blockIdx.x * NR + g) * per, nk = max(0, min(nkeys, k_beg + per) - k_beg), nit = (nk + 31) / 32;
        float* sc = s_sc + g * kCrossRangeMax;
        mx[g] = -1e30f;
        for (int it = 0; it < nit; it += 4) {
            V8 kv[4];
            int ii[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ii[u] = (it + u) * 32 + wave * 8 + r;
                kv[u] = ld_stream<NT>((const V8*)(K + (long)(k_beg + (ii[u] < nk ? ii[u] : 0)) * 64 + c * 8));
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; e++) a = __builtin_fmaf(qv[e], (float)kv[u][e], a);
                a = sum_lanes8(a);
                if (ii[u] < nk) {
                    if (c == 0) sc[ii[u]] = a;
                    mx[g] = fmaxf(mx[g], a);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < NR; g++) {
        mx[g] = wave_max(mx[g]);
        if (lane == 0) s_red[0][g][wave] = mx[g];
    }
    __syncthreads();
    float sum[NR];
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int k_beg = (blockIdx.x * NR + g) * per, nk = max(0, min(nkeys, k_beg + per) - k_beg);
        float* sc = s_sc + g * kCrossRangeMax;
        mx[g] = fmaxf(fmaxf(s_red[0][g][0], s_red[0][g][1]), fmaxf(s_red[0][g][2], s_red[0][g][3]));
        float sm = 0.f;
        for (int i = tid; i < nk; i += 256) {
            const float p = (float)(T)__expf(sc[i] - mx[g]);
            sc[i] = p;
            sm += p;
        }
        sm = wave_sum(sm);
        if (lane == 0) s_red[1][g][wave] = sm;
    }
    __syncthreads();
    // phase 2: o[c*8+e] += p[key] V[key][c*8+e]
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int k_beg = (blockIdx.x * NR + g) * per, nk = max(0, min(nkeys, k_beg + per) - k_beg), nit = (nk + 31) / 32;
        const float* sc = s_sc + g * kCrossRangeMax;
        sum[g] = ((s_red[1][g][0] + s_red[1][g][1]) + s_red[1][g][2]) + s_red[1][g][3];
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int it = 0; it < nit; it += 4) {
            V8 vv[4];
            float pw[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = (it + u) * 32 + wave * 8 + r;
                const bool okk = i < nk;
                vv[u] = ld_stream<NT>((const V8*)(V + (long)(k_beg + (okk ? i : 0)) * 64 + c * 8));
                pw[u] = okk ? sc[i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int e = 0; e < 8; e++) acc[e] = __builtin_fmaf(pw[u], (float)vv[u][e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = sum_stride8(acc[e]);
        if (r == 0) {
#pragma unroll
            for (int e = 0; e < 8; e++) s_o[g][wave][c * 8 + e] = acc[e];
        }
    }
    __syncthreads();
    if (tid < 64) {
        float os[NR];
#pragma unroll
        for (int g = 0; g < NR; g++) os[g] = ((s_o[g][0][tid] + s_o[g][1][tid]) + s_o[g][2][tid]) + s_o[g][3][tid];
        if constexpr (NR == kCrossSplitD) {
            out_direct[dec_wpack_off(m, h * 64 + tid, d)] = (T)cross_combine(mx, sum, os);   // the out-projection's B operand
        } else {
            static_assert(NR == 1, "one range per workgroup, or all of them");
            float* part = scratch + ((long)(m * H + h) * kCrossSplitD + blockIdx.x) * kCrossPartD;
            part[2 + tid] = os[0];
            if (tid == 0) { part[0] = mx[0]; part[1] = sum[0]; }
        }
    }
}

// flash-decoding combine of the key-split partials (max, sum, o[64]) the NR = 1 form leaves in scratch: out T [M][d], fragment-major
template <typename T>
__global__ void dec_cross_combine_kernel(const float* __restrict__ scratch, int d, int H, T* __restrict__ out) {
    SS_CHAIN_PRIO_STMT
    const int m = blockIdx.x;
    for (int col = threadIdx.x; col < d; col += blockDim.x) {
        const int h = col >> 6, j = col & 63;
        const float* part = scratch + (long)(m * H + h) * kCrossSplitD * kCrossPartD;
        float mxs[kCrossSplitD], sums[kCrossSplitD], os[kCrossSplitD];
#pragma unroll
        for (int s = 0; s < kCrossSplitD; s++) { mxs[s] = part[s * kCrossPartD]; sums[s] = part[s * kCrossPartD + 1]; os[s] = part[s * kCrossPartD + 2 + j]; }
        out[dec_wpack_off(m, col, d)] = (T)cross_combine(mxs, sums, os);
    }
}
template <typename T>
void launch_dec_cross_combine(const float* scratch, int d, int H, int M, T* out, hipStream_t st) {
    dec_cross_combine_kernel<T><<<M, 256, 0, st>>>(scratch, d, H, out); SS_LAUNCH_CHECK();
}
template void launch_dec_cross_combine<bf16>(const float*, int, int, int, bf16*, hipStream_t);
template void launch_dec_cross_combine<f16>(const float*, int, int, int, f16*, hipStream_t);

template <typename T>
void launch_dec_cross_attention_direct(const float* qpart, int n_qpart, const float* qbias, float qscale, const T* kc, const T* vc, long b_stride, int d,
                                       int H, int Tn, const RowCtl* ctl, int M, T* out, hipStream_t st) {
    if ((Tn + kCrossSplitD - 1) / kCrossSplitD > kCrossRangeMax) throw Error(-1, "cross attention: n_audio_ctx too large");
    dim3 grid(1, H, M);
    if (cross_nt()) dec_cross_attn_q_kernel<T, kCrossSplitD, true><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, vc, b_stride, d, H, Tn, ctl, nullptr, out);
    else dec_cross_attn_q_kernel<T, kCrossSplitD, false><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, vc, b_stride, d, H, Tn, ctl, nullptr, out);
    SS_LAUNCH_CHECK();
}
template void launch_dec_cross_attention_direct<bf16>(const float*, int, const float*, float, const bf16*, const bf16*, long, int, int, int, const RowCtl*, int,
                                                      bf16*, hipStream_t);
template void launch_dec_cross_attention_direct<f16>(const float*, int, const float*, float, const f16*, const f16*, long, int, int, int, const RowCtl*, int,
                                                     f16*, hipStream_t);

template <typename T>
void launch_dec_cross_attention_q(const float* qpart, int n_qpart, const float* qbias, float qscale, const T* kc, const T* vc, long b_stride, int d, int H,
                                  int Tn, const RowCtl* ctl, int M, float* scratch, hipStream_t st) {
    if ((Tn + kCrossSplitD - 1) / kCrossSplitD > kCrossRangeMax) throw Error(-1, "cross attention: n_audio_ctx too large");
    dim3 grid(kCrossSplitD, H, M);
    if (cross_nt()) dec_cross_attn_q_kernel<T, 1, true><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, vc, b_stride, d, H, Tn, ctl, scratch, nullptr);
    else dec_cross_attn_q_kernel<T, 1, false><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, vc, b_stride, d, H, Tn, ctl, scratch, nullptr);
    SS_LAUNCH_CHECK();
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
template <typename T, int NR, bool NT>   // NR as in dec_cross_attn_q_kernel: the same four key ranges and the same combine whichever form runs
__global__ __launch_bounds__(256) void dec_cross_attn_q8_kernel(const float* __restrict__ qpart, int n_qpart, const float* __restrict__ qbias, float qscale,
                                                                const unsigned char* __restrict__ kc, const unsigned char* __restrict__ ksc, long b_stride,
                                                                long sc_stride, int d, int H, int Tn, const RowCtl* __restrict__ ctl,
                                                                float* __restrict__ scratch, T* __restrict__ out_direct) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    __shared__ float s_sc[NR * kCrossRangeMax];
    __shared__ float s_red[2][NR][4];
    __shared__ float s_o[NR][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 2, c = lane & 3;            // 16 key rows x 4 chunks of 16 codes per wave-instruction
    const int h = blockIdx.y, m = blockIdx.z;
    const RowCtl rc = ctl[m];
    const int nkeys = rc.n_keys > 0 ? rc.n_keys : Tn;
    const int per = (nkeys + kCrossSplitD - 1) / kCrossSplitD;
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
            a = __builtin_fmaf(x[4 * j + 0], __builtin_amdgcn_cvt_f32_fp8((int)w[j], 0), a);
            a = __builtin_fmaf(x[4 * j + 1], __builtin_amdgcn_cvt_f32_fp8((int)w[j], 1), a);
            a = __builtin_fmaf(x[4 * j + 2], __builtin_amdgcn_cvt_f32_fp8((int)w[j], 2), a);
            a = __builtin_fmaf(x[4 * j + 3], __builtin_amdgcn_cvt_f32_fp8((int)w[j], 3), a);
        }
        return a;
    };
    // phase 1: scores; 64 keys per block iteration (4 waves x 16 rows)
    float mx[NR];
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int k_beg = (blockIdx.x * NR + g) * per, nk = max(0, min(nkeys, k_beg + per) - k_beg), nit = (nk + 63) / 64;
        float* sc = s_sc + g * kCrossRangeMax;
        mx[g] = -1e30f;
        for (int it = 0; it < nit; it += 4) {
            u32x4 kw[4];
            int ii[4];
            unsigned char eb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                ii[u] = (it + u) * 64 + wave * 16 + r;
                const int kk = k_beg + (ii[u] < nk ? ii[u] : 0);
                kw[u] = ld_stream<NT>((const u32x4*)(K + (long)kk * 64 + c * 16));
                eb[u] = KS[kk];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float a = dot16(kw[u], qv);
                a += dpp_mov<kDppXor1>(a);
                a += dpp_mov<kDppXor2>(a);
                a *= __builtin_bit_cast(float, (unsigned)eb[u] << 23);
                if (ii[u] < nk) {
                    if (c == 0) sc[ii[u]] = a;
                    mx[g] = fmaxf(mx[g], a);
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < NR; g++) {
        mx[g] = wave_max(mx[g]);
        if (lane == 0) s_red[0][g][wave] = mx[g];
    }
    __syncthreads();
    float sum[NR];
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int k_beg = (blockIdx.x * NR + g) * per, nk = max(0, min(nkeys, k_beg + per) - k_beg);
        float* sc = s_sc + g * kCrossRangeMax;
        mx[g] = fmaxf(fmaxf(s_red[0][g][0], s_red[0][g][1]), fmaxf(s_red[0][g][2], s_red[0][g][3]));
        float sm = 0.f;
        for (int i = tid; i < nk; i += 256) {
            const float p = (float)(T)__expf(sc[i] - mx[g]);
            sc[i] = p;
            sm += p;
        }
        sm = wave_sum(sm);
        if (lane == 0) s_red[1][g][wave] = sm;
    }
    __syncthreads();
    // phase 2: o[c*16 + e] += p[key] * 2^(ev - 127) * code
#pragma unroll
    for (int g = 0; g < NR; g++) {
        const int k_beg = (blockIdx.x * NR + g) * per, nk = max(0, min(nkeys, k_beg + per) - k_beg), nit = (nk + 63) / 64;
        const float* sc = s_sc + g * kCrossRangeMax;
        sum[g] = ((s_red[1][g][0] + s_red[1][g][1]) + s_red[1][g][2]) + s_red[1][g][3];
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
                vw[u] = ld_stream<NT>((const u32x4*)(V + (long)kk * 64 + c * 16));
                pw[u] = okk ? sc[i] * __builtin_bit_cast(float, (unsigned)VS[kk] << 23) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    acc[4 * j + 0] = __builtin_fmaf(pw[u], __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 0), acc[4 * j + 0]);
                    acc[4 * j + 1] = __builtin_fmaf(pw[u], __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 1), acc[4 * j + 1]);
                    acc[4 * j + 2] = __builtin_fmaf(pw[u], __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 2), acc[4 * j + 2]);
                    acc[4 * j + 3] = __builtin_fmaf(pw[u], __builtin_amdgcn_cvt_f32_fp8((int)vw[u][j], 3), acc[4 * j + 3]);
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
            for (int e = 0; e < 16; e++) s_o[g][wave][c * 16 + e] = acc[e];
        }
    }
    __syncthreads();
    if (tid < 64) {
        float os[NR];
#pragma unroll
        for (int g = 0; g < NR; g++) os[g] = ((s_o[g][0][tid] + s_o[g][1][tid]) + s_o[g][2][tid]) + s_o[g][3][tid];
        if constexpr (NR == kCrossSplitD) {
            out_direct[dec_wpack_off(m, h * 64 + tid, d)] = (T)cross_combine(mx, sum, os);
        } else {
            static_assert(NR == 1, "one range per workgroup, or all of them");
            float* part = scratch + ((long)(m * H + h) * kCrossSplitD + blockIdx.x) * kCrossPartD;
            part[2 + tid] = os[0];
            if (tid == 0) { part[0] = mx[0]; part[1] = sum[0]; }
        }
    }
}

template <typename T>
void launch_dec_cross_attention_f8(const float* qpart, int n_qpart, const float* qbias, float qscale, const unsigned char* kc, const unsigned char* ksc,
                                   long b_stride, long sc_stride, int d, int H, int Tn, const RowCtl* ctl, int M, float* scratch, T* out, hipStream_t st) {
    if ((Tn + kCrossSplitD - 1) / kCrossSplitD > kCrossRangeMax) throw Error(-1, "cross attention: n_audio_ctx too large");
    if (scratch) {
        dim3 grid(kCrossSplitD, H, M);
        if (cross_nt()) dec_cross_attn_q8_kernel<T, 1, true><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, ksc, b_stride, sc_stride, d, H, Tn, ctl, scratch, nullptr);
        else dec_cross_attn_q8_kernel<T, 1, false><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, ksc, b_stride, sc_stride, d, H, Tn, ctl, scratch, nullptr);
    } else {
        dim3 grid(1, H, M);
        if (cross_nt()) dec_cross_attn_q8_kernel<T, kCrossSplitD, true><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, ksc, b_stride, sc_stride, d, H, Tn, ctl, nullptr, out);
        else dec_cross_attn_q8_kernel<T, kCrossSplitD, false><<<grid, 256, 0, st>>>(qpart, n_qpart, qbias, qscale, kc, ksc, b_stride, sc_stride, d, H, Tn, ctl, nullptr, out);
    }
    SS_LAUNCH_CHECK();
}
template void launch_dec_cross_attention_f8<bf16>(const float*, int, const float*, float, const unsigned char*, const unsigned char*, long, long, int, int, int,
                                                  const RowCtl*, int, float*, bf16*, hipStream_t);
template void launch_dec_cross_attention_f8<f16>(const float*, int, const float*, float, const unsigned char*, const unsigned char*, long, long, int, int, int,
                                                 const RowCtl*, int, float*, f16*, hipStream_t);

}  // namespace ss
