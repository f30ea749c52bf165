// Decoder pass, 16..64 rows: the two attention blocks of a layer with their row-local projections folded in (gfx950).
//
// kernels_decode.hip runs a decoder layer as 11 launches; four of them are d x d projections (self-attention out, cross query, cross out) and
// the reduce + LayerNorm between them, each a whole-chip grid that lives for 5-10 us and moves 3 MB.  A (token row, head) workgroup of an
// attention kernel can do that work itself, because all of it is row-local once the head's 64 columns are fixed:
//
//   dec_self_attn_wo_kernel   self-attention of (row, head) over the growing KV cache, then the head's slice of the out-projection:
//                             part[h][row][n] = sum_j att[j] Wo[n][64 h + j]  (f32, one partial per head; summed in head order by the consumer)
//   dec_cross_fused_kernel    x = x + bo + sum_h part[h] -> LayerNorm -> q = (ln . Wcq[64 h + j] + bcq) dh^-1/4 -> attention over the 1500
//                             encoder positions -> the head's slice of the cross out-projection, again as per-head partials
//
// so a layer is 7 launches: reduce+LN1, QKV, [self-attn + Wo], [LNc + Wcq + cross-attn + Wco], reduce+LN2 (over the H head partials), FC1, FC2.
// The projections' weights are re-read per (row, head) workgroup -- from L2: workgroups are handed out so that an XCD owns whole heads
// (units_of_block), i.e. each XCD touches 2-4 heads' 160 KB slices of a weight.  Arithmetic as in the GEMV kernels: T x T products, f32
// accumulation on the MFMA (the row is broadcast to all 16 B columns), q / attention output / LayerNorm output rounded to T at the points
// ggml rounds them.  Replaces the same ggml nodes as kernels_decode.hip (/root/reference/resources/ggml-metal.metal:1307-1363 mul_mv, :571-621 norm,
// :351-435 soft_max); reached through engine.cpp fused_body when rows x heads >= 320 (the benchmarked 27-32-row passes).
#include "kernels.h"
#include "wave_ops.h"

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct MfmaF;
template <> struct MfmaF<bf16> {
    typedef bf16x8 V8;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaF<f16> {
    typedef f16x8 V8;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

// (row, head) units handed out so that an XCD (workgroup id % 8, observed placement, used for speed only) owns a contiguous range of the
// head-major order u = h M + m: all rows of a head -- which share that head's weight slices -- run on one or two XCDs.
__device__ __forceinline__ bool unit_of_block(int bid, int M, int H, int* h, int* m) {
    const int n_units = M * H, per = (n_units + 7) >> 3;
    const int u = (bid & 7) * per + (bid >> 3);
    if ((bid >> 3) >= per || u >= n_units) return false;
    *h = u / M; *m = u - *h * M;
    return true;
}
static inline int units_grid(int M, int H) { return 8 * ((M * H + 7) / 8); }

// part[n] = sum_{j < 64} a[j] W[n][col0 + j] for n in [0, N): one head's slice of a d x d projection applied to one row.  `a` = 64 T in LDS.
// MFMA 16x16x32 with the 16 weight rows as A and the row broadcast to all 16 columns of B; the lanes of column 0 (frow == 0) hold the sums.
// The weights do not depend on the row: the first batch of fragments (kHsBatch tiles per wave) is fetched by head_slice_prefetch BEFORE the
// attention loop that produces `a`, so its L2 round trip hides there; the remaining tiles follow in batches of the same depth.
constexpr int kHsBatch = 8;
template <typename T> struct HsFrag { typename MfmaF<T>::V8 w[kHsBatch][2]; };
template <typename T>
__device__ __forceinline__ void head_slice_load(HsFrag<T>& f, const T* __restrict__ W, long ldw, int col0, int n_tiles, int t0, int lane) {
    typedef typename MfmaF<T>::V8 V8;
    const int frow = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int u = 0; u < kHsBatch; u++) {
        const int t = t0 + 4 * u;
        const T* wp = W + (long)((t < n_tiles ? t : t0) * 16 + frow) * ldw + col0 + fg * 8;
        f.w[u][0] = *(const V8*)wp; f.w[u][1] = *(const V8*)(wp + 32);
    }
}
template <typename T>
__device__ __forceinline__ void head_slice_gemv(HsFrag<T>& f, const T* __restrict__ W, long ldw, int col0, int N, const T* a_lds, float* __restrict__ out, int wave, int lane) {
    typedef typename MfmaF<T>::V8 V8;
    const int frow = lane & 15, fg = lane >> 4;
    const V8 b0 = *(const V8*)(a_lds + fg * 8), b1 = *(const V8*)(a_lds + 32 + fg * 8);
    const int n_tiles = N >> 4;
    for (int t0 = wave; t0 < n_tiles; t0 += 4 * kHsBatch) {    // `f` already holds the batch that starts at t0 = wave
        if (t0 != wave) head_slice_load<T>(f, W, ldw, col0, n_tiles, t0, lane);
#pragma unroll
        for (int u = 0; u < kHsBatch; u++) {
            const int t = t0 + 4 * u;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = MfmaF<T>::mma(f.w[u][0], b0, acc);
            acc = MfmaF<T>::mma(f.w[u][1], b1, acc);
            if (frow == 0 && t < n_tiles) *(f32x4*)(out + t * 16 + fg * 4) = acc;     // D[n = 4 fg + r][column 0]
        }
    }
}

// ---------------------------------------------------------------------------------------------
// self-attention of one (row, head) + the head's slice of the out-projection.  256 threads: the four waves split the keys 32 at a time
// (8 keys x 128 B per wave-instruction, 4 loads in flight per lane), as the cross-attention kernel does.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dec_self_attn_wo_kernel(const T* __restrict__ q, const T* __restrict__ kcache, const T* __restrict__ vcache, long slot_stride,
                                                               int d, int H, int M, const RowCtl* __restrict__ ctl, const T* __restrict__ Wo,
                                                               float* __restrict__ part /* [H][kPartRows][d] */) {
    typedef typename MfmaF<T>::V8 V8;
    __shared__ float s_p[448 + 64];
    __shared__ float s_red[8];
    __shared__ float s_o[4][64];
    __shared__ __attribute__((aligned(16))) T s_att[64];
    int h, m;
    if (!unit_of_block(blockIdx.x, M, H, &h, &m)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane >> 3, c = lane & 7;
    const RowCtl rc = ctl[m];
    const int n_kv = rc.pos + 1;
    const T* K = kcache + (long)rc.slot * slot_stride + h * 64 + c * 8;
    const T* V = vcache + (long)rc.slot * slot_stride + h * 64 + c * 8;
    float qv[8];
    {
        const V8 t = *(const V8*)(q + (long)m * d + h * 64 + c * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) qv[e] = (float)t[e];
    }
    HsFrag<T> wfr;
    head_slice_load<T>(wfr, Wo, d, h * 64, d >> 4, wave, lane);      // hidden under the attention loops below
    const int nit = (n_kv + 31) / 32;
    float mx = -1e30f;
    for (int it = 0; it < nit; it += 4) {
        V8 kv[4];
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            kk[u] = (it + u) * 32 + wave * 8 + r;
            kv[u] = *(const V8*)(K + (long)(kk[u] < n_kv ? kk[u] : 0) * d);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) a += qv[e] * (float)kv[u][e];
            a = sum_lanes8(a);
            if (kk[u] < n_kv) {
                if (c == 0) s_p[kk[u]] = a;
                mx = fmaxf(mx, a);
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) s_red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float sum = 0.f;
    for (int key = tid; key < n_kv; key += 256) {
        const float p = (float)(T)__expf(s_p[key] - mx);
        s_p[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < nit; it += 4) {
        V8 vv[4];
        float pw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int key = (it + u) * 32 + wave * 8 + r;
            const bool okk = key < n_kv;
            vv[u] = *(const V8*)(V + (long)(okk ? key : 0) * d);
            pw[u] = okk ? s_p[key] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pw[u] * (float)vv[u][e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = sum_stride8(acc[e]);
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) s_o[wave][c * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) s_att[tid] = (T)((s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid]) / sum);
    __syncthreads();
    head_slice_gemv<T>(wfr, Wo, d, h * 64, d, s_att, part + ((long)h * kPartRows + m) * d, wave, lane);
}

template <typename T>
void launch_dec_self_attention_wo(const T* q, const T* kcache, const T* vcache, long slot_stride, int d, int H, const RowCtl* ctl, int M, const T* Wo, float* part,
                                  hipStream_t st) {
    if (d != H * 64 || d % 64 || M < 1 || M > kPartRows) throw Error(-1, "dec_self_attention_wo: bad shape");
    dec_self_attn_wo_kernel<T><<<units_grid(M, H), 256, 0, st>>>(q, kcache, vcache, slot_stride, d, H, M, ctl, Wo, part); SS_LAUNCH_CHECK();
}
template void launch_dec_self_attention_wo<bf16>(const bf16*, const bf16*, const bf16*, long, int, int, const RowCtl*, int, const bf16*, float*, hipStream_t);
template void launch_dec_self_attention_wo<f16>(const f16*, const f16*, const f16*, long, int, int, const RowCtl*, int, const f16*, float*, hipStream_t);

// ---------------------------------------------------------------------------------------------
// residual + LayerNorm + cross query + cross-attention + cross out-projection slice for one (row, head).  NC = float4 chunks of the row
// per thread (d <= 1024: 1, d <= 2048: 2).
// ---------------------------------------------------------------------------------------------
template <typename T, int NC>
__global__ __launch_bounds__(256) void dec_cross_fused_kernel(DecCrossFusedDesc g) {
    typedef typename MfmaF<T>::V8 V8;
    __shared__ float s_sc[1536 + 128];
    __shared__ float s_red[8];
    __shared__ float s_o[4][64];
    __shared__ __attribute__((aligned(16))) T s_ln[2048];
    __shared__ __attribute__((aligned(16))) T s_q[64];
    int h, m;
    if (!unit_of_block(blockIdx.x, g.M, g.H, &h, &m)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int d = g.d, Tn = g.Tn;

    // the cross-query weights of this head do not depend on the row: their first kQPre k-steps are fetched before the residual / LayerNorm
    // prologue, whose own loads then overlap them
    constexpr int kQPre = 10;
    const T* wq = (const T*)g.Wq + (long)(h * 64 + wave * 16 + frow) * d + fg * 8;
    const int nk = d >> 5;
    V8 wpre[kQPre];
#pragma unroll
    for (int u = 0; u < kQPre; u++) wpre[u] = *(const V8*)(wq + (long)(u < nk ? u : 0) * 32);

    // ---- x = x_in + bias_prev + sum_p parts[p] (head order), LayerNorm over the row -> s_ln (T) ----
    {
        f32x4 v[NC];
        int cc[NC];
        bool ok[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) { const int c4 = (i * 256 + tid) * 4; ok[i] = c4 < d; cc[i] = ok[i] ? c4 : 0; }
        f32x4 ww[NC], bb[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) { ww[i] = *(const f32x4*)(g.ln_w + cc[i]); bb[i] = *(const f32x4*)(g.ln_b + cc[i]); }
        const float* xr = g.x_in + (long)m * d;
#pragma unroll
        for (int i = 0; i < NC; i++) v[i] = *(const f32x4*)(xr + cc[i]) + *(const f32x4*)(g.bias_prev + cc[i]);
        constexpr int PB = NC == 1 ? 10 : 5;                // partial rows in flight; unused slots re-read slot p0 with weight 0
        for (int p0 = 0; p0 < g.n_parts; p0 += PB) {
            f32x4 t[PB][NC];
            float wgt[PB];
#pragma unroll
            for (int u = 0; u < PB; u++) {
                const int p = p0 + u < g.n_parts ? p0 + u : p0;
                wgt[u] = p0 + u < g.n_parts ? 1.f : 0.f;
                const float* pr = g.parts + ((long)p * kPartRows + m) * d;
#pragma unroll
                for (int i = 0; i < NC; i++) t[u][i] = *(const f32x4*)(pr + cc[i]);
            }
#pragma unroll
            for (int u = 0; u < PB; u++)
#pragma unroll
                for (int i = 0; i < NC; i++) v[i] += t[u][i] * wgt[u];
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NC; i++) {
            if (!ok[i]) v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        if (h == 0) {       // one workgroup of the row writes the updated residual stream
#pragma unroll
            for (int i = 0; i < NC; i++) if (ok[i]) *(f32x4*)(g.x_out + (long)m * d + cc[i]) = v[i];
        }
        sum = wave_sum(sum);
        if (lane == 0) s_red[wave] = sum;
        __syncthreads();
        const float mean = (s_red[0] + s_red[1] + s_red[2] + s_red[3]) / d;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NC; i++)
#pragma unroll
            for (int e = 0; e < 4; e++) { v[i][e] = ok[i] ? v[i][e] - mean : 0.f; sq += v[i][e] * v[i][e]; }
        sq = wave_sum(sq);
        if (lane == 0) s_red[4 + wave] = sq;
        __syncthreads();
        const float rstd = 1.0f / sqrtf((s_red[4] + s_red[5] + s_red[6] + s_red[7]) / d + 1e-5f);
#pragma unroll
        for (int i = 0; i < NC; i++) {
            if (ok[i]) {
#pragma unroll
                for (int e = 0; e < 4; e++) s_ln[cc[i] + e] = (T)(v[i][e] * rstd * ww[i][e] + bb[i][e]);
            }
        }
    }
    __syncthreads();

    // ---- q[j] = (sum_k ln[k] Wcq[64 h + j][k] + bcq) * qscale, rounded to T: wave w owns the 16 rows j of tile w ----
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < kQPre; u++) {
            if (u < nk) acc = MfmaF<T>::mma(wpre[u], *(const V8*)(s_ln + u * 32 + fg * 8), acc);
        }
        for (int k0 = kQPre; k0 < nk; k0 += kQPre) {       // the k-steps past the prefetched ones, kQPre loads in flight
#pragma unroll
            for (int u = 0; u < kQPre; u++) wpre[u] = *(const V8*)(wq + (long)(k0 + u < nk ? k0 + u : k0) * 32);
#pragma unroll
            for (int u = 0; u < kQPre; u++) {
                if (k0 + u < nk) acc = MfmaF<T>::mma(wpre[u], *(const V8*)(s_ln + (k0 + u) * 32 + fg * 8), acc);
            }
        }
        if (frow == 0) {
#pragma unroll
            for (int r4 = 0; r4 < 4; r4++) {
                const int j = wave * 16 + fg * 4 + r4;
                s_q[j] = (T)((acc[r4] + g.bq[h * 64 + j]) * g.qscale);
            }
        }
    }
    __syncthreads();

    // ---- attention over the Tn encoder positions (the unsplit form of dec_cross_attn_q_kernel) ----
    const int r = lane >> 3, c = lane & 7;
    const RowCtl rc = g.ctl[m];
    const T* K = (const T*)g.kc + (long)rc.cross * g.b_stride + (long)h * Tn * 64;
    const T* V = (const T*)g.vc + (long)rc.cross * g.b_stride + (long)h * Tn * 64;
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) qv[e] = (float)s_q[c * 8 + e];
    float mx = -1e30f;
    const int nit = (Tn + 31) / 32;
    for (int it = 0; it < nit; it += 4) {
        V8 kv[4];
        int ii[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ii[u] = (it + u) * 32 + wave * 8 + r;
            kv[u] = *(const V8*)(K + (long)(ii[u] < Tn ? ii[u] : 0) * 64 + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) a += qv[e] * (float)kv[u][e];
            a = sum_lanes8(a);
            if (ii[u] < Tn) {
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
    for (int i = tid; i < Tn; i += 256) {
        const float p = (float)(T)__expf(s_sc[i] - mx);
        s_sc[i] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) s_red[4 + wave] = sum;
    __syncthreads();
    sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    HsFrag<T> wfr;
    head_slice_load<T>(wfr, (const T*)g.Wo, d, h * 64, d >> 4, wave, lane);      // hidden under the P.V loop
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < nit; it += 4) {
        V8 vv[4];
        float pw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = (it + u) * 32 + wave * 8 + r;
            const bool okk = i < Tn;
            vv[u] = *(const V8*)(V + (long)(okk ? i : 0) * 64 + c * 8);
            pw[u] = okk ? s_sc[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pw[u] * (float)vv[u][e];
    }
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = sum_stride8(acc[e]);
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) s_o[wave][c * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < 64) s_q[tid] = (T)((s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid]) / sum);     // s_q now holds the attention output of this head
    __syncthreads();
    head_slice_gemv<T>(wfr, (const T*)g.Wo, d, h * 64, d, s_q, g.part_out + ((long)h * kPartRows + m) * d, wave, lane);
}

template <typename T>
void launch_dec_cross_fused(const DecCrossFusedDesc& g, hipStream_t st) {
    if (g.d != g.H * 64 || g.d > 2048 || g.Tn > 1536 || g.M < 1 || g.M > kPartRows || g.n_parts < 1) throw Error(-1, "dec_cross_fused: bad shape");
    const int grid = units_grid(g.M, g.H);
    if (g.d <= 1024) { dec_cross_fused_kernel<T, 1><<<grid, 256, 0, st>>>(g); SS_LAUNCH_CHECK(); }
    else { dec_cross_fused_kernel<T, 2><<<grid, 256, 0, st>>>(g); SS_LAUNCH_CHECK(); }
}
template void launch_dec_cross_fused<bf16>(const DecCrossFusedDesc&, hipStream_t);
template void launch_dec_cross_fused<f16>(const DecCrossFusedDesc&, hipStream_t);

}  // namespace ss
