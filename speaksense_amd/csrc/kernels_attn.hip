// Attention kernels for gfx950.
//  * encoder self-attention (non-causal, T=1500, dh=64): MFMA flash attention with the softmax held entirely in
//    registers.  S^T = K Q^T is computed with the operands swapped so a lane owns one query column: row max / sum are
//    in-lane plus two cross-group shuffles, and the exponentiated S^T registers ARE the B operand of the P·V MFMA
//    (the contraction index is permuted identically on the V^T side, so no LDS round trip and no transposes).
//  * decoder self-attention over the growing f16/bf16 KV cache and cross-attention over the 1500 encoder positions:
//    HBM-bound streaming kernels (one new token per row), split over T for occupancy.
// Replaces ggml's mul_mat(K,Q) -> soft_max(_ext) -> mul_mat(V,P) node chains in whisper.cpp's encoder / decoder graphs
// (/root/reference/resources/ggml-metal.metal:351-435 kernel_soft_max, :1229-1305 kernel_mul_mv_f16_f16).
#include <atomic>
#include <type_traits>
#include <cstdlib>

#include "kernels.h"
#include "wave_ops.h"

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct MfmaA;
template <> struct MfmaA<bf16> {
    typedef bf16x8 V8; typedef bf16x4 V4;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaA<f16> {
    typedef f16x8 V8; typedef f16x4 V4;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};


// ---------------------------------------------------------------------------------------------
// encoder flash attention, LDS-staged form: grid ceil(Tn / (64 QT)) * H * B workgroups of 256 threads; wave w owns 16 QT query rows.
// The first form of this kernel (tools/experiments/r02_variants/r02_variants.diff: enc_attn_kernel) let every wave pull its own K / V^T fragments from L2 in 32-key chunks: 8-B pieces of V^T rows and half lines of
// K rows, i.e. 3x the useful bytes through the CU's L1, re-read by every wave.  Here K and V^T tiles of 64 keys (whole 128-B lines)
// are DMA'd global -> LDS once per workgroup (global_load_lds, 3-stage ring, one s_barrier per chunk) and shared by the four waves.
//  * S^T = K Q^T with the K rows of a 32-key group permuted (MFMA row i of tile t <-> key (i>>2)*8 + (t&1)*4 + (i&3)), so the
//    exponentiated scores a lane holds are 8 CONSECUTIVE keys = the B operand of P.V with V^T fragments read as plain ds_read_b128.
//  * LDS rows are 128 B (8 chunks of 16 B); chunk c of row r sits at position c ^ ((r & 3) | ((r >> 3) & 1) << 2), applied on the DMA
//    source side: every 16-lane ds_read_b128 service group then touches 16 distinct 16-B slots for both the K and the V^T reads.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int attn_swz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <typename T, int QT, bool F8OUT = false>
__global__ __launch_bounds__(256, 2) void enc_attn_lds_kernel(const T* __restrict__ q, const T* __restrict__ k, long ld, const T* __restrict__ vT,
                                                              int Tpad, T* __restrict__ out, long ldo, int H, int Tn,
                                                              unsigned char* __restrict__ out8 = nullptr, unsigned char* __restrict__ out_sc = nullptr, long ldsc = 0) {
    typedef typename MfmaA<T>::V8 V8;
    typedef typename MfmaA<T>::V4 V4;
    constexpr int NS = 3, kStage = 16384;            // K tile 64 keys x 128 B, then V^T tile 64 dh rows x 128 B
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int frow = lane & 15, fg = lane >> 4;
    const int nqb = (Tn + 64 * QT - 1) / (64 * QT);
    int bid = blockIdx.x;
    {   // XCD-aware remap: the dispatcher places consecutive workgroup ids on different XCDs; each XCD gets a contiguous range of logical ids
        // so the query blocks of one (batch, head) share one L2 (K / V fetched once, not once per XCD)
        const int nwg = gridDim.x, qq = nwg / 8, rr = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int qb = bid % nqb, h = (bid / nqb) % H, b = bid / (nqb * H);
    const int q0 = qb * (64 * QT) + wave * (16 * QT);
    const long rowbase = (long)b * Tn;

    V8 qf[QT][2];
#pragma unroll
    for (int qi = 0; qi < QT; qi++) {
        int qr = q0 + qi * 16 + frow;
        if (qr > Tn - 1) qr = Tn - 1;
        const T* p = q + (rowbase + qr) * ld + h * 64 + fg * 8;
        qf[qi][0] = *(const V8*)p;
        qf[qi][1] = *(const V8*)(p + 32);
    }
    // DMA sources: wave w stages 8-row blocks {2w, 2w+1} of the K tile and of the V^T tile; lane L -> row L/8, position L%8
    const int srow = lane >> 3, spos = lane & 7;
    const T* ksrc[2]; const T* vsrc[2];
    int krow[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int r = (2 * wave + j) * 8 + srow;
        const int c = spos ^ attn_swz(r);
        krow[j] = r;
        ksrc[j] = k + rowbase * ld + h * 64 + c * 8;
        vsrc[j] = vT + ((long)(b * H + h) * 64 + r) * Tpad + c * 8;
    }
    auto stage = [&](int buf, int key0) {
        char* base = smem_a + buf * kStage;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int kr = key0 + krow[j];
            if (kr > Tn - 1) kr = Tn - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ksrc[j] + (long)kr * ld),
                                             (__attribute__((address_space(3))) void*)(base + (2 * wave + j) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vsrc[j] + key0),
                                             (__attribute__((address_space(3))) void*)(base + 8192 + (2 * wave + j) * 1024), 16, 0, 0);
        }
    };
    f32x4 o[4][QT];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < QT; j++) o[i][j] = (f32x4){0, 0, 0, 0};
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int j = 0; j < QT; j++) { mrun[j] = -1e30f; lrun[j] = 0.f; }
    const float c2 = 0.125f * 1.44269504088896341f;  // 1/sqrt(64) * log2(e)
    const int nchunk = (Tn + 63) / 64;

    // fragment read offsets (bytes within a stage)
    int koff[4][2], voff[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; kt++) {
        const int key = (kt >> 1) * 32 + (frow >> 2) * 8 + (kt & 1) * 4 + (frow & 3);
#pragma unroll
        for (int hh = 0; hh < 2; hh++) koff[kt][hh] = key * 128 + (((hh * 4 + fg) ^ attn_swz(key)) * 16);
    }
#pragma unroll
    for (int dt = 0; dt < 4; dt++) {
        const int r = dt * 16 + frow;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) voff[dt][ks] = 8192 + r * 128 + (((ks * 4 + fg) ^ attn_swz(r)) * 16);
    }

    // one 64-key chunk; TAIL (only the last chunk of a Tn that is not a multiple of 64) masks the keys beyond Tn
    auto chunk = [&](auto tail_tag, int kc) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (kc + 1 < nchunk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // chunk kc has landed for every wave; everyone is done reading chunk kc-1
        if (kc + 2 < nchunk) stage((kc + 2) % NS, (kc + 2) * 64);
        const char* base = smem_a + (kc % NS) * kStage;
        V8 kf[4][2], vf[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; kt++) { kf[kt][0] = *(const V8*)(base + koff[kt][0]); kf[kt][1] = *(const V8*)(base + koff[kt][1]); }
#pragma unroll
        for (int dt = 0; dt < 4; dt++) { vf[dt][0] = *(const V8*)(base + voff[dt][0]); vf[dt][1] = *(const V8*)(base + voff[dt][1]); }
        const int key0 = kc * 64;
#pragma unroll
        for (int qi = 0; qi < QT; qi++) {
            f32x4 s[4];
#pragma unroll
            for (int kt = 0; kt < 4; kt++) {
                f32x4 a = (f32x4){0, 0, 0, 0};
                a = MfmaA<T>::mma(kf[kt][0], qf[qi][0], a);
                a = MfmaA<T>::mma(kf[kt][1], qf[qi][1], a);
                s[kt] = a;
            }
            float mx = -1e30f;
#pragma unroll
            for (int kt = 0; kt < 4; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    if constexpr (TAIL) { if (key0 + (kt >> 1) * 32 + fg * 8 + (kt & 1) * 4 + r >= Tn) s[kt][r] = -1e30f; }
                    mx = fmaxf(mx, s[kt][r]);
                }
            mx = rows_max(mx);
            mx *= c2;
            if (__any(mx > mrun[qi])) {
                const float mnew = fmaxf(mrun[qi], mx);
                const float alpha = __builtin_amdgcn_exp2f(mrun[qi] - mnew);
                mrun[qi] = mnew;
                lrun[qi] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; dt++) o[dt][qi] *= alpha;
            }
            const float mcur = mrun[qi];
            float psum = 0.f;
            V8 pf[2];
#pragma unroll
            for (int kt = 0; kt < 4; kt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], c2, -mcur));
                    psum += pv;
                    pf[kt >> 1][(kt & 1) * 4 + r] = (T)pv;
                }
            lrun[qi] += psum;
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                o[dt][qi] = MfmaA<T>::mma(vf[dt][0], pf[0], o[dt][qi]);
                o[dt][qi] = MfmaA<T>::mma(vf[dt][1], pf[1], o[dt][qi]);
            }
        }
    };
    stage(0, 0);
    if (nchunk > 1) stage(1, 64);
    const int nfull = (Tn % 64) ? nchunk - 1 : nchunk;
    for (int kc = 0; kc < nfull; kc++) chunk(std::false_type{}, kc);
    if (nfull < nchunk) chunk(std::true_type{}, nchunk - 1);
#pragma unroll
    for (int qi = 0; qi < QT; qi++) {
        const float l = rows_sum(lrun[qi]);
        const float inv = 1.0f / l;
        const int qr = q0 + qi * 16 + frow;
        if constexpr (F8OUT) {
            // fp8 engine: output rounded to T (as the f16 engine stores it), then e4m3 with one exponent byte per (row, head): the 64 columns of
            // this head's row are this lane's 16 values and those of the lanes frow + 16 / 32 / 48
            float v[4][4], amax = 0.f;
#pragma unroll
            for (int dt = 0; dt < 4; dt++)
#pragma unroll
                for (int r = 0; r < 4; r++) { v[dt][r] = (float)(T)(o[dt][qi][r] * inv); amax = fmaxf(amax, fabsf(v[dt][r])); }
            amax = rows_max(amax);
            const int eb = e8m0_for_amax(amax);
            const float sc = pow2_neg_of_e8m0(eb);
            if (qr < Tn) {
#pragma unroll
                for (int dt = 0; dt < 4; dt++)
                    *(unsigned*)(out8 + (rowbase + qr) * ldo + h * 64 + dt * 16 + fg * 4) = pack_e4m3x4(v[dt][0] * sc, v[dt][1] * sc, v[dt][2] * sc, v[dt][3] * sc);
                if (fg == 0) out_sc[f8_scale_index(rowbase + qr, h, ldsc)] = (unsigned char)eb;
            }
        } else if (qr < Tn) {
#pragma unroll
            for (int dt = 0; dt < 4; dt++) {
                V4 ov;
#pragma unroll
                for (int r = 0; r < 4; r++) ov[r] = (T)(o[dt][qi][r] * inv);
                *(V4*)(out + (rowbase + qr) * ldo + h * 64 + dt * 16 + fg * 4) = ov;
            }
        }
    }
}

template <typename T, int QT>
static void launch_enc_attention_lds(const T* q, const T* k, long ld, const T* vT, int Tpad, T* out, long ldo, int B, int H, int Tn, hipStream_t st) {
    if (Tpad < (Tn + 63) / 64 * 64) throw Error(-1, "enc_attention: V^T rows must be padded to a multiple of 64 keys");
    static std::atomic<uint64_t> attr{0};
    once_per_device(attr, [] { SS_HIP(hipFuncSetAttribute((const void*)enc_attn_lds_kernel<T, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384)); });
    dim3 grid(((Tn + 64 * QT - 1) / (64 * QT)) * H * B);
    enc_attn_lds_kernel<T, QT><<<grid, 256, 3 * 16384, st>>>(q, k, ld, vT, Tpad, out, ldo, H, Tn); SS_LAUNCH_CHECK();
}

// V [B*Tn][ld] (head h at column h*64) -> V^T [B][H][64][Tpad], 64 x 64 tiles through LDS; key columns t >= Tn of the last tile are written as zeros.
// Round 5 experiment behind SS_VT_GEMM=0 (not the default): the V projection rides in the Q/K GEMM as plain rows and is transposed here -- the
// transposing GEMM epilogue stores 8-byte pieces into 16 different rows per instruction and runs at 0.25 of the MFMA pipe against 0.41 for the plain
// store (profiles/r05_ag_pmc_compute_encoder_kernels.txt).  Same bits; 1.4 % slower end to end (profiles/r05_ah_vt_gemm_ab.txt).
template <typename T>
__global__ __launch_bounds__(256) void v_transpose_kernel(const T* __restrict__ v, long ld, T* __restrict__ vT, int Tpad, int H, int Tn) {
    typedef typename MfmaA<T>::V8 V8;
    __shared__ T tile[64][64 + 8];          // +8 elements: rows 144 B apart, the column gather below walks 16-B slots without a common bank
    const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (tid >> 3) + i * 32, c = tid & 7;          // row = key, 8 chunks of 8 elements
        const int t = t0 + r;
        V8 x;
        if (t < Tn) x = *(const V8*)(v + ((long)b * Tn + t) * ld + h * 64 + c * 8);
        else { for (int e = 0; e < 8; e++) x[e] = (T)0.0f; }
        *(V8*)&tile[r][c * 8] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int j = (tid >> 3) + i * 32, c = tid & 7;          // row = head column j, chunk of 8 consecutive keys
        V8 y;
#pragma unroll
        for (int e = 0; e < 8; e++) y[e] = tile[c * 8 + e][j];
        *(V8*)(vT + ((long)(b * H + h) * 64 + j) * Tpad + t0 + c * 8) = y;
    }
}
template <typename T>
void launch_v_transpose(const T* v, long ld, T* vT, int Tpad, int B, int H, int Tn, hipStream_t st) {
    if (Tpad < (Tn + 63) / 64 * 64 || Tpad % 8 || ld % 8) throw Error(-1, "v_transpose: V^T rows must be padded to a multiple of 64 keys");
    v_transpose_kernel<T><<<dim3((Tn + 63) / 64, H, B), 256, 0, st>>>(v, ld, vT, Tpad, H, Tn); SS_LAUNCH_CHECK();
}
template void launch_v_transpose<bf16>(const bf16*, long, bf16*, int, int, int, int, hipStream_t);
template void launch_v_transpose<f16>(const f16*, long, f16*, int, int, int, int, hipStream_t);

template <typename T>
void launch_enc_attention(const T* q, const T* k, long ld, const T* vT, int Tpad, T* out, long ldo, int B, int H, int Tn, hipStream_t st) {
    launch_enc_attention_lds<T, 4>(q, k, ld, vT, Tpad, out, ldo, B, H, Tn, st);   // 4 query tiles per wave; 2 and 3 measured the same (VALU-bound)
}
template <typename T>
void launch_enc_attention_f8(const T* q, const T* k, long ld, const T* vT, int Tpad, unsigned char* out8, long ldo, unsigned char* out_scale, long ldsc, int B, int H, int Tn, hipStream_t st) {
    if (Tpad < (Tn + 63) / 64 * 64) throw Error(-1, "enc_attention: V^T rows must be padded to a multiple of 64 keys");
    static std::atomic<uint64_t> attr{0};
    once_per_device(attr, [] { SS_HIP(hipFuncSetAttribute((const void*)enc_attn_lds_kernel<T, 4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384)); });
    dim3 grid(((Tn + 255) / 256) * H * B);
    enc_attn_lds_kernel<T, 4, true><<<grid, 256, 3 * 16384, st>>>(q, k, ld, vT, Tpad, nullptr, ldo, H, Tn, out8, out_scale, ldsc); SS_LAUNCH_CHECK();
}
template void launch_enc_attention_f8<bf16>(const bf16*, const bf16*, long, const bf16*, int, unsigned char*, long, unsigned char*, long, int, int, int, hipStream_t);
template void launch_enc_attention_f8<f16>(const f16*, const f16*, long, const f16*, int, unsigned char*, long, unsigned char*, long, int, int, int, hipStream_t);
template void launch_enc_attention<bf16>(const bf16*, const bf16*, long, const bf16*, int, bf16*, long, int, int, int, hipStream_t);
template void launch_enc_attention<f16>(const f16*, const f16*, long, const f16*, int, f16*, long, int, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// decoder self-attention: grid (H, M), one wave per (row, head); n_kv = pos + 1 <= 448
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(64) void dec_self_attn_kernel(const T* __restrict__ q, const T* __restrict__ kcache, const T* __restrict__ vcache,
                                                           long slot_stride, int d, const RowCtl* __restrict__ ctl, T* __restrict__ out) {
    __builtin_amdgcn_s_setprio(3);   // a chain kernel: see SS_CHAIN_PRIO_STMT in kernels_decode.hip
    typedef typename MfmaA<T>::V8 V8;
    __shared__ float s_p[448 + 64];
    const int lane = threadIdx.x, h = blockIdx.x, m = blockIdx.y;
    const int r = lane >> 3, c = lane & 7;   // 8 keys x 8 column chunks per wave-instruction (one 128-B row per 8 lanes)
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
    const int nit = (n_kv + 7) / 8;
    float mx = -1e30f;
    for (int it = 0; it < nit; it += 4) {
        V8 kv[4];
        int kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            kk[u] = (it + u) * 8 + r;
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
    __syncthreads();
    float sum = 0.f;
    for (int key = lane; key < n_kv; key += 64) {
        const float p = (float)(T)__expf(s_p[key] - mx);
        s_p[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < nit; it += 4) {
        V8 vv[4];
        float pw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int key = (it + u) * 8 + r;
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
    for (int e = 0; e < 8; e++) {
        acc[e] = sum_stride8(acc[e]);
    }
    if (r == 0) {
        const float inv = 1.0f / sum;
        V8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = (T)(acc[e] * inv);
        *(V8*)(out + dec_wpack_off(m, h * 64 + c * 8, d)) = o;   // fragment-major (kernels.h): the out-projection GEMV's B operand
    }
}

template <typename T>
void launch_dec_self_attention(const T* q, const T* kcache, const T* vcache, long slot_stride, int d, int H, const RowCtl* ctl, int M, T* out,
                               hipStream_t st) {
    dim3 grid(H, M);
    dec_self_attn_kernel<T><<<grid, 64, 0, st>>>(q, kcache, vcache, slot_stride, d, ctl, out); SS_LAUNCH_CHECK();
}
template void launch_dec_self_attention<bf16>(const bf16*, const bf16*, const bf16*, long, int, int, const RowCtl*, int, bf16*, hipStream_t);
template void launch_dec_self_attention<f16>(const f16*, const f16*, const f16*, long, int, int, const RowCtl*, int, f16*, hipStream_t);

// (the flash-decoding combine of the cross-attention partials lives beside the kernel that produces them: kernels_decode.hip)

}  // namespace ss
