// EXPERIMENT, NOT IN THE BUILD: deferred-epilogue GEMM measured 2x slower than gemm256_kernel (profiles/r02_l_gemm_deferred_epilogue_ab.txt).
// Kept for the record of what was tried; it compiled against kernels.h with a `launch_gemm_de<T>` declaration added to it.
// MFMA GEMM with a DEFERRED epilogue for gfx950 (round 2; opt-in while it is being measured: SS_GEMM_DE=1).
//
// gemm256_kernel (kernels_gemm.hip) spends 19-28 % of every f16-output tile in its epilogue (bias / GELU / convert / 32 stores per thread)
// and 6 % in the DMA prologue of the next tile, with the matrix pipes idle.  Here the tile sequence of a persistent workgroup is ONE
// continuous stream of k-steps: the LDS ring keeps rolling across tile boundaries (the DMA cursor runs three steps ahead of the MFMA cursor
// and simply moves on to the next tile's operands), and the finished tile's accumulators are parked in a second register set whose
// epilogue is emitted one 16 x 16 sub-tile per k-step of the NEXT tile (4 values: bias, GELU, convert, one 8-byte store).
//   tile 256 (m) x 128 (n) x 32 (k), 8 waves = 2 (n) x 4 (m), wave tile 64 n x 64 m = 4 x 4 MFMA 16x16x32 tiles:
//   64 accumulator VGPRs + 64 parked + 2 fragment sets of 32 -> fits the 256-register budget of two waves per SIMD;
//   4-stage ring of 24 KB (X 256 rows + W 128 rows of 64 B), LDS-DMA 16 B per lane, same source-side XOR swizzle as gemm256_kernel.
// vmcnt retires in order and counts the epilogue stores too, so the wait for "stage s has landed" allows exactly the memory operations
// this wave issued AFTER that stage's DMA to stay outstanding: a running count of issued operations, marked per ring slot.
#include <cstdlib>

#include "kernels.h"

namespace ss {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct MfmaE;
template <> struct MfmaE<bf16> {
    typedef bf16x8 V8; typedef bf16x4 V4;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaE<f16> {
    typedef f16x8 V8; typedef f16x4 V4;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
__device__ __forceinline__ float gelu_e(float x) {
    const float u = 0.79788456080286535588f * x * (1.0f + 0.044715f * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}
template <typename T> __device__ __forceinline__ float gelu_round_e(float x, int on);
template <> __device__ __forceinline__ float gelu_round_e<bf16>(float x, int) { return x; }
template <> __device__ __forceinline__ float gelu_round_e<f16>(float x, int on) { return on ? (float)(f16)x : x; }

// the rasterisation of gemm256_kernel (bijective XCD remap + column groups of ~5 n-panels, m fastest inside a group)
__device__ __forceinline__ void tile_of_block_e(int bid, int nbm, int nbn, int* mb, int* nb) {
    const int nwg = nbm * nbn;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int GN = nbn;
    if (nbn > 8) {
        GN = 10;    // 10 panels of 128 columns = the 5 panels of 256 the wider kernel groups
        if (nbn % 10 != 0) { if (nbn % 12 == 0) GN = 12; else if (nbn % 8 == 0) GN = 8; else if (nbn % 5 == 0) GN = 5; }
    }
    const int per_group = nbm * GN, full = nbn / GN;
    int g = bid / per_group, rem = bid - g * per_group, gn = GN;
    if (g >= full) { g = full; rem = bid - full * per_group; gn = nbn - full * GN; }
    *mb = rem / gn;
    *nb = g * GN + rem % gn;
}

constexpr int DTM = 256, DTN = 128, DTK = 32, DNST = 4;
constexpr int kDStage = (DTM + DTN) * DTK * 2;   // 24 KB
constexpr int kDLds = DNST * kDStage;            // 96 KB

template <typename T>
__device__ __forceinline__ void glds16e(const char* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n operations of this wave may remain outstanding (n is wave-uniform)
__device__ __forceinline__ void wait_vm_dyn(int n) {
    switch (n) {
        case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; case 7: wait_vm<7>(); break;
        case 8: wait_vm<8>(); break; case 9: wait_vm<9>(); break; case 10: wait_vm<10>(); break; case 11: wait_vm<11>(); break;
        case 12: wait_vm<12>(); break; case 13: wait_vm<13>(); break; case 14: wait_vm<14>(); break; default: wait_vm<15>(); break;
    }
}

template <typename T, int KIND>
__global__ __launch_bounds__(512, 2) void gemm_de_kernel(GemmDesc g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename MfmaE<T>::V8 V8;
    typedef typename MfmaE<T>::V4 V4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wm = wave & 3;
    const int frow = lane & 15, fg = lane >> 4;
    const int nbn = g.N / DTN, nbm = (g.M + DTM - 1) / DTM, ntile = nbn * nbm;
    const int nk = g.K / DTK;
    if ((int)blockIdx.x >= ntile) return;
    const int n_my = (ntile - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total = n_my * nk;
    const char* __restrict__ Ab = (const char*)g.A;
    const char* __restrict__ Wb = (const char*)g.W;

    // ---- DMA cursor (runs three k-steps ahead of the MFMA cursor, across tile boundaries) ----
    unsigned d_so[3];
    int d_t = 0, d_kt = 0;
    auto d_set_tile = [&](int t) {
        int mb, nb;
        tile_of_block_e((int)blockIdx.x + t * (int)gridDim.x, nbm, nbn, &mb, &nb);
        const int m0 = mb * DTM, n0 = nb * DTN;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const int ra = p * 128 + wave * 16 + (lane >> 2);                 // row of the stacked [X 256; W 128] stage
            const int c = (lane & 3) ^ (3 * ((ra >> 3) & 1));
            if (p < 2) {
                long m = m0 + ra;
                if (m > g.M - 1) m = g.M - 1;
                d_so[p] = (unsigned)(((m / g.a_rows_per_batch) * g.a_batch_stride + (m % g.a_rows_per_batch) * g.lda + c * 8) * (long)sizeof(T));
            } else {
                d_so[p] = (unsigned)(((long)(n0 + ra - DTM) * g.K + c * 8) * (long)sizeof(T));
            }
        }
    };
    d_set_tile(0);
    const int wave_off = wave * 16 * 64;
    // vmcnt bookkeeping (wave-uniform): the DMA of stage g+1 is issued during step g-2; after it this wave issues the epilogue store(s) of
    // step g-2, the 3 DMA + store(s) of step g-1 -- so at the start of step g exactly 3 + st1 + st2 newer operations may stay outstanding
    int st1 = 0, st2 = 0;   // epilogue stores issued in the previous step / the one before

#define DE_DMA_PASS(p, dbase)                                                                                             \
    glds16e<T>(((p) < 2 ? Ab : Wb) + d_so[p] + (size_t)d_kt * (DTK * sizeof(T)), smem + (dbase) + (p) * (128 * 64) + wave_off)
#define DE_DMA_DONE()                                                                                                     \
    { d_kt++; if (d_kt == nk) { d_kt = 0; d_t++; if (d_t < n_my) d_set_tile(d_t); } }

    // ---- MFMA cursor / parked tile ----
    int c_kt = 0, c_t = 0, m0 = 0, n0 = 0;
    {
        int mb, nb;
        tile_of_block_e((int)blockIdx.x, nbm, nbn, &mb, &nb);
        m0 = mb * DTM; n0 = nb * DTN;
    }
    f32x4 acc[4][4], accP[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accP[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    int pm0 = 0, pn0 = 0, pend = 16;     // parked tile origin; next epilogue slice to emit (16 = nothing parked)
    int pb0 = 0, pmap0 = 0, pmap1 = 0;   // EPI_CROSS_KV with a slot map: first window of the parked tile and the cache slots of it and the next one

    const int foff = frow * 64 + ((fg ^ (3 * ((frow >> 3) & 1))) * 16);
    const int xoff = (wm * 64) * 64 + foff, woff = DTM * 64 + (wn * 64) * 64 + foff;

    // one 16 (n) x 16 (m) sub-tile of the parked tile (bias already added): slice s = mi * 4 + ni; returns the stores issued (0 or 1)
    auto emit_slice = [&](int s) -> int {
        int issued = 0;
#define DE_SLICE(S)                                                                                                      \
    case S: {                                                                                                            \
        constexpr int mi = (S) >> 2, ni = (S) & 3;                                                                       \
        const int row_base = pm0 + wm * 64 + mi * 16;                                                                    \
        if (row_base < g.M) {                                                                                            \
            const long m = row_base + frow;                                                                              \
            const int n = pn0 + wn * 64 + ni * 16 + fg * 4;                                                              \
            const f32x4 v = accP[ni][mi];                                                                                \
            if (m < g.M) {                                                                                               \
                if constexpr (KIND == EPI_STORE_F32) {                                                                   \
                    const long orow = (m / g.o_rows_per_batch) * g.o_batch_stride + (m % g.o_rows_per_batch) * g.ldo;    \
                    *(f32x4*)((float*)g.out + orow + n) = v;                                                             \
                } else if constexpr (KIND == EPI_CROSS_KV) {                                                             \
                    const int H = g.d / 64;                                                                              \
                    const int l = n / (2 * g.d), rem = n % (2 * g.d), kv = rem / g.d, hj = rem % g.d, h = hj >> 6, j = hj & 63; \
                    int b = (int)(m / g.rows_per_batch);                                                                 \
                    const int t = (int)(m % g.rows_per_batch);                                                           \
                    if (g.use_batch_map) b = b == pb0 ? pmap0 : pmap1;   /* a 256-row tile touches at most two windows */ \
                    const float sc = kv == 0 ? g.scale : 1.0f;                                                           \
                    V4 o;                                                                                                \
                    o[0] = (T)(v[0] * sc); o[1] = (T)(v[1] * sc); o[2] = (T)(v[2] * sc); o[3] = (T)(v[3] * sc);          \
                    const long off = ((((long)(l * g.n_batch + b) * 2 + kv) * H + h) * g.rows_per_batch + t) * 64 + j;   \
                    *(V4*)((T*)g.out + off) = o;                                                                         \
                } else {                                                                                                 \
                    const long orow = (m / g.o_rows_per_batch) * g.o_batch_stride + (m % g.o_rows_per_batch) * g.ldo;    \
                    V4 o;                                                                                                \
                    if constexpr (KIND == EPI_GELU_T) {                                                                  \
                        o[0] = (T)gelu_e(gelu_round_e<T>(v[0], g.gelu_f16_in)); o[1] = (T)gelu_e(gelu_round_e<T>(v[1], g.gelu_f16_in)); \
                        o[2] = (T)gelu_e(gelu_round_e<T>(v[2], g.gelu_f16_in)); o[3] = (T)gelu_e(gelu_round_e<T>(v[3], g.gelu_f16_in)); \
                    } else {                                                                                             \
                        o[0] = (T)(v[0] * g.scale); o[1] = (T)(v[1] * g.scale); o[2] = (T)(v[2] * g.scale); o[3] = (T)(v[3] * g.scale); \
                    }                                                                                                    \
                    *(V4*)((T*)g.out + orow + n) = o;                                                                    \
                }                                                                                                        \
            }                                                                                                            \
            issued = 1;   /* one store instruction of this wave (some lanes may be masked: it still issues) */           \
        }                                                                                                                \
    } break;
        switch (s) {
            DE_SLICE(0) DE_SLICE(1) DE_SLICE(2) DE_SLICE(3) DE_SLICE(4) DE_SLICE(5) DE_SLICE(6) DE_SLICE(7)
            DE_SLICE(8) DE_SLICE(9) DE_SLICE(10) DE_SLICE(11) DE_SLICE(12) DE_SLICE(13) DE_SLICE(14) DE_SLICE(15)
            default: break;
        }
#undef DE_SLICE
        return issued;
    };
    // the finished tile becomes the parked one, bias added on the way: the 64 consecutive bias values of this wave's column half come through
    // the scalar cache and the lane's 16 are picked with selects -- no vector memory operation, nothing for vmcnt to track
    auto park = [&]() {
        pm0 = m0; pn0 = n0; pend = 0;
        const float* bp = g.bias ? g.bias + pn0 + wn * 64 : nullptr;      // wave-uniform address
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (bp) {
                // written as an explicit scalar load: the compiler will not use the scalar cache in a kernel that also stores (possible aliasing),
                // and a vector load here would sit in the same in-order vmcnt queue as the DMA stages
                typedef float f32x16 __attribute__((ext_vector_type(16)));
                f32x16 bs;
                asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(bs) : "s"(bp + ni * 16) : "memory");
#pragma unroll
                for (int r = 0; r < 4; r++) bv[r] = fg == 0 ? bs[r] : fg == 1 ? bs[4 + r] : fg == 2 ? bs[8 + r] : bs[12 + r];
            }
#pragma unroll
            for (int mi = 0; mi < 4; mi++) { accP[ni][mi] = acc[ni][mi] + bv; acc[ni][mi] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        }
        if constexpr (KIND == EPI_CROSS_KV) {
            if (g.use_batch_map) {      // uniform indices: scalar loads from the kernel arguments
                pb0 = pm0 / g.rows_per_batch;
                const int nb1 = (g.M + g.rows_per_batch - 1) / g.rows_per_batch - 1;
                pmap0 = g.batch_map[pb0 < nb1 ? pb0 : nb1];
                pmap1 = g.batch_map[pb0 + 1 < nb1 ? pb0 + 1 : nb1];
            }
        }
    };

    // ---- prologue: three stages in flight, fragments of step 0 in registers ----
#pragma unroll
    for (int s = 0; s < DNST - 1; s++) {
        if (s < total) {
            DE_DMA_PASS(0, s * kDStage); DE_DMA_PASS(1, s * kDStage); DE_DMA_PASS(2, s * kDStage);
            DE_DMA_DONE()
        }
    }
    V8 wfA[4], wfB[4], xf[4];     // W fragments double-buffered; X fragments refreshed in place, row block by row block
    if (total >= 3) wait_vm<6>(); else if (total == 2) wait_vm<3>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; i++) { wfA[i] = *(const V8*)(smem + woff + i * 16 * 64); xf[i] = *(const V8*)(smem + xoff + i * 16 * 64); }

    // quarter q of a step: the 4 MFMAs of token-row block mi = q (all 4 weight-row blocks), then X block q of the NEXT stage replaces the one just
    // used and W block q of the next stage goes to the other W set; one DMA pass in three of the four quarters
#define DE_QUARTER(WC, WN, q, rbase, dbase, do_dma, do_read)                                                               \
    {                                                                                                                    \
        _Pragma("unroll") for (int ni = 0; ni < 4; ni++) acc[ni][q] = MfmaE<T>::mma(WC[ni], xf[q], acc[ni][q]);           \
        if constexpr ((q) < 3) { if (do_dma) DE_DMA_PASS(q, dbase); }                                                    \
        if (do_read) {                                                                                                   \
            WN[q] = *(const V8*)(smem + (rbase) + woff + (q) * 16 * 64);                                                 \
            xf[q] = *(const V8*)(smem + (rbase) + xoff + (q) * 16 * 64);                                                 \
        }                                                                                                                \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                               \
        if constexpr ((q) < 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                                               \
    }
    // step gcur: MFMAs on (WC, xf) = stage gcur; stage gcur+1 -> (WN, xf); DMA of stage gcur+3 into the slot of stage gcur-1
#define DE_STEP(gcur, WC, WN)                                                                                              \
    if ((gcur) < total) {                                                                                                \
        const bool has_next = (gcur) + 1 < total;                                                                        \
        if (has_next) {                                                                                                  \
            const int later_dma = (gcur) + 2 < total ? 3 : 0;          /* DMA of stage gcur+2, issued after that of gcur+1 */ \
            switch (later_dma + st1 + st2) {                                                                             \
                case 0: wait_vm<0>(); break; case 1: wait_vm<1>(); break; case 2: wait_vm<2>(); break; case 3: wait_vm<3>(); break; \
                case 4: wait_vm<4>(); break; case 5: wait_vm<5>(); break; case 6: wait_vm<6>(); break; default: wait_vm<7>(); break; \
            }                                                                                                            \
            __builtin_amdgcn_s_barrier();                                                                                \
        }                                                                                                                \
        const bool dma = (gcur) + 3 < total;                                                                             \
        const int rbase = (((gcur) + 1) & 3) * kDStage, dbase = (((gcur) + 3) & 3) * kDStage;                            \
        DE_QUARTER(WC, WN, 0, rbase, dbase, dma, has_next)                                                               \
        DE_QUARTER(WC, WN, 1, rbase, dbase, dma, has_next)                                                               \
        DE_QUARTER(WC, WN, 2, rbase, dbase, dma, has_next)                                                               \
        DE_QUARTER(WC, WN, 3, rbase, dbase, dma, has_next)                                                               \
        if (dma) DE_DMA_DONE()                                                                                           \
        st2 = st1; st1 = 0;                                                                                              \
        if (pend < 16) { st1 += emit_slice(pend); pend++; }                                                              \
        c_kt++;                                                                                                          \
        if (c_kt == nk) {          /* this tile's sum is complete */                                                    \
            while (pend < 16) { st1 += emit_slice(pend); pend++; }   /* (only when nk < 16: the previous parked tile is not drained yet) */ \
            if (st1 > 2) { wait_vm<0>(); st1 = 0; st2 = 0; }   /* a burst of stores would overflow the small wait table: drain once */ \
            park();                                                                                                      \
            c_kt = 0; c_t++;                                                                                             \
            if (c_t < n_my) {                                                                                            \
                int mb, nb;                                                                                              \
                tile_of_block_e((int)blockIdx.x + c_t * (int)gridDim.x, nbm, nbn, &mb, &nb);                             \
                m0 = mb * DTM; n0 = nb * DTN;                                                                            \
            }                                                                                                            \
        }                                                                                                                \
    }

    for (int gb = 0; gb < total; gb += 2) {
        DE_STEP(gb, wfA, wfB)
        DE_STEP(gb + 1, wfB, wfA)
    }
    while (pend < 16) { emit_slice(pend); pend++; }      // the last tile
#undef DE_STEP
#undef DE_QUARTER
#undef DE_DMA_PASS
#undef DE_DMA_DONE
}

}  // namespace

// true if the deferred-epilogue kernel took the launch (shape and epilogue kind supported)
template <typename T>
bool launch_gemm_de(const GemmDesc& g, hipStream_t st) {
    if (g.N % DTN || g.K % DTK || g.K < 4 * DTK || g.M < 1024) return false;
    // EPI_CROSS_KV is instantiated but not routed here yet: its index arithmetic pushes the kernel over the register budget (8 spilled VGPRs)
    static const bool with_ckv = getenv("SS_GEMM_DE_CKV") != nullptr;
    if (!(g.kind == EPI_STORE_T || g.kind == EPI_GELU_T || (g.kind == EPI_CROSS_KV && with_ckv) || g.kind == EPI_STORE_F32)) return false;
    int n_cu = device_cu_count() / 8 * 8;
    if (n_cu < 8) n_cu = 8;
    const int ntile = (g.N / DTN) * ((g.M + DTM - 1) / DTM);
    const int grid = ntile < n_cu ? ntile : n_cu;
#define DE_LAUNCH(KIND)                                                                                                                    \
    {                                                                                                                                      \
        static std::atomic<uint64_t> attr{0};                                                                                              \
        once_per_device(attr, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm_de_kernel<T, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, kDLds)); }); \
        gemm_de_kernel<T, KIND><<<grid, 512, kDLds, st>>>(g); SS_LAUNCH_CHECK();                                                           \
    }
    switch (g.kind) {
        case EPI_STORE_T: DE_LAUNCH(EPI_STORE_T) break;
        case EPI_GELU_T: DE_LAUNCH(EPI_GELU_T) break;
        case EPI_CROSS_KV: DE_LAUNCH(EPI_CROSS_KV) break;
        default: DE_LAUNCH(EPI_STORE_F32) break;
    }
#undef DE_LAUNCH
    return true;
}
template bool launch_gemm_de<bf16>(const GemmDesc&, hipStream_t);
template bool launch_gemm_de<f16>(const GemmDesc&, hipStream_t);

}  // namespace ss
