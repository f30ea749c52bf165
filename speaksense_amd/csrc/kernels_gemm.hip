// MFMA GEMMs for gfx950 (CDNA4): the encoder / cross-KV projections (compute-bound, bf16|f16 MFMA roofline).  The decode-step
// projections (HBM-bound: weights streamed once per pass) live in kernels_decode.hip.
//
// Replaces ggml's mul_mat + add + gelu graph nodes that whisper.cpp builds for the encoder conv stem, the QKV/O and
// FFN projections, the cross-KV precompute and the decoder projections (op inventory in
// /root/reference/resources/ggml-metal.metal:3861-4003 kernel_mul_mm, :1307-1363 kernel_mul_mv_f16_f32).
// Not a translation of those: wave64 16x16x32 MFMA tiles, direct global->LDS DMA with a source-side XOR swizzle,
// operands swapped so each lane owns 4 consecutive output features (8/16-byte stores), epilogues fused.
#include <cstdlib>

#include "gemm_common.h"

namespace ss {

// ---------------------------------------------------------------------------------------------
// 128 x 128 x 64 tile, 256 threads = 4 waves (2 n x 2 m), each wave 64 x 64 = 4 x 4 MFMA 16x16x32 tiles.
// LDS: 2 stages x (X tile 16 KB + W tile 16 KB) = 64 KB  ->  2 workgroups / CU.
// LDS rows are 128 B (64 k); 16-B chunk c of row r is stored at chunk position c ^ ((r>>1)&7): the DMA writes
// lane-linear, so the permutation is applied to the per-lane SOURCE address and again on the ds_read_b128 side.
// ---------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int kTileBytes = BM * BK * 2;       // 16 KB
constexpr int kGemmLds = 4 * kTileBytes;      // 64 KB


template <typename T, int KIND>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmDesc g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mfma<T>::V8 V8;
    typedef typename Mfma<T>::V4 V4;
    constexpr bool SWAP = (KIND == EPI_VT);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int nbn = g.N / BN, nbm = (g.M + BM - 1) / BM;
    int mb, nb;
    tile_of_block(blockIdx.x, nbm, nbn, &mb, &nb);
    const int m0 = mb * BM, n0 = nb * BN;

    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ W = (const T*)g.W;

    // staging: pass p covers tile rows p*32 + tid/8; this lane's LDS chunk position is tid%8
    const int srow = tid >> 3, cpos = tid & 7;
    const T* xsrc[4];
    const T* wsrc[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int r = p * 32 + srow;
        const int c = cpos ^ ((r >> 1) & 7);
        long m = m0 + r;
        if (m > g.M - 1) m = g.M - 1;
        xsrc[p] = A + row_off(m, g.a_rows_per_batch, g.a_batch_stride, g.lda) + c * 8;
        wsrc[p] = W + (long)(n0 + r) * g.K + c * 8;
    }
    char* const sX0 = smem;
    char* const sW0 = smem + 2 * kTileBytes;
    const int stage_off = (wave * 8) * 128;  // this wave's first row within a 32-row pass

    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int p = 0; p < 4; p++) {
            glds16<T>(xsrc[p] + k0, sX0 + buf * kTileBytes + p * 32 * 128 + stage_off);
            glds16<T>(wsrc[p] + k0, sW0 + buf * kTileBytes + p * 32 * 128 + stage_off);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4, sw = (lane >> 1) & 7;
    const int nk = g.K / BK;
    stage(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
        if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
        const char* bX = sX0 + cur * kTileBytes + (wm * 64 + frow) * 128;
        const char* bW = sW0 + cur * kTileBytes + (wn * 64 + frow) * 128;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int coff = ((kk * 4 + fg) ^ sw) * 16;
            V8 xf[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                xf[i] = *(const V8*)(bX + i * 16 * 128 + coff);
                wf[i] = *(const V8*)(bW + i * 16 * 128 + coff);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
#pragma unroll
                for (int mi = 0; mi < 4; mi++) {
                    if (SWAP) acc[ni][mi] = Mfma<T>::mma(xf[mi], wf[ni], acc[ni][mi]);
                    else acc[ni][mi] = Mfma<T>::mma(wf[ni], xf[mi], acc[ni][mi]);
                }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---------------- epilogue ----------------
    if constexpr (!SWAP) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const long m = m0 + wm * 64 + mi * 16 + frow;
            if (m >= g.M) continue;
            const long orow = row_off(m, g.o_rows_per_batch, g.o_batch_stride, g.ldo);
#pragma unroll
            for (int ni = 0; ni < 4; ni++) {
                const int n = n0 + wn * 64 + ni * 16 + fg * 4;
                f32x4 v = acc[ni][mi];
                if (g.bias) {
                    const f32x4 b = *(const f32x4*)(g.bias + n);
                    v += b;
                }
                if constexpr (KIND == EPI_STORE_T) {
                    V4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * g.scale);
                    *(V4*)((T*)g.out + orow + n) = o;
                } else if constexpr (KIND == EPI_GELU_T) {
                    const V4 o = gelu4<T>(v, g.gelu_f16_in);
                    *(V4*)((T*)g.out + orow + n) = o;
                } else if constexpr (KIND == EPI_RES_F32) {
                    const f32x4 rsd = *(const f32x4*)(g.res + orow + n);
                    *(f32x4*)((float*)g.out + orow + n) = rsd + v;
                } else if constexpr (KIND == EPI_STORE_F32) {
                    *(f32x4*)((float*)g.out + orow + n) = v;
                } else if constexpr (KIND == EPI_GELU_POS_F32) {
                    const f32x4 pe = *(const f32x4*)(g.pos + (long)(m % g.rows_per_batch) * g.N + n);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = gelu_tanh_f(gelu_in_round<T>(v[r], g.gelu_f16_in)) + pe[r];
                    *(f32x4*)((float*)g.out + orow + n) = o;
                } else if constexpr (KIND == EPI_CROSS_KV) {
                    // n = l*2d + kv*d + h*64 + j ; cache [l][b][kv][h][t][64]
                    const int H = g.d / 64;
                    const int l = n / (2 * g.d), rem = n % (2 * g.d), kv = rem / g.d, hj = rem % g.d, h = hj >> 6, j = hj & 63;
                    int b = (int)(m / g.rows_per_batch);
                    const int t = (int)(m % g.rows_per_batch);
                    if (g.use_batch_map) b = g.batch_map[b];
                    const float sc = kv == 0 ? g.scale : 1.0f;
                    V4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * sc);
                    const long off = ((((long)(l * g.n_batch + b) * 2 + kv) * H + h) * g.cache_rows + t) * 64 + j;
                    *(V4*)((T*)g.out + off) = o;
                }
            }
        }
    } else {
        // D[m][n]: lane owns 4 consecutive m (time) for one n: V^T [b][h][64][Tpad]
        const int H = g.d / 64;
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            const int n = n0 + wn * 64 + ni * 16 + frow;
            const float b = g.bias ? g.bias[n] : 0.0f;
            const int h = n >> 6, j = n & 63;
#pragma unroll
            for (int mi = 0; mi < 4; mi++) {
                const long m = m0 + wm * 64 + mi * 16 + fg * 4;
                if (m >= g.M) continue;
                const int bb = (int)(m / g.rows_per_batch), t = (int)(m % g.rows_per_batch);
                V4 o;
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] = (T)(acc[ni][mi][r] + b);
                *(V4*)((T*)g.out + (((long)(bb * H + h) * 64 + j) * g.Tpad + t)) = o;
            }
        }
    }
}

// The epilogue of a 256 x 256 tile whose accumulators are laid out as in gemm256_kernel: wave (wm, wn) owns rows [m0 + 128 wm, +128) x columns
// [n0 + 64 wn, +64), acc[ni][mi] = the 16 x 16 fragment at (column group ni, row group mi); a lane holds 4 consecutive columns of row frow
// (SWAP: 4 consecutive rows of column frow).  A function of its own so that other main loops can share it (tools/experiments/r03_gemm_8phase/).
// 16-byte output stores of the f16 / bf16 epilogues, non-temporal where the output cannot be cache-resident for its consumer anyway (round 6;
// A/B/A/B in profiles/r06_f_gemm_nt_stores_ab.txt).  The cross-K/V cache (245.8 MB per window, read by the decoder milliseconds later) always; the other
// outputs when the launch writes more than kNtOutBytes -- FC1 of a 32-window pass writes 491 MB, more than L2 + MALL hold, and as ordinary
// write-back lines those stores push the operand tiles the other workgroups are about to re-read out of L2: FC1 + GELU at M = 48 000
// 1005 - 1021 -> 1063 - 1068 TF/s, QK projection 934 - 952 -> 987 - 995, plain FC1 958 - 964 -> 1005 - 1009; at M = 12 000 (outputs of 61 - 123 MB that the
// next kernel reads back from the MALL) nt LOSES 4 % on the QK projection, hence the threshold.  Same bytes, same bits.
constexpr long kNtOutBytes = 160L << 20;
template <typename V> __device__ __forceinline__ void st16_out(V* p, const V& v, bool nt) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}
#ifndef SS_RES_AHEAD
#define SS_RES_AHEAD 1     // operand row groups in flight in the f32-residual epilogue; 2 / 3 / 4 measured in round 6 (see there): no gain
#endif
template <typename T, int KIND, bool ST16>
__device__ __forceinline__ void epilogue256(const GemmDesc& g, f32x4 (&acc)[4][8], const f32x4 (&bias_v)[4], int m0, int n0, int wm, int wn, int frow, int fg) {
    typedef typename Mfma<T>::V8 V8;
    typedef typename Mfma<T>::V4 V4;
    constexpr bool SWAP = (KIND == EPI_VT);
    // s_memtime stamps (tools/gemm_bench.cpp, SS_TRACE): on FC1 the store-only epilogue is 25 % of a tile, +GELU 28 %, +f32 residual 48 %.
    // Neither a start-time stagger of the workgroups, nor a second workgroup per CU (128 x 256 tiles), nor an LDS-transposed epilogue that
    // writes whole 128-B lines (4x fewer requests) shortened it: a CU drains its 128 KB tile at ~10 B/clk whatever the request shape.
    if constexpr (KIND == EPI_RES_F32 || KIND == EPI_GELU_POS_F32) {
        // f32 output added to a second f32 operand (residual / positional embedding): the operand rows are loaded one row group AHEAD of
        // the stores, so each wait for loads leaves the previous group's stores in flight (the compiler counts them into its vmcnt)
        // every element is read and then written by the same thread exactly once, so treating operand and output as non-aliasing is safe even
        // when they are the same buffer (x += ...); without it the compiler drains all stores (vmcnt(0)) before each group of loads
        const float* __restrict__ resp = g.res;
        const float* __restrict__ posp = g.pos;
        float* __restrict__ outp = (float*)g.out;
        auto src_of = [&](int mi, int ni) -> const float* {
            long m = m0 + wm * 128 + mi * 16 + frow;
            if (m > g.M - 1) m = g.M - 1;
            const int n = n0 + wn * 64 + ni * 16 + fg * 4;
            if constexpr (KIND == EPI_RES_F32) return resp + row_off(m, g.o_rows_per_batch, g.o_batch_stride, g.ldo) + n;
            else return posp + (long)(m % g.rows_per_batch) * g.N + n;
        };
        // kResAhead row groups of operand loads in flight.  Round 6 asked whether one group ahead (64 B per lane, 32 KB per CU outstanding) is what
        // holds this epilogue at 12 - 14 B/clk per CU: 2 / 3 / 4 groups ahead (3: 248 VGPRs, 4: 13 spills) changed nothing (Ox4 571 -> 578 / 578 / 542 TF/s,
        // FC2x4 1010 -> 1004 / 1014 / 985, A/B/A in profiles/r06_c_res_ahead_and_vendor_kernel.txt).  What the rate IS (tools/diag/store_rate_bench.cpp,
        // profiles/r06_m_store_rate_bench.txt): persistent workgroups with equal tiles all reach their epilogue together, and 256 CUs storing at once share the
        // chip's ~6.8 TB/s of store bandwidth = 26.6 GB/s per CU, whatever the request shape -- 128 KB in 9 000 cycles at 1.62 GHz is 23.6 GB/s, this epilogue's
        // 512 KB in 39 000 is 21.8; 32 CUs storing alone reach 82 GB/s each.  De-phasing half of every XCD (tools/experiments/r06_gemm_stagger/) gave nothing:
        // it splits the operand-sharing patch of an XCD, and any start offset is paid back as a tail of the same length (tiles are quantised).
        constexpr int kResAhead = SS_RES_AHEAD;
        f32x4 buf[kResAhead][4];
#pragma unroll
        for (int p = 0; p < kResAhead; p++)
#pragma unroll
            for (int ni = 0; ni < 4; ni++) buf[p][ni] = *(const f32x4*)src_of(p, ni);
#pragma unroll
        for (int mi = 0; mi < 8; mi++) {
            f32x4 cur[4];
#pragma unroll
            for (int ni = 0; ni < 4; ni++) cur[ni] = buf[mi % kResAhead][ni];
            if (mi + kResAhead < 8) {
#pragma unroll
                for (int ni = 0; ni < 4; ni++) buf[mi % kResAhead][ni] = *(const f32x4*)src_of(mi + kResAhead, ni);
            }
            const long m = m0 + wm * 128 + mi * 16 + frow;
            if (m >= g.M) continue;
            const long orow = row_off(m, g.o_rows_per_batch, g.o_batch_stride, g.ldo);
#pragma unroll
            for (int ni = 0; ni < 4; ni++) {
                const int n = n0 + wn * 64 + ni * 16 + fg * 4;
                f32x4 v = acc[ni][mi] + bias_v[ni];
                if constexpr (KIND == EPI_GELU_POS_F32) {
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = gelu_tanh_f(gelu_in_round<T>(v[r], g.gelu_f16_in));
                }
                *(f32x4*)(outp + orow + n) = cur[ni] + v;
            }
        }
    } else if constexpr (ST16) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int mi = 0; mi < 8; mi++) {
            const long m = m0 + wm * 128 + mi * 16 + frow;
            const bool row_ok = m < g.M;        // the swaps below involve every lane: no early exit
            const long orow = row_ok ? row_off(m, g.o_rows_per_batch, g.o_batch_stride, g.ldo) : 0;
            u32x2 pk[4];
#pragma unroll
            for (int ni = 0; ni < 4; ni++) {
                const f32x4 v = acc[ni][mi] + bias_v[ni];
                V4 o;
                if constexpr (KIND == EPI_STORE_T) {
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * g.scale);
                } else if constexpr (KIND == EPI_GELU_T) {
                    o = gelu4<T>(v, g.gelu_f16_in);
                } else {   // EPI_CROSS_KV: the K half of every layer's [K; V] column block is pre-scaled
                    const int nq = n0 + wn * 64 + ni * 16 + fg * 4;
                    const float sc = (nq % (2 * g.d)) < g.d ? g.scale : 1.0f;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * sc);
                }
                pk[ni] = __builtin_bit_cast(u32x2, o);
            }
#pragma unroll
            for (int p = 0; p < 4; p += 2) {
                // rows (16-lane groups) 1 and 3 of the first register swap with rows 0 and 2 of the second: afterwards a lane with fg even holds
                // columns [(fg >> 1) * 8, +8) of fragment p, a lane with fg odd the same columns of fragment p + 1
                asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                             : "+v"(pk[p][0]), "+v"(pk[p + 1][0]), "+v"(pk[p][1]), "+v"(pk[p + 1][1]));
                if (!row_ok) continue;
                const int n = n0 + wn * 64 + (p + (fg & 1)) * 16 + (fg >> 1) * 8;
                const u32x4 o16 = {pk[p][0], pk[p][1], pk[p + 1][0], pk[p + 1][1]};
                if constexpr (KIND == EPI_CROSS_KV) {
                    const int H = g.d / 64;
                    const int l = n / (2 * g.d), rem = n % (2 * g.d), kv = rem / g.d, hj = rem % g.d, h = hj >> 6, j = hj & 63;
                    int b = (int)(m / g.rows_per_batch);
                    const int t = (int)(m % g.rows_per_batch);
                    if (g.use_batch_map) b = g.batch_map[b];
                    const long off = ((((long)(l * g.n_batch + b) * 2 + kv) * H + h) * g.cache_rows + t) * 64 + j;
                    __builtin_nontemporal_store(o16, (u32x4*)((T*)g.out + off));
                } else {
                    st16_out((u32x4*)((T*)g.out + orow + n), o16, g.nt_out != 0);
                }
            }
        }
    } else if constexpr (!SWAP) {
#pragma unroll
        for (int mi = 0; mi < 8; mi++) {
            const long m = m0 + wm * 128 + mi * 16 + frow;
            if (m >= g.M) continue;
            const long orow = row_off(m, g.o_rows_per_batch, g.o_batch_stride, g.ldo);
#pragma unroll
            for (int ni = 0; ni < 4; ni++) {
                const int n = n0 + wn * 64 + ni * 16 + fg * 4;
                f32x4 v = acc[ni][mi] + bias_v[ni];
                if constexpr (KIND == EPI_STORE_T) {
                    V4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * g.scale);
                    *(V4*)((T*)g.out + orow + n) = o;
                } else if constexpr (KIND == EPI_GELU_T) {
                    const V4 o = gelu4<T>(v, g.gelu_f16_in);
                    *(V4*)((T*)g.out + orow + n) = o;
                } else if constexpr (KIND == EPI_RES_F32) {
                    const f32x4 rsd = *(const f32x4*)(g.res + orow + n);
                    *(f32x4*)((float*)g.out + orow + n) = rsd + v;
                } else if constexpr (KIND == EPI_STORE_F32) {
                    *(f32x4*)((float*)g.out + orow + n) = v;
                } else if constexpr (KIND == EPI_GELU_POS_F32) {
                    const f32x4 pe = *(const f32x4*)(g.pos + (long)(m % g.rows_per_batch) * g.N + n);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = gelu_tanh_f(gelu_in_round<T>(v[r], g.gelu_f16_in)) + pe[r];
                    *(f32x4*)((float*)g.out + orow + n) = o;
                } else if constexpr (KIND == EPI_CROSS_KV) {
                    const int H = g.d / 64;
                    const int l = n / (2 * g.d), rem = n % (2 * g.d), kv = rem / g.d, hj = rem % g.d, h = hj >> 6, j = hj & 63;
                    int b = (int)(m / g.rows_per_batch);
                    const int t = (int)(m % g.rows_per_batch);
                    if (g.use_batch_map) b = g.batch_map[b];
                    const float sc = kv == 0 ? g.scale : 1.0f;
                    V4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * sc);
                    const long off = ((((long)(l * g.n_batch + b) * 2 + kv) * H + h) * g.cache_rows + t) * 64 + j;
                    *(V4*)((T*)g.out + off) = o;
                }
            }
        }
    } else {
        const int H = g.d / 64;
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            const int n = n0 + wn * 64 + ni * 16 + frow;
            const float b = bias_v[ni][0];
            const int h = n >> 6, j = n & 63;
#pragma unroll
            for (int mi = 0; mi < 8; mi++) {
                const long m = m0 + wm * 128 + mi * 16 + fg * 4;
                if (m >= g.M) continue;
                const int bb = (int)(m / g.rows_per_batch), t = (int)(m % g.rows_per_batch);
                V4 o;
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] = (T)(acc[ni][mi][r] + b);
                *(V4*)((T*)g.out + (((long)(bb * H + h) * 64 + j) * g.Tpad + t)) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 32 tile, 512 threads = 8 waves (4 n x 2 m), each wave 64 n x 128 m = 4 x 8 MFMA tiles (128 acc VGPRs).
// 4-stage LDS ring (4 x 32 KB), filled by global_load_lds 16 B DMA; three stages stay in flight across the single
// raw s_barrier per k-step (counted s_waitcnt vmcnt, never 0 in the steady state), which is what hides the ~2 us
// HBM/L2 latency that the 2-stage kernel above exposes at K = 1280 (20 short k-steps).
// LDS rows are 64 B (32 k); chunk c of row r sits at position c ^ (3 * ((r >> 3) & 1)), applied on the DMA source
// address and on the ds_read_b128 side: every 16-lane ds_read_b128 service group then touches 16 distinct 16-B slots.
// ---------------------------------------------------------------------------------------------
// (The WM template parameter is a remnant of a 128 x 256, two-workgroups-per-CU variant that measured 15-20 % slower: only WM = 2 is instantiated.)
#ifndef SS_RING
#define SS_RING 4   // LDS ring depth of the 256 x 256 kernel (-DSS_RING=3|5 for experiments): 3 stages measured +40 % main-loop time, 5 (all 160 KB) +27 %
#endif
constexpr int TN = 256, TK = 32;
template <int WM> struct G256 {
    static constexpr int TM = 128 * WM, NWV = 4 * WM, NST = WM == 2 ? SS_RING : 3;
    static constexpr int kStage = (TM + TN) * TK * 2;      // 32 KB / 24 KB
    static constexpr int kLds = NST * kStage;              // 128 KB (one workgroup per CU) / 72 KB (two)
    static constexpr int RPP = 16 * NWV;                   // tile rows one staging pass covers (16 per wave)
    static constexpr int NP = (TM + TN) / RPP;             // DMA instructions per thread per stage: 4 / 6
};


// DMA4 (WM = 2 only): waves 0-3 issue ALL the LDS-DMA (two 16-row slots each per pass) and are the only ones that wait on vmcnt; waves 4-7
// never wait on it inside the main loop, so THEIR output stores of the previous tile drain in the background for a whole tile.
// ST16: f16 / bf16 output kinds store 16 B per lane instead of 8.  A lane's accumulator quad is 4 consecutive output columns = 8 bytes, and the
// four lanes frow + 16 fg of a fragment row cover 32 contiguous bytes; one v_permlane16_swap per dword between the fragments ni and ni + 1
// regroups them so that a lane holds 8 consecutive columns of ONE fragment: half the store instructions for the same bytes (measured r03_g:
// QK projection +4..6 %, FC1 +-2 %: this epilogue is not store-issue-bound the way cdna_hip_programming.md T21's attention tail is).
template <typename T, int KIND, int WM, bool DMA4 = false>
__global__ __launch_bounds__(WM * 256, 2) void gemm256_kernel(GemmDesc g) {
    constexpr bool ST16 = WM == 2 && (KIND == EPI_STORE_T || KIND == EPI_GELU_T || KIND == EPI_CROSS_KV);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mfma<T>::V8 V8;
    typedef typename Mfma<T>::V4 V4;
    typedef G256<WM> G;
    constexpr int TM = G::TM, NST = G::NST, kStageBytes = G::kStage, NP = G::NP, RPP = G::RPP;
    constexpr bool SWAP = (KIND == EPI_VT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = WM == 2 ? wave >> 1 : wave, wm = WM == 2 ? wave & 1 : 0;
    const int nbn = g.N / TN, nbm = (g.M + TM - 1) / TM;
    // persistent: one workgroup per CU walks tiles vb = blockIdx.x + j*gridDim.x (gridDim.x is a multiple of 8, so the XCD
    // of the remap is preserved); the next tile's DMA prologue is issued right behind the previous tile's output stores
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ W = (const T*)g.W;
    // staging: NP DMA instructions per thread per stage; pass p covers RPP rows of [X tile; W tile] (16 per wave), 4 lanes per 64-B row
    constexpr int NSLOT = DMA4 ? 2 : 1;                 // 16-row slots of a pass this wave stages (slot = wave, and wave + 4 with DMA4)
    const bool dma_wave = !DMA4 || wave < 4;
    const int spos = lane & 3;
    unsigned soff[NP * NSLOT];   // byte offsets from the (uniform) A / W base: 32-bit VGPRs, the base stays in SGPRs (saddr + voffset DMA)
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int u = 0; u < NSLOT; u++) {
                const int ra = p * RPP + (wave + 4 * u) * 16 + (lane >> 2);   // row of the stacked [X; W] stage; a pass lies entirely in X or in W
                const int c = spos ^ (3 * ((ra >> 3) & 1));     // TM is a multiple of 16: same parity as the row within its tile
                if (p * RPP < TM) {
                    long m = m0 + ra;
                    if (m > g.M - 1) m = g.M - 1;
                    soff[p * NSLOT + u] = (unsigned)((row_off(m, g.a_rows_per_batch, g.a_batch_stride, g.lda) + c * 8) * (long)sizeof(T));
                } else {
                    soff[p * NSLOT + u] = (unsigned)(((long)(n0 + ra - TM) * g.K + c * 8) * (long)sizeof(T));
                }
            }
        }
    };
    const int wave_off = wave * 16 * 64;
#define SS_DMA1(p, u, k0, dst) glds16<T>((const T*)((const char*)(((p) * RPP < TM ? A : W) + (k0)) + soff[(p) * NSLOT + (u)]), (dst) + (u) * (4 * 16 * 64))
#define SS_DMA(p, k0, dst) { SS_DMA1(p, 0, k0, dst); if constexpr (DMA4) SS_DMA1(p, 1, k0, dst); }
    auto stage = [&](int buf, int k0) {
        if (!dma_wave) return;
        char* base = smem + buf * kStageBytes;
#pragma unroll
        for (int p = 0; p < NP; p++) SS_DMA(p, k0, base + p * (RPP * 64) + wave_off);
    };
    // Early prologue (one workgroup per CU form): the first NST-1 DMA stages of tile i+1 are issued BEFORE the output stores of tile i, so
    // their HBM/L2 latency (7 % of an FC1 tile when exposed) hides under the epilogue.  gfx9's vmcnt retires in order and counts stores:
    // the waits for those stages must therefore allow the `carry` stores issued after them to remain outstanding.
    constexpr bool EARLY = WM == 2;
    constexpr int kCarry = ST16 ? 12 : 24;   // a full interior tile issues 32 (ST16: 16) output stores per thread after the prologue; margin kept
    bool pro_issued = false;
    int issued = 0, carry = 0;
    for (int vb = blockIdx.x; vb < nbn * nbm; vb += gridDim.x) {
    int mb, nb;
    tile_of_block(vb, nbm, nbn, &mb, &nb);
    const int m0 = mb * TM, n0 = nb * TN;
    long long* tr = g.trace ? g.trace + ((long)blockIdx.x * 8 + (vb - blockIdx.x) / gridDim.x) * 4 : nullptr;
    if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memtime();
    const int nk = g.K / TK;
    if (!pro_issued) {
        __builtin_amdgcn_s_barrier();   // every wave is done with the previous tile's LDS stages
        set_tile(m0, n0);
        carry = 0;
        for (issued = 0; issued < NST - 1 && issued < nk; issued++) stage(issued, issued * TK);
    }

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4;
    const int foff = frow * 64 + ((fg ^ (3 * ((frow >> 3) & 1))) * 16);
    const int xoff = (wm * 128) * 64 + foff, woff = TM * 64 + (wn * 64) * 64 + foff;

    // fragment sets A / B (register double buffer): the ds_reads of stage kt+1 are issued among the MFMAs of stage kt.
    // An LDS-DMA instruction costs ~60-185 issue cycles (MI355X_MICROARCH.md): left to itself the compiler clusters the
    // 4 DMAs and 12 ds_reads of a step in front of its 32 MFMAs, serialising that cost with MFMA issue.  Each quarter of a
    // step is therefore written as {8 MFMA, 1 DMA, 3 ds_read} and pinned with sched_group_barrier so the matrix pipe keeps
    // executing while the wave issues memory instructions.
    V8 wfA[4], xfA[8], wfB[4], xfB[8];
    constexpr int OPS = NP * NSLOT;               // DMA instructions a staging wave issues per stage
    auto wait_stage = [&](int st, int issued) {   // MY DMA of stage `st` has landed; later stages (OPS each) stay in flight
        if (!dma_wave) return;                    // the barrier that follows publishes the staging waves' data
        const int later = issued - 1 - st;
        if (carry && st < NST - 1) {              // a prologue stage of an early-issued tile: the previous tile's stores came after it
            if (NST >= 5 && later >= 3) wait_vmcnt<(NST >= 5 ? 3 : 2) * OPS + kCarry>();
            else if (later >= 2) wait_vmcnt<2 * OPS + kCarry>();
            else if (later == 1) wait_vmcnt<OPS + kCarry>();
            else wait_vmcnt<kCarry>();
        } else if (NST >= 5 && later >= 3) wait_vmcnt<(NST >= 5 ? 3 : 2) * OPS>();
        else if (later >= 2) wait_vmcnt<2 * OPS>();
        else if (later == 1) wait_vmcnt<OPS>();
        else wait_vmcnt<0>();
    };
#define SS_MMA_Q(WF, XF, q)                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < 8; mi++) {                                              \
        if (SWAP) acc[q][mi] = Mfma<T>::mma(XF[mi], WF[q], acc[q][mi]);                              \
        else acc[q][mi] = Mfma<T>::mma(WF[q], XF[mi], acc[q][mi]);                                   \
    }
    // one k-step: MFMAs on (WC, XC); optionally DMA the stage `dma_buf` and read the fragments of stage `rd_buf` into (WN, XN)
#define SS_QUARTER(WC, XC, WN, XN, do_dma, dma_k0, do_read, q)                                        \
    {                                                                                                \
        SS_MMA_Q(WC, XC, q)                                                                          \
        if ((do_dma) && dma_wave) {                                                                  \
            SS_DMA(q, dma_k0, dbase + (q) * (RPP * 64) + wave_off);                                  \
            if constexpr ((q) + 4 < NP) SS_DMA(((q) + 4) % NP, dma_k0, dbase + ((q) + 4) * (RPP * 64) + wave_off); \
        }                                                                                            \
        if (do_read) {                                                                               \
            WN[q] = *(const V8*)(rbase + woff + (q) * 16 * 64);                                      \
            XN[2 * (q)] = *(const V8*)(rbase + xoff + (2 * (q)) * 16 * 64);                          \
            XN[2 * (q) + 1] = *(const V8*)(rbase + xoff + (2 * (q) + 1) * 16 * 64);                  \
        }                                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                           \
        __builtin_amdgcn_sched_group_barrier(0x020, (((q) + 4 < NP) ? 2 : 1) * NSLOT, 0);            \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                           \
    }
#define SS_STEP(WC, XC, WN, XN, do_dma, dma_buf, dma_k0, do_read, rd_buf)                              \
    {                                                                                                \
        char* dbase = smem + (dma_buf) * kStageBytes;                                                \
        const char* rbase = smem + (rd_buf) * kStageBytes;                                           \
        SS_QUARTER(WC, XC, WN, XN, do_dma, dma_k0, do_read, 0)                                        \
        SS_QUARTER(WC, XC, WN, XN, do_dma, dma_k0, do_read, 1)                                        \
        SS_QUARTER(WC, XC, WN, XN, do_dma, dma_k0, do_read, 2)                                        \
        SS_QUARTER(WC, XC, WN, XN, do_dma, dma_k0, do_read, 3)                                        \
    }

    // prologue (issued above or behind the previous tile's main loop): NST-1 stages in flight, fragments of stage 0 into registers
    wait_stage(0, issued);
    __builtin_amdgcn_s_barrier();
    {
        const char* base = smem;
#pragma unroll
        for (int i = 0; i < 4; i++) wfA[i] = *(const V8*)(base + woff + i * 16 * 64);
#pragma unroll
        for (int i = 0; i < 8; i++) xfA[i] = *(const V8*)(base + xoff + i * 16 * 64);
    }
    if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < nk; kt += 2) {
        // ---- even step: stage kt from set A; set B <- stage kt+1 ----
        {
            const bool has_next = kt + 1 < nk;
            if (has_next) {
                wait_stage(kt + 1, issued);
                __builtin_amdgcn_s_barrier();   // stage kt+1 visible to all; every wave already holds stage kt in registers
            }
            const bool dma = has_next && issued < nk;   // refills the buffer of stage kt-1
            const int db = issued % NST, dk = issued * TK;
            if (dma) issued++;
            SS_STEP(wfA, xfA, wfB, xfB, dma, db, dk, has_next, (kt + 1) % NST)
        }
        // ---- odd step: stage kt+1 from set B; set A <- stage kt+2 ----
        if (kt + 1 < nk) {
            const bool has_next = kt + 2 < nk;
            if (has_next) {
                wait_stage(kt + 2, issued);
                __builtin_amdgcn_s_barrier();
            }
            const bool dma = has_next && issued < nk;
            const int db = issued % NST, dk = issued * TK;
            if (dma) issued++;
            SS_STEP(wfB, xfB, wfA, xfA, dma, db, dk, has_next, (kt + 2) % NST)
        }
    }
#undef SS_STEP
#undef SS_QUARTER
#undef SS_DMA
#undef SS_DMA1
#undef SS_MMA_Q

    if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memtime();
    // bias for this tile's columns, loaded BEFORE the next tile's DMAs are queued: vmcnt retires in order, so a load issued behind them
    // would make the epilogue wait for their whole HBM round trip (the fragment registers are dead here, so this costs no pressure)
    f32x4 bias_v[4];
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
        bias_v[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
            if constexpr (SWAP) bias_v[ni][0] = g.bias[n0 + wn * 64 + ni * 16 + frow];
            else bias_v[ni] = *(const f32x4*)(g.bias + n0 + wn * 64 + ni * 16 + fg * 4);
        }
        // the values must exist HERE: otherwise the compiler sinks the loads next to their uses -- 32 loads, each followed by
        // s_waitcnt vmcnt(0), which (in-order vmcnt) made every output store wait for the previous one to complete
        asm volatile("" : "+v"(bias_v[ni][0]), "+v"(bias_v[ni][1]), "+v"(bias_v[ni][2]), "+v"(bias_v[ni][3]));
    }
    pro_issued = false;
    if constexpr (EARLY) {
        const int vbn = vb + gridDim.x;
        if (vbn < nbn * nbm) {
            int mbn, nbn2;
            tile_of_block(vbn, nbm, nbn, &mbn, &nbn2);
            __builtin_amdgcn_s_barrier();   // every wave has read its last fragments: the ring is free
            set_tile(mbn * TM, nbn2 * TN);
#pragma unroll
            for (int i = 0; i < NST - 1; i++) stage(i, i * TK);   // straight-line (nk >= NST - 1 is checked at launch): the compiler's vmcnt bookkeeping stays exact
            issued = NST - 1;
            carry = (m0 + TM <= g.M) ? kCarry : 0;   // the stores of a partial tile are predicated: count none of them
            pro_issued = true;
        }
    }
    // ---------------- epilogue ----------------
    epilogue256<T, KIND, ST16>(g, acc, bias_v, m0, n0, wm, wn, frow, fg);
    if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memtime();
    }  // tile loop
}

// ---------------------------------------------------------------------------------------------
// The 256 x 256 tile with 64-deep stages (128-byte LDS rows).  Why: an LDS-DMA instruction occupies the CU's address path for ~28 cycles when
// its 64 lanes fetch sixteen 64-byte rows (the TK = 32 layout above) and ~12 when they fetch eight 128-byte rows (tools/diag/dma_issue_bench.cpp,
// profiles/r04_af_dma_issue_bench.txt) -- and that path is shared by all waves of the CU.  A 256 x 256 x 32 step needs 32 such instructions:
// ~900 of the 1024 cycles its MFMAs take, which is why the TK = 32 main loop measures ~1500 cycles per step whatever the schedule.  With
// 128-byte rows a 64-deep stage takes 64 instructions x 12 = ~790 of 2048 cycles.
// Ring: 2 stages x 64 KB.  Stage s is consumed in two 32-deep halves (register sets A / B as above); the one barrier per stage sits between
// them: by then every wave has read both halves of stage s into registers, so its buffer is refilled with stage s + 2 during the second half,
// and stage s + 1 (issued one stage earlier) is waited for there.  Chunk c of row r sits at position c ^ ((r >> 1) & 7): with the row
// parity selecting the bank half, every ds_read_b128 service group touches 16 distinct 16-byte slots (same swizzle as gemm_kernel).
// ---------------------------------------------------------------------------------------------
constexpr int TK2 = 64;
template <typename T, int KIND>
__global__ __launch_bounds__(512, 2) void gemm256k64_kernel(GemmDesc g) {
    constexpr bool ST16 = KIND == EPI_STORE_T || KIND == EPI_GELU_T || KIND == EPI_CROSS_KV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mfma<T>::V8 V8;
    constexpr int TM = 256, kStageBytes = (TM + TN) * TK2 * 2, RPP = 64, NP = (TM + TN) / RPP, NPX = TM / RPP;   // 64 KB per stage; a pass = 8 rows per wave, 64 per workgroup
    constexpr bool SWAP = (KIND == EPI_VT);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    const int nbn = g.N / TN, nbm = (g.M + TM - 1) / TM;
    const T* __restrict__ A = (const T*)g.A;
    const T* __restrict__ W = (const T*)g.W;
    // Source offsets.  All eight waves stage (8 DMA instructions per thread and stage; the "half the waves stage" form of gemm256_kernel needs 16
    // offsets per thread, which spilled inside the main loop).  A pass covers 64 consecutive rows of the stacked [X; W] stage, so a lane's
    // swizzle term (row >> 1) & 7 does not depend on the pass.  X rows go through row_off (conv-stem batching) and the clamp to M - 1: one
    // 32-bit offset per pass.  W rows are plain and always in range: ONE per-lane offset; the pass part (p - 4) * 64 * K elements is wave-uniform.
    const int lrow = wave * 8 + (lane >> 3);
    const int cpos = (lane & 7) ^ ((lrow >> 1) & 7);
    unsigned sx[NPX], sw_lane = 0;
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int p = 0; p < NPX; p++) {
            long m = m0 + p * RPP + lrow;
            if (m > g.M - 1) m = g.M - 1;
            sx[p] = (unsigned)((row_off(m, g.a_rows_per_batch, g.a_batch_stride, g.lda) + cpos * 8) * (long)sizeof(T));
        }
        sw_lane = (unsigned)(((long)(n0 + lrow) * g.K + cpos * 8) * (long)sizeof(T));
    };
    const int wave_off = wave * 8 * 128;
    const unsigned wpass = (unsigned)g.K * (unsigned)sizeof(T);   // bytes per W row (wave-uniform)
    // pass p (a compile-time constant) of the stage that starts at element k0, into the stage buffer at `base`
#define SS_DMA(p, k0, base)                                                                                                              \
    {                                                                                                                                    \
        if constexpr ((p) < NPX) glds16<T>((const T*)((const char*)(A + (k0)) + sx[(p) < NPX ? (p) : 0]), (base) + (p) * (RPP * 128) + wave_off); \
        else glds16<T>((const T*)((const char*)(W + (k0)) + (size_t)((((p) - NPX) * RPP) * wpass) + sw_lane), (base) + (p) * (RPP * 128) + wave_off); \
    }
    auto stage = [&](int buf, int k0) {
        char* base = smem + buf * kStageBytes;
        SS_DMA(0, k0, base) SS_DMA(1, k0, base) SS_DMA(2, k0, base) SS_DMA(3, k0, base)
        SS_DMA(4, k0, base) SS_DMA(5, k0, base) SS_DMA(6, k0, base) SS_DMA(7, k0, base)
    };
    constexpr int OPS = NP;                    // DMA instructions a thread issues per stage
    constexpr int kCarry = ST16 ? 12 : 24;     // as gemm256_kernel: output stores issued behind the early prologue that a wait may leave outstanding
    bool pro_issued = false;
    int carry = 0;
    const int ns = g.K / TK2;
    for (int vb = blockIdx.x; vb < nbn * nbm; vb += gridDim.x) {
    int mb, nb;
    tile_of_block(vb, nbm, nbn, &mb, &nb);
    const int m0 = mb * TM, n0 = nb * TN;
    long long* tr = g.trace ? g.trace + ((long)blockIdx.x * 8 + (vb - blockIdx.x) / gridDim.x) * 4 : nullptr;
    if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memtime();
    if (!pro_issued) {
        __builtin_amdgcn_s_barrier();
        set_tile(m0, n0);
        carry = 0;
        stage(0, 0);
        stage(1, TK2);      // ns >= 2 is checked at launch
    }

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fg = lane >> 4, swz = (frow >> 1) & 7;
    const int fo0 = frow * 128 + ((fg ^ swz) * 16), fo1 = frow * 128 + (((4 + fg) ^ swz) * 16);   // this lane's chunk in the first / second 32-deep half
    const int xrow = (wm * 128) * 128, wrow = TM * 128 + (wn * 64) * 128;

    // Fragments: a quarter of a 32-deep half runs the 8 MFMAs of TWO row fragments (xf[2q], xf[2q + 1]) against all four column fragments.  The
    // column fragments are therefore live for the whole half (two register sets, A / B), while a row fragment is dead once its quarter has
    // issued and is reloaded with the next half's data, first used four quarters later: 32 + 32 fragment registers where a full double
    // buffer (gemm256_kernel) holds 96, which is what lets the DMA offsets live beside 128 accumulators without spilling in the loop.
    V8 wfA[4], wfB[4], xf[8];
#define SS_MMA_X(WF, q)                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                 \
        _Pragma("unroll") for (int ni = 0; ni < 4; ni++) {                                          \
            if (SWAP) acc[ni][2 * (q) + j] = Mfma<T>::mma(xf[2 * (q) + j], WF[ni], acc[ni][2 * (q) + j]); \
            else acc[ni][2 * (q) + j] = Mfma<T>::mma(WF[ni], xf[2 * (q) + j], acc[ni][2 * (q) + j]);      \
        }                                                                                           \
    }
    // a quarter: 8 MFMAs; optionally passes 2q and 2q + 1 of the DMA of stage `dma_k0` (measured: all eight passes in the first one or two
    // quarters instead is within +-2 %, profiles/r04_ah_gemm_k64_dma_placement.txt); optionally the next half's fragments: one column fragment
    // into the other set, this quarter's two row fragments
#define SS_QUARTER(WC, WN, do_dma, dma_k0, do_read, fo, q)                                            \
    {                                                                                                \
        SS_MMA_X(WC, q)                                                                              \
        if (do_dma) { SS_DMA(2 * (q), dma_k0, dbase) SS_DMA(2 * (q) + 1, dma_k0, dbase) }            \
        if (do_read) {                                                                               \
            WN[q] = *(const V8*)(rbase + wrow + (fo) + (q) * 16 * 128);                              \
            xf[2 * (q)] = *(const V8*)(rbase + xrow + (fo) + (2 * (q)) * 16 * 128);                  \
            xf[2 * (q) + 1] = *(const V8*)(rbase + xrow + (fo) + (2 * (q) + 1) * 16 * 128);          \
        }                                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);                                           \
    }
#define SS_HALF(WC, WN, do_dma, dma_buf, dma_k0, do_read, rd_buf, fo)                                  \
    {                                                                                                \
        char* dbase = smem + (dma_buf) * kStageBytes;                                                \
        const char* rbase = smem + (rd_buf) * kStageBytes;                                           \
        SS_QUARTER(WC, WN, do_dma, dma_k0, do_read, fo, 0)                                            \
        SS_QUARTER(WC, WN, do_dma, dma_k0, do_read, fo, 1)                                            \
        SS_QUARTER(WC, WN, do_dma, dma_k0, do_read, fo, 2)                                            \
        SS_QUARTER(WC, WN, do_dma, dma_k0, do_read, fo, 3)                                            \
    }
    // stage 0 has landed (stage 1 and, after an early prologue, the previous tile's last stores may stay in flight)
    if (carry) wait_vmcnt<OPS + kCarry>(); else wait_vmcnt<OPS>();
    __builtin_amdgcn_s_barrier();
    {
        const char* base = smem;
#pragma unroll
        for (int i = 0; i < 4; i++) wfA[i] = *(const V8*)(base + wrow + fo0 + i * 16 * 128);
#pragma unroll
        for (int i = 0; i < 8; i++) xf[i] = *(const V8*)(base + xrow + fo0 + i * 16 * 128);
    }
    if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memtime();
#ifdef SS_K64_WAITTRACE
    long long wait_dma = 0, wait_bar = 0;
#endif
    for (int s = 0; s < ns; s++) {
        const int buf = s & 1;
        // first half: k 0..31 of stage s from set A; set B and the row fragments <- its k 32..63
        SS_HALF(wfA, wfB, false, 0, 0, true, buf, fo1)
        const bool has_next = s + 1 < ns;
        if (has_next) {
            // stage s + 1 was issued a whole stage ago; nothing younger of this tile is in flight behind it
#ifdef SS_K64_WAITTRACE
            const long long w0 = __builtin_amdgcn_s_memtime();
#endif
            if (carry && s == 0) wait_vmcnt<kCarry>(); else wait_vmcnt<0>();
#ifdef SS_K64_WAITTRACE
            const long long w1 = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads of stage s have completed: its buffer may be overwritten once everybody is here
            __builtin_amdgcn_s_barrier();
#ifdef SS_K64_WAITTRACE
            wait_dma += w1 - w0; wait_bar += __builtin_amdgcn_s_memtime() - w1;
#endif
        }
        // second half: k 32..63 from set B; the buffer of stage s takes stage s + 2; set A and the row fragments <- k 0..31 of stage s + 1
        const bool dma = s + 2 < ns;
        SS_HALF(wfB, wfA, dma, buf, (s + 2) * TK2, has_next, buf ^ 1, fo0)
    }
#undef SS_HALF
#undef SS_QUARTER
#undef SS_MMA_X

    if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memtime();
    f32x4 bias_v[4];
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
        bias_v[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (g.bias) {
            if constexpr (SWAP) bias_v[ni][0] = g.bias[n0 + wn * 64 + ni * 16 + frow];
            else bias_v[ni] = *(const f32x4*)(g.bias + n0 + wn * 64 + ni * 16 + fg * 4);
        }
        asm volatile("" : "+v"(bias_v[ni][0]), "+v"(bias_v[ni][1]), "+v"(bias_v[ni][2]), "+v"(bias_v[ni][3]));
    }
    pro_issued = false;
    {
        const int vbn = vb + gridDim.x;
        if (vbn < nbn * nbm) {
            int mbn, nbn2;
            tile_of_block(vbn, nbm, nbn, &mbn, &nbn2);
            __builtin_amdgcn_s_barrier();   // every wave has read its last fragments: both buffers are free
            set_tile(mbn * TM, nbn2 * TN);
            stage(0, 0);
            stage(1, TK2);
            carry = (m0 + TM <= g.M) ? kCarry : 0;
            pro_issued = true;
        }
    }
    epilogue256<T, KIND, ST16>(g, acc, bias_v, m0, n0, wm, wn, frow, fg);
    if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memtime();
#ifdef SS_K64_WAITTRACE
    if (tr && tid == 0) { tr[0] = 0; tr[1] = wait_dma; tr[2] = wait_dma + wait_bar; tr[3] = tr[2]; }   // debug build: "prologue" = cycles in the vmcnt wait, "loop" = cycles in the barrier
#endif
    }  // tile loop
#undef SS_DMA
}

int g_gemm_k64 = -1;     // env SS_GEMM_K64, read at the first launch: 0 = the 32-deep-stage main loop (gemm256_kernel; A/B reference), default 1
int g_gemm_cu_cap = 0;   // dev hook (tools/gemm_bench.cpp SS_GEMM_CUS): persistent workgroups of the 256 x 256 kernel on at most this many CUs (0 = all)

template <typename T, int KIND>
static void launch_gemm_kind(const GemmDesc& g, hipStream_t st) {
    static std::atomic<uint64_t> attr128{0};
    once_per_device(attr128, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm_kernel<T, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, kGemmLds)); });
    // Which kernel runs must not depend on M: M = windows x positions, and a window's bits may not depend on how many others share its encoder pass
    // (the two kernels sum over K in different groupings).  Until round 5 the 256 x 256 form required M >= 1024 -- harmless while every window had
    // 1500 positions, wrong with whisper_full_params.audio_ctx: one 752-position window took the 128 x 128 kernel, two together the 256 x 256 one, and a
    // sampled (t > 0) decode told them apart (found by the 600 s soak, profiles/r05_ak_soak_600s_finding.txt).  Rows past M are clamped, any M works.
    if (g.N % TN == 0 && g.K % TK == 0 && g.K >= 4 * TK) {
        // 256 x 256 tiles, one persistent workgroup per CU (a multiple of 8 so the XCD of the remap is preserved).  Measured and archived under
        // tools/experiments/r02_variants/: two 128 x 256 workgroups per CU (-15..20 %), a start-time stagger of the workgroups (no change, r03_i: worse); static wave priorities (s_setprio 1 for waves 4-7: -5 %, for the staging waves 0-3: no change, r03_r); an 8-phase main loop (tools/experiments/r03_gemm_8phase/: correct, 860-890 against 1090 TF/s at 4096^3).
        int n_cu = device_cu_count() / 8 * 8;
        if (g_gemm_cu_cap > 0 && n_cu > g_gemm_cu_cap) n_cu = g_gemm_cu_cap / 8 * 8;
        if (n_cu < 8) n_cu = 8;
        // f16-output kinds: half the waves do all the staging (main loop -9 %, their partners' stores drain unobserved); the f32 residual
        // kinds move 4x the epilogue bytes and measured 20 % slower that way (tools/gemm_bench.cpp)
        static constexpr bool dma4 = KIND == EPI_STORE_T || KIND == EPI_GELU_T || KIND == EPI_CROSS_KV;
        const int nwg = (g.N / TN) * ((g.M + G256<2>::TM - 1) / G256<2>::TM);
        if (g_gemm_k64 < 0) { const char* e = getenv("SS_GEMM_K64"); g_gemm_k64 = e ? atoi(e) : 1; }
        if (g_gemm_k64 && g.K % TK2 == 0 && g.K >= 2 * TK2) {
            static std::atomic<uint64_t> attrk64{0};
            once_per_device(attrk64, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm256k64_kernel<T, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, G256<2>::kLds)); });
            gemm256k64_kernel<T, KIND><<<nwg < n_cu ? nwg : n_cu, 512, G256<2>::kLds, st>>>(g); SS_LAUNCH_CHECK();
            return;
        }
        static std::atomic<uint64_t> attr256{0};
        once_per_device(attr256, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm256_kernel<T, KIND, 2, dma4>, hipFuncAttributeMaxDynamicSharedMemorySize, G256<2>::kLds)); });
        gemm256_kernel<T, KIND, 2, dma4><<<nwg < n_cu ? nwg : n_cu, 512, G256<2>::kLds, st>>>(g); SS_LAUNCH_CHECK();
        return;
    }
    const int nwg = (g.N / BN) * ((g.M + BM - 1) / BM);
    gemm_kernel<T, KIND><<<nwg, 256, kGemmLds, st>>>(g); SS_LAUNCH_CHECK();
}

template <typename T>
void launch_gemm(const GemmDesc& g_in, hipStream_t st) {
    GemmDesc g = g_in;
    if (g.cache_rows <= 0) g.cache_rows = g.rows_per_batch;
    g.nt_out = (g.kind == EPI_STORE_T || g.kind == EPI_GELU_T) && (long)g.M * g.N * (long)sizeof(T) > kNtOutBytes;
    if (g.N % BN || g.K % BK || g.M <= 0) throw Error(-1, "gemm: N must be a multiple of 128 and K of 64");
    switch (g.kind) {
        case EPI_STORE_T: launch_gemm_kind<T, EPI_STORE_T>(g, st); break;
        case EPI_GELU_T: launch_gemm_kind<T, EPI_GELU_T>(g, st); break;
        case EPI_RES_F32: launch_gemm_kind<T, EPI_RES_F32>(g, st); break;
        case EPI_GELU_POS_F32: launch_gemm_kind<T, EPI_GELU_POS_F32>(g, st); break;
        case EPI_VT: launch_gemm_kind<T, EPI_VT>(g, st); break;
        case EPI_CROSS_KV: launch_gemm_kind<T, EPI_CROSS_KV>(g, st); break;
        case EPI_STORE_F32: launch_gemm_kind<T, EPI_STORE_F32>(g, st); break;
        default: throw Error(-1, "gemm: bad epilogue kind");
    }
}
template void launch_gemm<bf16>(const GemmDesc&, hipStream_t);
template void launch_gemm<f16>(const GemmDesc&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// self-test hook (ss_selftest_gemm): the tiled kernels against a one-thread-per-output reference on seeded operands.  The parity tests
// run toy models whose GEMMs are one tile per workgroup; this drives the multi-tile machinery (early prologue, store-aware vmcnt,
// DMA4, partial tiles) at the large-v3 shapes.
// ---------------------------------------------------------------------------------------------
namespace {
template <typename T>
__global__ void st_fill(T* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = (T)(((x & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
template <typename T>
__global__ void st_ref(const T* A, const T* W, const float* bias, const float* res, float* C, int M, int N, int K, int kind) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = 0; k < K; k++) acc += (float)A[(long)m * K + k] * (float)W[(long)n * K + k];
    acc += bias[n];
    if (kind == EPI_GELU_T) acc = gelu_tanh_f(acc);
    if (kind == EPI_RES_F32) acc += res[(long)m * N + n];
    C[(long)m * N + n] = acc;
}
template <typename T>
__global__ void st_diff(const void* out, const float* ref, size_t n, int f32out, float* maxes /* [2]: max |diff|, max |ref| */) {
    float d = 0.f, r = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float o = f32out ? ((const float*)out)[i] : (float)((const T*)out)[i];
        d = fmaxf(d, fabsf(o - ref[i])); r = fmaxf(r, fabsf(ref[i]));
    }
    atomicMax((int*)maxes, __float_as_int(d));       // non-negative floats order like their bit patterns
    atomicMax((int*)maxes + 1, __float_as_int(r));
}
}  // namespace

template <typename T>
void gemm_selftest(int M, int N, int K, int kind, float* max_err, float* max_ref, hipStream_t st, int reps, float* avg_ms) {
    if (kind != EPI_STORE_T && kind != EPI_GELU_T && kind != EPI_RES_F32 && kind != EPI_STORE_F32) throw Error(-1, "gemm selftest: kind not covered");
    T *A, *W; float *bias, *res, *ref, *mx; void* out;
    const bool f32out = kind == EPI_RES_F32 || kind == EPI_STORE_F32;
    SS_HIP(hipMalloc(&A, (size_t)M * K * sizeof(T))); SS_HIP(hipMalloc(&W, (size_t)N * K * sizeof(T))); SS_HIP(hipMalloc(&bias, (size_t)N * 4));
    SS_HIP(hipMalloc(&res, (size_t)M * N * 4)); SS_HIP(hipMalloc(&ref, (size_t)M * N * 4)); SS_HIP(hipMalloc(&out, (size_t)M * N * 4)); SS_HIP(hipMalloc(&mx, 8));
    st_fill<T><<<1024, 256, 0, st>>>(A, (size_t)M * K, 11, 1.0f);
    st_fill<T><<<1024, 256, 0, st>>>(W, (size_t)N * K, 12, 0.05f);
    st_fill<float><<<64, 256, 0, st>>>(bias, (size_t)N, 13, 0.5f);
    st_fill<float><<<1024, 256, 0, st>>>(res, (size_t)M * N, 14, 2.0f);
    SS_HIP(hipMemsetAsync(mx, 0, 8, st));
    st_ref<T><<<dim3((N + 255) / 256, M), 256, 0, st>>>(A, W, bias, res, ref, M, N, K, kind);
    GemmDesc g{};
    g.A = A; g.lda = K; g.a_rows_per_batch = 0; g.W = W; g.M = M; g.N = N; g.K = K; g.kind = kind; g.bias = bias; g.out = out; g.ldo = N;
    g.o_rows_per_batch = 0; g.scale = 1.0f; g.rows_per_batch = 1500;
    if (kind == EPI_RES_F32) { SS_HIP(hipMemcpyAsync(out, res, (size_t)M * N * 4, hipMemcpyDeviceToDevice, st)); g.res = (const float*)out; }   // in place, as the engine uses it
    launch_gemm<T>(g, st);
    st_diff<T><<<1024, 256, 0, st>>>(out, ref, (size_t)M * N, f32out, mx);
    float h[2];
    SS_HIP(hipMemcpyAsync(h, mx, 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    *max_err = h[0]; *max_ref = h[1];
    if (reps > 0 && avg_ms) {
        hipEvent_t e0, e1;
        SS_HIP(hipEventCreate(&e0)); SS_HIP(hipEventCreate(&e1));
        SS_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) launch_gemm<T>(g, st);
        SS_HIP(hipEventRecord(e1, st));
        SS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(bias); (void)hipFree(res); (void)hipFree(ref); (void)hipFree(out); (void)hipFree(mx);
}
template void gemm_selftest<bf16>(int, int, int, int, float*, float*, hipStream_t, int, float*);
template void gemm_selftest<f16>(int, int, int, int, float*, float*, hipStream_t, int, float*);

}  // namespace ss
