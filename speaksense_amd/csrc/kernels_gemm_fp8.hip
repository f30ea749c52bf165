// fp8 (OCP e4m3) encoder / cross-KV GEMMs for gfx950 on the MX-scaled matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4
// (BASELINE.json configs[4]: "fp8 weights on CDNA4 fp8 MFMA").  The reference has no fp8 path; its nearest analogue is the quantised
// ggml family (/root/reference/script/download-ggml-model.sh:28-51).  What this replaces is the same mul_mat + add + gelu graph as
// kernels_gemm.hip, with both operands in one byte per element:
//   * a k-step is 64 BYTES of k per row, so the LDS image, the LDS-DMA pattern (4 lanes per 64-B row, 16 rows per wave instruction), the
//     4-stage ring and the counted vmcnt waits are those of gemm256_kernel -- but every step advances k by 64 instead of 32;
//   * one 32x32x64 MFMA consumes a whole 64-k row segment per lane pair (lane l: row l % 32, bytes 32 (l / 32) .. +31), so a 64 n x 128 m
//     wave tile is 2 x 4 MFMAs per k-step and the fragment registers are the same 48 per set as the f16 kernel's;
//   * the activation operand carries one E8M0 exponent byte per (row, 64-k block) -- exactly one byte per row per k-step -- which the
//     hardware applies inside the MFMA; the bytes of a 256-row tile for one k-step are 256 contiguous bytes (f8_scale_index) that each wave
//     brings into its own LDS slot with one 4-byte-per-lane LDS-DMA riding in the same in-order vmcnt queue as the operand DMAs;
//   * weights carry one f32 scale per output channel, applied with the bias in the epilogue.
#include <cstdlib>

#include "gemm_common.h"
#include "wave_ops.h"

namespace ss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr int QTM = 256, QTN = 256, QTK = 64;        // QTK: bytes = e4m3 elements of k per stage
constexpr int QNST = 4;
constexpr int kQStage = (QTM + QTN) * QTK;           // 32 KB
constexpr int kQScaleOff = QNST * kQStage;           // exponent-byte slots behind the ring: [stage][wave][256]
constexpr int kQLds = kQScaleOff + QNST * 8 * 256;   // 136 KB

// LDS-DMA through a buffer descriptor: the row / chunk offset of a lane is one 32-bit VGPR, the k position is an SGPR offset, the base
// lives in four SGPRs (the flat global_load_lds form kept a 64-bit per-lane address per DMA instruction and pushed the kernel into spills)
typedef __amdgpu_buffer_rsrc_t Rsrc;
__device__ __forceinline__ Rsrc make_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(bytes > 0xffffffffL ? 0xffffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ void blds16(Rsrc r, unsigned voff, int soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void blds4(Rsrc r, unsigned voff, int soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff, soff, 0, 0);
}
struct Frag8 { i32x4 lo, hi; };
__device__ __forceinline__ i32x8 frag_join(const Frag8& f) { return __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// ---- shared by the two main-loop forms below: what happens to a wave's 64 n x 128 m accumulator block after the k loop -------------------------
// per-column weight scale and bias, folded into the accumulators (128 FMAs) before the next tile's DMAs are queued (in-order vmcnt)
// non-SWAP: register r of acc[ni][mi] is column n0 + wn*64 + ni*32 + 8 (r >> 2) + 4 kh + (r & 3), row m0 + wm*128 + mi*32 + l32
// SWAP:     register r is ROW m0 + wm*128 + mi*32 + 8 (r >> 2) + 4 kh + (r & 3), column n0 + wn*64 + ni*32 + l32
template <int KIND>
__device__ __forceinline__ void f8_fold_scale_bias(const GemmF8Desc& g, f32x16 (&acc)[2][4], int n0, int wn, int l32, int kh) {
    constexpr bool SWAP = (KIND == F8_VT);
    {
        f32x4 bias_v[2][4], ws_v[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                if constexpr (SWAP) {
                    const int n = n0 + wn * 64 + ni * 32 + l32;
                    const float bn = g.bias ? g.bias[n] : 0.f, sn = g.w_scale[n];
                    bias_v[ni][gq] = (f32x4){bn, bn, bn, bn};
                    ws_v[ni][gq] = (f32x4){sn, sn, sn, sn};
                } else {
                    const int n = n0 + wn * 64 + ni * 32 + 8 * gq + 4 * kh;
                    bias_v[ni][gq] = g.bias ? *(const f32x4*)(g.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    ws_v[ni][gq] = *(const f32x4*)(g.w_scale + n);
                }
            }
        // folded into the accumulators here (128 FMAs): the 64 registers are free again before the residual prefetch of the epilogue needs them
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[ni][mi][r] = acc[ni][mi][r] * ws_v[ni][r >> 2][r & 3] + bias_v[ni][r >> 2][r & 3];
    }
}

template <typename T, int KIND>
__device__ __forceinline__ void f8_epilogue(const GemmF8Desc& g, f32x16 (&acc)[2][4], int m0, int n0, int wm, int wn, int l32, int kh) {
    typedef typename Mfma<T>::V4 V4;
    constexpr bool SWAP = (KIND == F8_VT);
    if constexpr (KIND == F8_RES_F32) {
        const float* __restrict__ resp = g.res;
        float* __restrict__ outp = (float*)g.out;
        auto src_of = [&](int mi, int ni, int gq) -> const float* {
            long m = m0 + wm * 128 + mi * 32 + l32;
            if (m > g.M - 1) m = g.M - 1;
            return resp + m * g.ldo + n0 + wn * 64 + ni * 32 + 8 * gq + 4 * kh;
        };
        // the residual of half-row-group t+1 (32 rows x 32 columns of this wave) is loaded before half-row-group t is stored
        f32x4 nxt[4];
#pragma unroll
        for (int gq = 0; gq < 4; gq++) nxt[gq] = *(const f32x4*)src_of(0, 0, gq);
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int mi = t >> 1, ni = t & 1;
            f32x4 cur[4];
#pragma unroll
            for (int gq = 0; gq < 4; gq++) cur[gq] = nxt[gq];
            if (t + 1 < 8) {
#pragma unroll
                for (int gq = 0; gq < 4; gq++) nxt[gq] = *(const f32x4*)src_of((t + 1) >> 1, (t + 1) & 1, gq);
            }
            const long m = m0 + wm * 128 + mi * 32 + l32;
            if (m >= g.M) continue;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; r++) v[r] = acc[ni][mi][gq * 4 + r];
                *(f32x4*)(outp + m * g.ldo + n0 + wn * 64 + ni * 32 + 8 * gq + 4 * kh) = cur[gq] + v;
            }
        }
    } else if constexpr (KIND == F8_GELU_F8) {
        unsigned ebytes = 0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const long m = m0 + wm * 128 + mi * 32 + l32;
            float v[2][4][4];
            float amax = 0.f;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {   // packed-f32 GELU (gemm_common.h gelu_tanh_pk): two elements per VALU instruction
                        f32x2 t = {acc[ni][mi][gq * 4 + r], acc[ni][mi][gq * 4 + r + 1]};
                        if (std::is_same<T, f16>::value && g.gelu_f16_in) t = gelu_tanh_pk<true>(t); else t = gelu_tanh_pk<false>(t);
                        v[ni][gq][r] = t[0]; v[ni][gq][r + 1] = t[1];
                        amax = fmaxf(amax, fmaxf(fabsf(t[0]), fabsf(t[1])));
                    }
            amax = swap32_max(amax);   // the other half of this row's 64 columns is in lane ^ 32
            const int e = e8m0_for_amax(amax);
            const float inv = pow2_neg_of_e8m0(e);
            ebytes |= (unsigned)e << (8 * mi);
            if (m < g.M) {
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++)
                        *(unsigned*)((unsigned char*)g.out + m * g.ldo + n0 + wn * 64 + ni * 32 + 8 * gq + 4 * kh) =
                            pack_e4m3x4(v[ni][gq][0] * inv, v[ni][gq][1] * inv, v[ni][gq][2] * inv, v[ni][gq][3] * inv);
            }
        }
        // exponent bytes of rows (mi*32 + l32, mi = 0..3) of this wave's 128-row half: one dword per lane in the tiled layout
        if (kh == 0) *(unsigned*)(g.out_scale + (long)((n0 + wn * 64) >> 6) * g.ld_osc + m0 + wm * 128 + l32 * 4) = ebytes;
    } else if constexpr (KIND == F8_CROSS_KV8) {
        // this wave's 64 columns are one head of one (layer, K|V): quantised per (key row, head) like the GELU output, K pre-scaled by dh^-1/4
        const int H = g.d / 64;
        const int nw = n0 + wn * 64;
        const int l = nw / (2 * g.d), rem = nw % (2 * g.d), kv = rem / g.d, h = (rem % g.d) >> 6;
        const float ksc = kv == 0 ? g.scale : 1.0f;
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const long m = m0 + wm * 128 + mi * 32 + l32;
            float v[2][4][4];
            float amax = 0.f;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++)
#pragma unroll
                    for (int r = 0; r < 4; r++) { v[ni][gq][r] = acc[ni][mi][gq * 4 + r] * ksc; amax = fmaxf(amax, fabsf(v[ni][gq][r])); }
            amax = swap32_max(amax);
            const int e = e8m0_for_amax(amax);
            const float inv = pow2_neg_of_e8m0(e);
            if (m < g.M) {
                int b = (int)(m / g.rows_per_batch);
                const int t = (int)(m % g.rows_per_batch);
                if (g.use_batch_map) b = g.batch_map[b];
                const long row = (((long)(l * g.n_batch + b) * 2 + kv) * H + h) * g.cache_rows + t;
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
#pragma unroll
                    for (int gq = 0; gq < 4; gq++)
                        *(unsigned*)((unsigned char*)g.out + row * 64 + ni * 32 + 8 * gq + 4 * kh) =
                            pack_e4m3x4(v[ni][gq][0] * inv, v[ni][gq][1] * inv, v[ni][gq][2] * inv, v[ni][gq][3] * inv);
                if (kh == 0) g.out_scale[row] = (unsigned char)e;
            }
        }
    } else if constexpr (!SWAP) {
#pragma unroll
        for (int mi = 0; mi < 4; mi++) {
            const long m = m0 + wm * 128 + mi * 32 + l32;
            if (m >= g.M) continue;
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const int n = n0 + wn * 64 + ni * 32 + 8 * gq + 4 * kh;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; r++) v[r] = acc[ni][mi][gq * 4 + r];
                    if constexpr (KIND == F8_STORE_T) {
                        V4 o;
#pragma unroll
                        for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * g.scale);
                        *(V4*)((T*)g.out + m * g.ldo + n) = o;
                    } else if constexpr (KIND == F8_STORE_F32) {
                        *(f32x4*)((float*)g.out + m * g.ldo + n) = v;
                    } else if constexpr (KIND == F8_CROSS_KV) {
                        const int H = g.d / 64;
                        const int l = n / (2 * g.d), rem = n % (2 * g.d), kv = rem / g.d, hj = rem % g.d, h = hj >> 6, j = hj & 63;
                        int b = (int)(m / g.rows_per_batch);
                        const int t = (int)(m % g.rows_per_batch);
                        if (g.use_batch_map) b = g.batch_map[b];
                        const float sc = kv == 0 ? g.scale : 1.0f;
                        V4 o;
#pragma unroll
                        for (int r = 0; r < 4; r++) o[r] = (T)(v[r] * sc);
                        *(V4*)((T*)g.out + ((((long)(l * g.n_batch + b) * 2 + kv) * H + h) * g.cache_rows + t) * 64 + j) = o;
                    }
                }
        }
    } else {
        const int H = g.d / 64;
#pragma unroll
        for (int ni = 0; ni < 2; ni++) {
            const int n = n0 + wn * 64 + ni * 32 + l32;
            const int h = n >> 6, j = n & 63;
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int gq = 0; gq < 4; gq++) {
                    const long m = m0 + wm * 128 + mi * 32 + 8 * gq + 4 * kh;
                    if (m >= g.M) continue;
                    const int bb = (int)(m / g.rows_per_batch), t = (int)(m % g.rows_per_batch);
                    V4 o;
#pragma unroll
                    for (int r = 0; r < 4; r++) o[r] = (T)acc[ni][mi][gq * 4 + r];
                    *(V4*)((T*)g.out + (((long)(bb * H + h) * 64 + j) * g.Tpad + t)) = o;
                }
        }
    }
}

// 256 (m) x 256 (n) x 64 tile, 512 threads = 8 waves (4 n x 2 m), each wave 64 n x 128 m = 2 x 4 MFMA 32x32x64; one workgroup per CU, persistent.
// LDS rows are 64 B; the 16-B chunk c of row r sits at chunk position c ^ ((r >> 2) & 3), which makes the ds_read_b128 of a fragment
// (32 rows x one chunk per half-wave) conflict-free; the permutation is applied to the per-lane DMA source address.
// KO (dev tool, tools/gemm_fp8_bench.py via SS_F8_KO; results are then WRONG): knock-outs that time what the k loop is made of --
// bit 0 no barriers, bit 1 no vmcnt waits, bit 2 no DMA, bit 3 no LDS fragment reads, bit 4 no MFMAs (all inside the k loop only)
template <typename T, int KIND, int KO = 0>
__global__ __launch_bounds__(512, 2) void gemm_f8_kernel(GemmF8Desc g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef typename Mfma<T>::V4 V4;
    constexpr bool SWAP = (KIND == F8_VT);
    constexpr int NP = 4, RPP = 128, OPS = NP + 1, NST = QNST;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in SGPRs
    const int wn = wave >> 1, wm = wave & 1;
    const int nbn = g.N / QTN, nbm = (g.M + QTM - 1) / QTM;
    const Rsrc rA = make_rsrc(g.A, (long)g.M * g.lda), rW = make_rsrc(g.W, (long)g.N * g.K), rS = make_rsrc(g.a_scale, (long)g.ldsc * (g.K / QTK));
    unsigned soff[NP], sc_off = 0;
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            const int ra = p * RPP + wave * 16 + (lane >> 2);
            const int c = (lane & 3) ^ ((lane >> 4) & 3);
            if (p * RPP < QTM) {
                long m = m0 + ra;
                if (m > g.M - 1) m = g.M - 1;
                soff[p] = (unsigned)(m * g.lda + c * 16);
            } else {
                soff[p] = (unsigned)((long)(n0 + ra - QTM) * g.K + c * 16);
            }
        }
        sc_off = (unsigned)(m0 + lane * 4);
    };
    const int wave_off = wave * 16 * 64;
#define SS_DMA(p, kt, dst) blds16((p) * RPP < QTM ? rA : rW, soff[p], (kt) * QTK, (dst))
#define SS_DMA_SC(buf, kt) blds4(rS, sc_off, (kt) * (int)g.ldsc, smem + kQScaleOff + ((buf) * 8 + wave) * 256)
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * kQStage;
#pragma unroll
        for (int p = 0; p < NP; p++) SS_DMA(p, kt, base + p * (RPP * 64) + wave_off);
        SS_DMA_SC(buf, kt);
    };
    constexpr int kCarry = 24;
    bool pro_issued = false;
    int issued = 0, carry = 0;
    const int l32 = lane & 31, kh = lane >> 5;
    const int sw = (l32 >> 2) & 3;
    const int foff0 = l32 * 64 + (((2 * kh) ^ sw) * 16), foff1 = l32 * 64 + (((2 * kh + 1) ^ sw) * 16);
    // four per-lane LDS read bases (X / W region x first / second 16-byte chunk); everything else of a fragment address is an immediate
    const char* const xb0 = smem + (wm * 128) * 64 + foff0;
    const char* const xb1 = smem + (wm * 128) * 64 + foff1;
    const char* const wb0 = smem + QTM * 64 + (wn * 64) * 64 + foff0;
    const char* const wb1 = smem + QTM * 64 + (wn * 64) * 64 + foff1;
    const char* const scb = smem + kQScaleOff + wave * 256 + wm * 128 + l32 * 4;
    const int nk = g.K / QTK;

    for (int vb = blockIdx.x; vb < nbn * nbm; vb += gridDim.x) {
    int mb, nb;
    tile_of_block(vb, nbm, nbn, &mb, &nb);
    const int m0 = mb * QTM, n0 = nb * QTN;
    if (!pro_issued) {
        __builtin_amdgcn_s_barrier();
        set_tile(m0, n0);
        carry = 0;
        for (issued = 0; issued < NST - 1 && issued < nk; issued++) stage(issued, issued);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // X fragments are single-buffered and refreshed in place: xf[q] is dead after quarter q's two MFMAs, and its successor (stage kt+1) is
    // not needed before quarter q of the next step -- a full step of latency for 32 fewer registers than a second set.
    Frag8 wfA[2], wfB[2], xf[4];
    int scA = 0, scB = 0;
    // The k loop runs in groups of NST = 4 steps so that every ring index is a compile-time constant and each step is ONE basic block
    // (with run-time "is there a next stage / a DMA" flags the compiler split a step into a dozen blocks, sank all MFMAs below the loads and
    // spilled LDS addresses with s_waitcnt vmcnt(0) reloads inside the loop).  K % 256 == 0 is checked at launch: every group but the last
    // is {DMA, read} x 4, the last one is {DMA, read}, {read}, {read}, {}.
    // Waits (in-order vmcnt, OPS DMA instructions per stage): at the start of step kt the stages kt+1 and kt+2 are in flight and kt+1 must have
    // landed -> vmcnt(OPS); the last group's third step waits for everything.  The first two waits of an early-issued tile additionally leave
    // the previous tile's output stores (>= kCarry of them, issued after the prologue stages) in flight.
#define SS_MMA1(WF, SCV, ni, q)                                                                                                       \
    {                                                                                                                                 \
        const int sx = ((SCV) >> (8 * (q))) & 0xff;                                                                                   \
        if constexpr (KO & 16) { acc[ni][q][0] += __builtin_bit_cast(float, frag_join(WF[ni])[0] ^ frag_join(xf[q])[(ni)] ^ sx); }          \
        else if (SWAP) acc[ni][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag_join(xf[q]), frag_join(WF[ni]), acc[ni][q], 0, 0, 0, sx, 0, 127); \
        else acc[ni][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag_join(WF[ni]), frag_join(xf[q]), acc[ni][q], 0, 0, 0, 127, 0, sx);      \
    }
    // one quarter of a step: 2 MFMAs on xf[q]; one operand DMA (+ the exponent-byte DMA in the last quarter); then the in-place refresh of
    // xf[q] and half a W fragment of the next stage
#define SS_QUARTER(WC, SCC, WN, SCN, DMA, READ, BUF, dma_kt, q)                                                                         \
    {                                                                                                                                 \
        SS_MMA1(WC, SCC, 0, q)                                                                                                        \
        SS_MMA1(WC, SCC, 1, q)                                                                                                        \
        if constexpr (DMA && !(KO & 4)) {                                                                                             \
            SS_DMA(q, dma_kt, smem + (((BUF) + NST - 1) % NST) * kQStage + (q) * (RPP * 64) + wave_off);                              \
            if constexpr ((q) == 3) SS_DMA_SC(((BUF) + NST - 1) % NST, dma_kt);                                                       \
        }                                                                                                                             \
        if constexpr (READ && !(KO & 8)) {                                                                                            \
            constexpr int ro = (((BUF) + 1) % NST) * kQStage;                                                                         \
            xf[q].lo = *(const i32x4*)(xb0 + ro + (q) * 32 * 64);                                                                     \
            xf[q].hi = *(const i32x4*)(xb1 + ro + (q) * 32 * 64);                                                                     \
            if constexpr (((q) & 1) == 0) WN[(q) >> 1].lo = *(const i32x4*)(wb0 + ro + ((q) >> 1) * 32 * 64);                         \
            else WN[(q) >> 1].hi = *(const i32x4*)(wb1 + ro + ((q) >> 1) * 32 * 64);                                                  \
            if constexpr ((q) == 0) SCN = *(const int*)(scb + (((BUF) + 1) % NST) * (8 * 256));                                       \
        }                                                                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                                            \
        if constexpr (DMA && !(KO & 4)) __builtin_amdgcn_sched_group_barrier(0x020, (q) == 3 ? 2 : 1, 0);                             \
        if constexpr (READ && !(KO & 8)) __builtin_amdgcn_sched_group_barrier(0x100, (q) == 0 ? 4 : 3, 0);                            \
    }
#define SS_STEP(WC, SCC, WN, SCN, DMA, READ, BUF, dma_kt)                                                                               \
    {                                                                                                                                 \
        SS_QUARTER(WC, SCC, WN, SCN, DMA, READ, BUF, dma_kt, 0)                                                                        \
        SS_QUARTER(WC, SCC, WN, SCN, DMA, READ, BUF, dma_kt, 1)                                                                        \
        SS_QUARTER(WC, SCC, WN, SCN, DMA, READ, BUF, dma_kt, 2)                                                                        \
        SS_QUARTER(WC, SCC, WN, SCN, DMA, READ, BUF, dma_kt, 3)                                                                        \
    }

    // prologue: stage 0 landed (stages 1, 2 and possibly the previous tile's stores stay in flight), its fragments into registers
    if (carry) wait_vmcnt<2 * OPS + kCarry>();
    else wait_vmcnt<2 * OPS>();
    __builtin_amdgcn_s_barrier();
    {
#pragma unroll
        for (int i = 0; i < 2; i++) { wfA[i].lo = *(const i32x4*)(wb0 + i * 32 * 64); wfA[i].hi = *(const i32x4*)(wb1 + i * 32 * 64); }
#pragma unroll
        for (int i = 0; i < 4; i++) { xf[i].lo = *(const i32x4*)(xb0 + i * 32 * 64); xf[i].hi = *(const i32x4*)(xb1 + i * 32 * 64); }
        scA = *(const int*)scb;
    }
    // A step must stay ONE basic block: a run-time branch between steps lets LLVM sink the MFMAs (whose results are only read after the
    // loop) below it, which keeps the old fragments alive and spills the new ones.  Hence compile-time wait counts per group kind:
    // CARRY groups (first group of an early-issued tile) leave the previous tile's stores in flight for their first two waits.
#define SS_SYNC(N) { if constexpr (!(KO & 2)) wait_vmcnt<(N)>(); if constexpr (!(KO & 1)) __builtin_amdgcn_s_barrier(); }
#define SS_GROUP(CARRY)                                                                                                                \
    {                                                                                                                                 \
        SS_SYNC(OPS + ((CARRY) ? kCarry : 0))                                                                                         \
        SS_STEP(wfA, scA, wfB, scB, true, true, 0, kt + 3)                                                                            \
        SS_SYNC(OPS + ((CARRY) ? kCarry : 0))                                                                                         \
        SS_STEP(wfB, scB, wfA, scA, true, true, 1, kt + 4)                                                                            \
        SS_SYNC(OPS)                                                                                                                  \
        SS_STEP(wfA, scA, wfB, scB, true, true, 2, kt + 5)                                                                            \
        SS_SYNC(OPS)                                                                                                                  \
        SS_STEP(wfB, scB, wfA, scA, true, true, 3, kt + 6)                                                                            \
    }
    int kt = 0;
    if (carry && nk > 4) { SS_GROUP(true) kt = 4; }
    for (; kt + 4 < nk; kt += 4) SS_GROUP(false)   // steady groups: stage kt+j+3 goes into the buffer step kt+j-1 just finished with
    {   // last group (kt == nk - 4): one more DMA (stage nk-1), then drain.  Its waits never leave stores in flight (only matters when nk == 4)
        SS_SYNC(OPS)
        SS_STEP(wfA, scA, wfB, scB, true, true, 0, kt + 3)
        SS_SYNC(OPS)
        SS_STEP(wfB, scB, wfA, scA, false, true, 1, 0)
        SS_SYNC(0)
        SS_STEP(wfA, scA, wfB, scB, false, true, 2, 0)
        SS_STEP(wfB, scB, wfA, scA, false, false, 3, 0)
    }
#undef SS_GROUP
#undef SS_SYNC
    // the accumulators must exist HERE: otherwise the last group's MFMAs are sunk below the next tile's prologue and the branches of the
    // epilogue (their results are first read there), with every fragment they need spilled on the way
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" : "+v"(acc[i][j]));
#undef SS_STEP
#undef SS_QUARTER
#undef SS_MMA1

    f8_fold_scale_bias<KIND>(g, acc, n0, wn, l32, kh);
    pro_issued = false;
    {
        const int vbn = vb + gridDim.x;
        if (vbn < nbn * nbm) {
            int mbn, nbn2;
            tile_of_block(vbn, nbm, nbn, &mbn, &nbn2);
            __builtin_amdgcn_s_barrier();
            set_tile(mbn * QTM, nbn2 * QTN);
#pragma unroll
            for (int i = 0; i < NST - 1; i++) stage(i, i);
            issued = NST - 1;
            carry = (m0 + QTM <= g.M) ? kCarry : 0;
            pro_issued = true;
        }
    }
    f8_epilogue<T, KIND>(g, acc, m0, n0, wm, wn, l32, kh);
    }  // tile loop
#undef SS_DMA
#undef SS_DMA_SC
}

// ---------------------------------------------------------------------------------------------
// The same tile with 128-byte LDS rows (round 5): a stage is 128 e4m3 of k per row = TWO 64-k MFMA blocks, 2 x 64 KB ring.
// Why: an LDS-DMA instruction whose 64 lanes fetch sixteen 64-byte rows holds the CU's address path for ~28 cycles, one that fetches eight
// 128-byte rows for ~12 (tools/diag/dma_issue_bench.cpp, profiles/r04_af_dma_issue_bench.txt), and that path is shared by the 8 waves.  A 64-byte-row
// stage above issues 40 of them (32 operand + 8 exponent-byte) = ~1100 cycles against the 1024 cycles its 16 e4m3 MFMAs per SIMD take: the main loop
// was bound by the address path, not by the matrix pipe.  With 128-byte rows a stage issues 64 + 16 instructions for 2048 cycles of MFMA (~800 +).
// Schedule as gemm256k64_kernel (kernels_gemm.hip): stage s is consumed in two halves (its two 64-k blocks; register sets A / B for the column
// fragments, the row fragments refreshed in place a quarter at a time), ONE barrier per stage between them -- by then every wave holds both
// halves of stage s in registers, so its buffer takes stage s + 2 during the second half while stage s + 1 (issued a whole stage earlier) has
// landed.  Chunk c (16 B) of row r sits at position c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 service group (16 consecutive rows, one
// logical chunk) touch 16 distinct slots of the 256-byte bank window.  Stages run in pairs so that every ring index is a compile-time constant
// and each half is one basic block (see the notes in gemm_f8_kernel on what run-time flags do to this loop); K % 256 == 0 makes the pairs whole.
// ---------------------------------------------------------------------------------------------
constexpr int Q2K = 128;
constexpr int kQ2Stage = (QTM + QTN) * Q2K;            // 64 KB
constexpr int kQ2ScaleOff = 2 * kQ2Stage;              // exponent bytes behind the ring: [stage][wave][block][256]
constexpr int kQ2Lds = kQ2ScaleOff + 2 * 8 * 2 * 256;  // 136 KB

template <typename T, int KIND>
__global__ __launch_bounds__(512, 2) void gemm_f8k128_kernel(GemmF8Desc g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool SWAP = (KIND == F8_VT);
    constexpr int NPX = 4, RPP = 64, OPS = 8 + 2;     // DMA instructions a thread issues per stage: 4 X passes + 4 W passes of 64 rows, 2 exponent-byte blocks
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wm = wave & 1;
    const int nbn = g.N / QTN, nbm = (g.M + QTM - 1) / QTM;
    const Rsrc rA = make_rsrc(g.A, (long)g.M * g.lda), rW = make_rsrc(g.W, (long)g.N * g.K), rS = make_rsrc(g.a_scale, (long)g.ldsc * (g.K / QTK));
    const int lrow = wave * 8 + (lane >> 3);              // row of a 64-row pass this lane stages (8 lanes x 16 B per 128-byte row)
    const int cpos = (lane & 7) ^ ((lrow >> 1) & 7);      // the logical chunk that belongs at this lane's LDS position
    unsigned sx[NPX], sw_lane = 0, sc_off = 0;
    auto set_tile = [&](int m0, int n0) {
#pragma unroll
        for (int p = 0; p < NPX; p++) {
            long m = m0 + p * RPP + lrow;
            if (m > g.M - 1) m = g.M - 1;
            sx[p] = (unsigned)(m * g.lda + cpos * 16);
        }
        sw_lane = (unsigned)((long)(n0 + lrow) * g.K + cpos * 16);
        sc_off = (unsigned)(m0 + lane * 4);
    };
    const int wave_off = wave * 8 * 128;
    const int wpass = RPP * g.K;                          // bytes between two W passes (wave-uniform: rides in the DMA's scalar offset)
#define SS_DMA2(p, kt, base)                                                                                                      \
    {                                                                                                                             \
        if constexpr ((p) < NPX) blds16(rA, sx[(p) < NPX ? (p) : 0], (kt) * Q2K, (base) + (p) * (RPP * 128) + wave_off);           \
        else blds16(rW, sw_lane, (kt) * Q2K + ((p) - NPX) * wpass, (base) + (p) * (RPP * 128) + wave_off);                         \
    }
#define SS_DMA2_SC(buf, kt, blk) blds4(rS, sc_off, ((kt) * 2 + (blk)) * (int)g.ldsc, smem + kQ2ScaleOff + (((buf) * 8 + wave) * 2 + (blk)) * 256)
    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * kQ2Stage;
        SS_DMA2(0, kt, base) SS_DMA2(1, kt, base) SS_DMA2(2, kt, base) SS_DMA2(3, kt, base)
        SS_DMA2(4, kt, base) SS_DMA2(5, kt, base) SS_DMA2(6, kt, base) SS_DMA2(7, kt, base)
        SS_DMA2_SC(buf, kt, 0); SS_DMA2_SC(buf, kt, 1);
    };
    constexpr int kCarry = 24;
    bool pro_issued = false;
    int carry = 0;
    const int l32 = lane & 31, kh = lane >> 5;
    const int swz = (l32 >> 1) & 7;
    // per-lane LDS read bases: {X, W} x {block 0, block 1} x {first, second 16-byte chunk of the lane's 32 bytes}; the rest of an address is an immediate
    const char* const xb00 = smem + (wm * 128) * 128 + l32 * 128 + (((2 * kh) ^ swz) * 16);
    const char* const xb01 = smem + (wm * 128) * 128 + l32 * 128 + (((2 * kh + 1) ^ swz) * 16);
    const char* const xb10 = smem + (wm * 128) * 128 + l32 * 128 + (((4 + 2 * kh) ^ swz) * 16);
    const char* const xb11 = smem + (wm * 128) * 128 + l32 * 128 + (((5 + 2 * kh) ^ swz) * 16);
    const char* const wb00 = smem + QTM * 128 + (wn * 64) * 128 + l32 * 128 + (((2 * kh) ^ swz) * 16);
    const char* const wb01 = smem + QTM * 128 + (wn * 64) * 128 + l32 * 128 + (((2 * kh + 1) ^ swz) * 16);
    const char* const wb10 = smem + QTM * 128 + (wn * 64) * 128 + l32 * 128 + (((4 + 2 * kh) ^ swz) * 16);
    const char* const wb11 = smem + QTM * 128 + (wn * 64) * 128 + l32 * 128 + (((5 + 2 * kh) ^ swz) * 16);
    const char* const scb = smem + kQ2ScaleOff + wave * 512 + wm * 128 + l32 * 4;
    const int ns = g.K / Q2K;                              // even (K % 256 == 0, checked at launch), >= 2

    for (int vb = blockIdx.x; vb < nbn * nbm; vb += gridDim.x) {
    int mb, nb;
    tile_of_block(vb, nbm, nbn, &mb, &nb);
    const int m0 = mb * QTM, n0 = nb * QTN;
    if (!pro_issued) {
        __builtin_amdgcn_s_barrier();
        set_tile(m0, n0);
        carry = 0;
        stage(0, 0);
        stage(1, 1);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    Frag8 wfA[2], wfB[2], xf[4];
    int scA = 0, scB = 0;
#define SS_MMA2(WF, SCV, ni, q)                                                                                                       \
    {                                                                                                                                 \
        const int sxp = ((SCV) >> (8 * (q))) & 0xff;                                                                                  \
        if (SWAP) acc[ni][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag_join(xf[q]), frag_join(WF[ni]), acc[ni][q], 0, 0, 0, sxp, 0, 127); \
        else acc[ni][q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag_join(WF[ni]), frag_join(xf[q]), acc[ni][q], 0, 0, 0, 127, 0, sxp);      \
    }
    // a quarter of a half: 2 MFMAs on xf[q]; optionally two operand DMAs of stage dma_kt (+ its two exponent-byte DMAs in the last quarter) into
    // buffer DBUF; optionally the refresh of xf[q] and half a W fragment from block RBLK of buffer RBUF (the other register set)
#define SS_Q2(WC, SCC, WN, SCN, DMA, DBUF, dma_kt, READ, RBUF, XB0, XB1, WB0, WB1, RBLK, q)                                              \
    {                                                                                                                                 \
        SS_MMA2(WC, SCC, 0, q)                                                                                                        \
        SS_MMA2(WC, SCC, 1, q)                                                                                                        \
        if constexpr (DMA) {                                                                                                          \
            SS_DMA2(2 * (q), dma_kt, smem + (DBUF) * kQ2Stage) SS_DMA2(2 * (q) + 1, dma_kt, smem + (DBUF) * kQ2Stage)                   \
            if constexpr ((q) == 3) { SS_DMA2_SC(DBUF, dma_kt, 0); SS_DMA2_SC(DBUF, dma_kt, 1); }                                      \
        }                                                                                                                             \
        if constexpr (READ) {                                                                                                         \
            constexpr int ro = (RBUF) * kQ2Stage;                                                                                     \
            xf[q].lo = *(const i32x4*)(XB0 + ro + (q) * 32 * 128);                                                                    \
            xf[q].hi = *(const i32x4*)(XB1 + ro + (q) * 32 * 128);                                                                    \
            if constexpr (((q) & 1) == 0) WN[(q) >> 1].lo = *(const i32x4*)(WB0 + ro + ((q) >> 1) * 32 * 128);                         \
            else WN[(q) >> 1].hi = *(const i32x4*)(WB1 + ro + ((q) >> 1) * 32 * 128);                                                  \
            if constexpr ((q) == 0) SCN = *(const int*)(scb + (RBUF) * (8 * 512) + (RBLK) * 256);                                      \
        }                                                                                                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                                            \
        if constexpr (DMA) __builtin_amdgcn_sched_group_barrier(0x020, (q) == 3 ? 4 : 2, 0);                                          \
        if constexpr (READ) __builtin_amdgcn_sched_group_barrier(0x100, (q) == 0 ? 4 : 3, 0);                                         \
    }
    // first half of stage BUF: block 0 from set A; set B and the row fragments <- its block 1
#define SS_H0(BUF)                                                                                                                     \
    {                                                                                                                                 \
        SS_Q2(wfA, scA, wfB, scB, false, 0, 0, true, BUF, xb10, xb11, wb10, wb11, 1, 0)                                                \
        SS_Q2(wfA, scA, wfB, scB, false, 0, 0, true, BUF, xb10, xb11, wb10, wb11, 1, 1)                                                \
        SS_Q2(wfA, scA, wfB, scB, false, 0, 0, true, BUF, xb10, xb11, wb10, wb11, 1, 2)                                                \
        SS_Q2(wfA, scA, wfB, scB, false, 0, 0, true, BUF, xb10, xb11, wb10, wb11, 1, 3)                                                \
    }
    // second half: block 1 from set B; buffer BUF takes stage dma_kt; set A and the row fragments <- block 0 of the other buffer (stage s + 1)
#define SS_H1(BUF, DMA, dma_kt, NEXT)                                                                                                  \
    {                                                                                                                                 \
        SS_Q2(wfB, scB, wfA, scA, DMA, BUF, dma_kt, NEXT, (BUF) ^ 1, xb00, xb01, wb00, wb01, 0, 0)                                      \
        SS_Q2(wfB, scB, wfA, scA, DMA, BUF, dma_kt, NEXT, (BUF) ^ 1, xb00, xb01, wb00, wb01, 0, 1)                                      \
        SS_Q2(wfB, scB, wfA, scA, DMA, BUF, dma_kt, NEXT, (BUF) ^ 1, xb00, xb01, wb00, wb01, 0, 2)                                      \
        SS_Q2(wfB, scB, wfA, scA, DMA, BUF, dma_kt, NEXT, (BUF) ^ 1, xb00, xb01, wb00, wb01, 0, 3)                                      \
    }
    // between the halves: stage s + 1 has landed (nothing younger of this tile is in flight behind it; WAITN > 0 leaves the previous tile's
    // stores in flight), my reads of stage s have completed, and everybody is here -> its buffer may be overwritten
#define SS_MID(WAITN) { wait_vmcnt<(WAITN)>(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
#define SS_PAIR(CARRY)                                                                                                                 \
    {                                                                                                                                 \
        SS_H0(0) SS_MID((CARRY) ? kCarry : 0) SS_H1(0, true, s + 2, true)                                                             \
        SS_H0(1) SS_MID(0) SS_H1(1, true, s + 3, true)                                                                                \
    }

    // prologue: stage 0 has landed (stage 1 and, after an early prologue, the previous tile's last stores stay in flight); block 0 -> set A
    if (carry) wait_vmcnt<OPS + kCarry>(); else wait_vmcnt<OPS>();
    __builtin_amdgcn_s_barrier();
    {
#pragma unroll
        for (int i = 0; i < 2; i++) { wfA[i].lo = *(const i32x4*)(wb00 + i * 32 * 128); wfA[i].hi = *(const i32x4*)(wb01 + i * 32 * 128); }
#pragma unroll
        for (int i = 0; i < 4; i++) { xf[i].lo = *(const i32x4*)(xb00 + i * 32 * 128); xf[i].hi = *(const i32x4*)(xb01 + i * 32 * 128); }
        scA = *(const int*)scb;
    }
    int s = 0;
    if (carry && ns > 2) { SS_PAIR(true) s = 2; }
    for (; s + 3 < ns; s += 2) SS_PAIR(false)
    {   // last pair (ONE unconditional block: a branch here lets LLVM sink the MFMAs below it and spill every fragment on the way): nothing left to
        // stage, its second stage has no successor.  Its wait never leaves stores in flight (only matters when K = 256 and the tile was issued early)
        SS_H0(0) SS_MID(0) SS_H1(0, false, 0, true)
        SS_H0(1) SS_H1(1, false, 0, false)
    }
#undef SS_PAIR
#undef SS_MID
#undef SS_H1
#undef SS_H0
#undef SS_Q2
#undef SS_MMA2
    // the accumulators must exist HERE (see gemm_f8_kernel)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" : "+v"(acc[i][j]));

    f8_fold_scale_bias<KIND>(g, acc, n0, wn, l32, kh);
    pro_issued = false;
    {
        const int vbn = vb + gridDim.x;
        if (vbn < nbn * nbm) {
            int mbn, nbn2;
            tile_of_block(vbn, nbm, nbn, &mbn, &nbn2);
            __builtin_amdgcn_s_barrier();   // every wave has read its last fragments: both buffers are free
            set_tile(mbn * QTM, nbn2 * QTN);
            stage(0, 0);
            stage(1, 1);
            carry = (m0 + QTM <= g.M) ? kCarry : 0;
            pro_issued = true;
        }
    }
    f8_epilogue<T, KIND>(g, acc, m0, n0, wm, wn, l32, kh);
    }  // tile loop
#undef SS_DMA2
#undef SS_DMA2_SC
}

int g_f8_k128 = -1;   // env SS_F8_K128, read at the first launch: 0 = the 64-byte-row main loop (gemm_f8_kernel; A/B reference), default 1

template <typename T, int KIND>
static void launch_f8_kind(const GemmF8Desc& g, hipStream_t st) {
    static std::atomic<uint64_t> attr{0};
    once_per_device(attr, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm_f8_kernel<T, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, kQLds)); });
    int n_cu = device_cu_count() / 8 * 8;
    if (n_cu < 8) n_cu = 8;
    const int nwg = (g.N / QTN) * ((g.M + QTM - 1) / QTM);
#ifdef SS_DEV_KNOCKOUTS   // NOT in the product build (speaksense_amd/build.py): tools/gemm_fp8_ko.py compiles its own library with -DSS_DEV_KNOCKOUTS
    if constexpr (KIND == F8_STORE_T) {   // dev tool: knock-out variants of the k loop (wrong results, timing only)
        static const int ko = [] {
            const int v = getenv("SS_F8_KO") ? atoi(getenv("SS_F8_KO")) : 0;
            if (v) fprintf(stderr, "[ss] SS_F8_KO=%d: parts of the e4m3 GEMM's k loop are knocked out -- its results are WRONG by construction (timing tool only)\n", v);
            return v;
        }();
#define SS_KO_CASE(V) case V: { static std::atomic<uint64_t> a{0}; once_per_device(a, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm_f8_kernel<T, KIND, V>, hipFuncAttributeMaxDynamicSharedMemorySize, kQLds)); }); \
                                gemm_f8_kernel<T, KIND, V><<<nwg < n_cu ? nwg : n_cu, 512, kQLds, st>>>(g); SS_LAUNCH_CHECK(); return; }
        switch (ko) { SS_KO_CASE(1) SS_KO_CASE(2) SS_KO_CASE(3) SS_KO_CASE(4) SS_KO_CASE(7) SS_KO_CASE(8) SS_KO_CASE(15) SS_KO_CASE(16) SS_KO_CASE(20) SS_KO_CASE(28) default: break; }
#undef SS_KO_CASE
    }
#endif
    if (g_f8_k128 < 0) { const char* e = getenv("SS_F8_K128"); g_f8_k128 = e ? atoi(e) : 1; }
    if (g_f8_k128) {
        static std::atomic<uint64_t> attr2{0};
        once_per_device(attr2, [] { SS_HIP(hipFuncSetAttribute((const void*)gemm_f8k128_kernel<T, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, kQ2Lds)); });
        gemm_f8k128_kernel<T, KIND><<<nwg < n_cu ? nwg : n_cu, 512, kQ2Lds, st>>>(g); SS_LAUNCH_CHECK();
        return;
    }
    gemm_f8_kernel<T, KIND><<<nwg < n_cu ? nwg : n_cu, 512, kQLds, st>>>(g); SS_LAUNCH_CHECK();
}

template <typename T>
void launch_gemm_f8(const GemmF8Desc& g_in, hipStream_t st) {
    GemmF8Desc g = g_in;
    if (g.cache_rows <= 0) g.cache_rows = g.rows_per_batch;
    if (g.M <= 0 || g.N % QTN || g.K % (QNST * QTK) || g.lda % 16 || g.ldsc % 256 || g.ldsc < ((g.M + 255) & ~255))
        throw Error(-1, "fp8 gemm: N and K must be multiples of 256, lda of 16, and the exponent-byte pitch a multiple of 256 >= M");
    if ((long)g.M * g.lda >= (1L << 32) || (long)g.N * g.K >= (1L << 32)) throw Error(-1, "fp8 gemm: operand larger than 4 GiB");
    switch (g.kind) {
        case F8_STORE_T: launch_f8_kind<T, F8_STORE_T>(g, st); break;
        case F8_GELU_F8:
            if (!g.out_scale) throw Error(-1, "fp8 gemm: the e4m3 output needs its exponent-byte buffer");
            if (g.ld_osc % 256 || g.ld_osc < ((g.M + 255) & ~255)) throw Error(-1, "fp8 gemm: bad output exponent-byte pitch");
            launch_f8_kind<T, F8_GELU_F8>(g, st); break;
        case F8_RES_F32: launch_f8_kind<T, F8_RES_F32>(g, st); break;
        case F8_VT: launch_f8_kind<T, F8_VT>(g, st); break;
        case F8_CROSS_KV: launch_f8_kind<T, F8_CROSS_KV>(g, st); break;
        case F8_STORE_F32: launch_f8_kind<T, F8_STORE_F32>(g, st); break;
        case F8_CROSS_KV8:
            if (!g.out_scale || g.d % 64 || g.N % (2 * g.d)) throw Error(-1, "fp8 gemm: the e4m3 cross cache needs its exponent-byte buffer and N = layers * 2 * d");
            launch_f8_kind<T, F8_CROSS_KV8>(g, st); break;
        default: throw Error(-1, "fp8 gemm: bad epilogue kind");
    }
}
template void launch_gemm_f8<bf16>(const GemmF8Desc&, hipStream_t);
template void launch_gemm_f8<f16>(const GemmF8Desc&, hipStream_t);

// ---------------------------------------------------------------------------------------------
// producers of quantised activations
// ---------------------------------------------------------------------------------------------
// LayerNorm (f32 residual stream -> e4m3 + exponent bytes): one wave per row as layernorm_kernel; a 64-column block is the 16 lanes of
// one DPP row at a fixed load index, so the block maximum is four DPP steps.
template <int NI>
__global__ __launch_bounds__(256) void layernorm_f8_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                           unsigned char* __restrict__ y8, unsigned char* __restrict__ ysc, long ldsc, int rows, int d) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * d;
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
    f32x4 ww[NI], bb[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) { ww[i] = *(const f32x4*)(w + cc[i]); bb[i] = *(const f32x4*)(b + cc[i]); }
#pragma unroll
    for (int i = 0; i < NI; i++) {
        f32x4 o;
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) { o[e] = ok[i] ? v[i][e] * scale * ww[i][e] + bb[i][e] : 0.f; amax = fmaxf(amax, fabsf(o[e])); }
        amax = fmaxf(amax, dpp_mov<kDppRor8>(amax));
        amax = fmaxf(amax, dpp_mov<kDppRor4>(amax));
        amax = fmaxf(amax, dpp_mov<kDppXor2>(amax));
        amax = fmaxf(amax, dpp_mov<kDppXor1>(amax));
        const int eb = e8m0_for_amax(amax);
        const float inv = pow2_neg_of_e8m0(eb);
        if (ok[i]) {
            *(unsigned*)(y8 + (long)row * d + cc[i]) = pack_e4m3x4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
            if ((lane & 15) == 0) ysc[f8_scale_index(row, i * 4 + (lane >> 4), ldsc)] = (unsigned char)eb;
        }
    }
}
void launch_layernorm_f8(const float* x, const float* w, const float* b, unsigned char* y8, unsigned char* y_scale, long ldsc, int rows, int d, hipStream_t st) {
    if (d > 2048 || d % 64) throw Error(-1, "layernorm_f8: d must be <= 2048 and a multiple of 64");
    const int grid = (rows + 3) / 4;
    if (d <= 512) { layernorm_f8_kernel<2><<<grid, 256, 0, st>>>(x, w, b, y8, y_scale, ldsc, rows, d); SS_LAUNCH_CHECK(); }
    else if (d <= 1280) { layernorm_f8_kernel<5><<<grid, 256, 0, st>>>(x, w, b, y8, y_scale, ldsc, rows, d); SS_LAUNCH_CHECK(); }
    else { layernorm_f8_kernel<8><<<grid, 256, 0, st>>>(x, w, b, y8, y_scale, ldsc, rows, d); SS_LAUNCH_CHECK(); }
}

// T [rows][ldx] -> e4m3 + exponent bytes (the attention output feeding the out-projection): one wave per row, 4 columns per lane per pass
template <typename T>
__global__ __launch_bounds__(256) void quantize_f8_kernel(const T* __restrict__ x, long ldx, unsigned char* __restrict__ y8, unsigned char* __restrict__ ysc, long ldsc,
                                                          int rows, int d) {
    typedef typename Mfma<T>::V4 V4;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    for (int c0 = 0; c0 < d; c0 += 256) {
        const int c = c0 + lane * 4;
        const bool ok = c < d;
        V4 in = {};
        if (ok) in = *(const V4*)(x + (long)row * ldx + c);
        float o[4], amax = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) { o[e] = (float)in[e]; amax = fmaxf(amax, fabsf(o[e])); }
        amax = fmaxf(amax, dpp_mov<kDppRor8>(amax));
        amax = fmaxf(amax, dpp_mov<kDppRor4>(amax));
        amax = fmaxf(amax, dpp_mov<kDppXor2>(amax));
        amax = fmaxf(amax, dpp_mov<kDppXor1>(amax));
        const int eb = e8m0_for_amax(amax);
        const float inv = pow2_neg_of_e8m0(eb);
        if (ok) {
            *(unsigned*)(y8 + (long)row * d + c) = pack_e4m3x4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
            if ((lane & 15) == 0) ysc[f8_scale_index(row, c >> 6, ldsc)] = (unsigned char)eb;
        }
    }
}
template <typename T>
void launch_quantize_f8(const T* x, long ldx, unsigned char* y8, unsigned char* y_scale, long ldsc, int rows, int d, hipStream_t st) {
    if (d % 64) throw Error(-1, "quantize_f8: d must be a multiple of 64");
    quantize_f8_kernel<T><<<(rows + 3) / 4, 256, 0, st>>>(x, ldx, y8, y_scale, ldsc, rows, d); SS_LAUNCH_CHECK();
}
template void launch_quantize_f8<bf16>(const bf16*, long, unsigned char*, unsigned char*, long, int, int, hipStream_t);
template void launch_quantize_f8<f16>(const f16*, long, unsigned char*, unsigned char*, long, int, int, hipStream_t);

// ---------------------------------------------------------------------------------------------
// self-test (ss_engine_selftest_gemm_f8): the tiled kernel against a one-thread-per-output reference on seeded e4m3 codes and exponent bytes
// ---------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ unsigned st_hash(size_t i, unsigned seed) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
    return x;
}
__global__ void st8_fill_codes(unsigned char* p, size_t n, unsigned seed) {   // any finite e4m3 code (0x7f / 0xff are NaN)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned c = st_hash(i, seed) & 0xff;
        if ((c & 0x7f) == 0x7f) c ^= 1;
        if ((c & 0x7f) > 0x5f) c -= 0x20;   // keep magnitudes <= 60 so sums stay well inside f16 outputs
        p[i] = (unsigned char)c;
    }
}
__global__ void st8_fill_scales(unsigned char* p, long ldsc, int M, int nblk, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)M * nblk; i += (size_t)gridDim.x * blockDim.x) {
        const long m = i / nblk, blk = i % nblk;
        p[f8_scale_index(m, blk, ldsc)] = (unsigned char)(118 + st_hash(i, seed) % 6);   // 2^-9 .. 2^-4
    }
}
__global__ void st8_fill_f32(float* p, size_t n, unsigned seed, float lo, float hi) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = lo + (hi - lo) * ((st_hash(i, seed) & 0xffff) / 65536.0f);
}
__device__ __forceinline__ float e4m3_value(unsigned char c) { return __builtin_amdgcn_cvt_f32_fp8((int)c, 0); }
__global__ void st8_ref(const unsigned char* A, const unsigned char* asc, long ldsc, const unsigned char* W, const float* ws, const float* bias, const float* res, float* C,
                        int M, int N, int K, int kind) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0.f;
    for (int blk = 0; blk < K / 64; blk++) {
        float part = 0.f;
        for (int k = blk * 64; k < blk * 64 + 64; k++) part += e4m3_value(A[(long)m * K + k]) * e4m3_value(W[(long)n * K + k]);
        acc += part * __builtin_bit_cast(float, (unsigned)asc[f8_scale_index(m, blk, ldsc)] << 23);
    }
    acc = acc * ws[n] + bias[n];
    if (kind == F8_GELU_F8) acc = gelu_tanh_f(acc);
    if (kind == F8_RES_F32) acc += res[(long)m * N + n];
    C[(long)m * N + n] = acc;
}
// dequantise an e4m3 + exponent-byte matrix (the F8_GELU_F8 output) to f32
__global__ void st8_dequant(const unsigned char* y8, const unsigned char* ysc, long ldsc, float* out, int M, int N) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)M * N; i += (size_t)gridDim.x * blockDim.x) {
        const long m = i / N, n = i % N;
        out[i] = e4m3_value(y8[i]) * __builtin_bit_cast(float, (unsigned)ysc[f8_scale_index(m, n >> 6, ldsc)] << 23);
    }
}
template <typename T>
__global__ void st8_diff(const void* out, const float* ref, size_t n, int f32out, float rel_quant, float* maxes /* [2]: max excess |diff|, max |ref| */) {
    float d = 0.f, r = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float o = f32out ? ((const float*)out)[i] : (float)((const T*)out)[i];
        // rel_quant > 0: the output itself is e4m3 (3 mantissa bits, block exponent): subtract the quantisation step it may legitimately be off by
        float e = fabsf(o - ref[i]);
        if (rel_quant > 0.f) e = fmaxf(0.f, e - rel_quant * fabsf(ref[i]));
        d = fmaxf(d, e); r = fmaxf(r, fabsf(ref[i]));
    }
    atomicMax((int*)maxes, __float_as_int(d));
    atomicMax((int*)maxes + 1, __float_as_int(r));
}
}  // namespace

template <typename T>
void gemm_f8_selftest(int M, int N, int K, int kind, float* max_err, float* max_ref, hipStream_t st, int reps, float* avg_ms) {
    if (kind != F8_STORE_T && kind != F8_GELU_F8 && kind != F8_RES_F32 && kind != F8_STORE_F32) throw Error(-1, "fp8 gemm selftest: kind not covered");
    unsigned char *A, *W, *asc, *o8 = nullptr, *osc = nullptr; float *ws, *bias, *res, *ref, *mx, *deq = nullptr; void* out;
    const long ldsc = (M + 255) & ~255L;
    const int nblk = K / 64;
    SS_HIP(hipMalloc(&A, (size_t)M * K)); SS_HIP(hipMalloc(&W, (size_t)N * K)); SS_HIP(hipMalloc(&asc, (size_t)ldsc * nblk));
    SS_HIP(hipMalloc(&ws, (size_t)N * 4)); SS_HIP(hipMalloc(&bias, (size_t)N * 4));
    SS_HIP(hipMalloc(&res, (size_t)M * N * 4)); SS_HIP(hipMalloc(&ref, (size_t)M * N * 4)); SS_HIP(hipMalloc(&out, (size_t)M * N * 4)); SS_HIP(hipMalloc(&mx, 8));
    SS_HIP(hipMemsetAsync(asc, 127, (size_t)ldsc * nblk, st));
    st8_fill_codes<<<1024, 256, 0, st>>>(A, (size_t)M * K, 21);
    st8_fill_codes<<<1024, 256, 0, st>>>(W, (size_t)N * K, 22);
    st8_fill_scales<<<1024, 256, 0, st>>>(asc, ldsc, M, nblk, 23);
    st8_fill_f32<<<64, 256, 0, st>>>(ws, (size_t)N, 24, 0.02f, 0.04f);
    st8_fill_f32<<<64, 256, 0, st>>>(bias, (size_t)N, 25, -0.5f, 0.5f);
    st8_fill_f32<<<1024, 256, 0, st>>>(res, (size_t)M * N, 26, -2.0f, 2.0f);
    SS_HIP(hipMemsetAsync(mx, 0, 8, st));
    st8_ref<<<dim3((N + 255) / 256, M), 256, 0, st>>>(A, asc, ldsc, W, ws, bias, res, ref, M, N, K, kind);
    GemmF8Desc g{};
    g.A = A; g.lda = K; g.a_scale = asc; g.ldsc = ldsc; g.W = W; g.w_scale = ws; g.M = M; g.N = N; g.K = K; g.kind = kind; g.bias = bias; g.out = out; g.ldo = N;
    g.scale = 1.0f; g.rows_per_batch = 1500;
    if (kind == F8_RES_F32) { SS_HIP(hipMemcpyAsync(out, res, (size_t)M * N * 4, hipMemcpyDeviceToDevice, st)); g.res = (const float*)out; }
    if (kind == F8_GELU_F8) {
        SS_HIP(hipMalloc(&o8, (size_t)M * N)); SS_HIP(hipMalloc(&osc, (size_t)ldsc * (N / 64))); SS_HIP(hipMalloc(&deq, (size_t)M * N * 4));
        g.out = o8; g.out_scale = osc; g.ld_osc = ldsc;
    }
    launch_gemm_f8<T>(g, st);
    if (kind == F8_GELU_F8) {
        st8_dequant<<<1024, 256, 0, st>>>(o8, osc, ldsc, deq, M, N);
        // an e4m3 output is within half a step of 2^-3 relative to the block maximum; checked as: |diff| - 0.0625 |ref| - (tiny) <= tolerance of the f32 path
        st8_diff<T><<<1024, 256, 0, st>>>(deq, ref, (size_t)M * N, 1, 0.0626f, mx);
    } else {
        st8_diff<T><<<1024, 256, 0, st>>>(out, ref, (size_t)M * N, kind == F8_RES_F32 || kind == F8_STORE_F32, 0.0f, mx);
    }
    float h[2];
    SS_HIP(hipMemcpyAsync(h, mx, 8, hipMemcpyDeviceToHost, st));
    SS_HIP(hipStreamSynchronize(st));
    *max_err = h[0]; *max_ref = h[1];
    if (reps > 0 && avg_ms) {
        hipEvent_t e0, e1;
        SS_HIP(hipEventCreate(&e0)); SS_HIP(hipEventCreate(&e1));
        SS_HIP(hipEventRecord(e0, st));
        for (int i = 0; i < reps; i++) launch_gemm_f8<T>(g, st);
        SS_HIP(hipEventRecord(e1, st));
        SS_HIP(hipEventSynchronize(e1));
        float ms = 0.f;
        SS_HIP(hipEventElapsedTime(&ms, e0, e1));
        *avg_ms = ms / reps;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(asc); (void)hipFree(ws); (void)hipFree(bias); (void)hipFree(res); (void)hipFree(ref); (void)hipFree(out); (void)hipFree(mx);
    if (o8) (void)hipFree(o8); if (osc) (void)hipFree(osc); if (deq) (void)hipFree(deq);
}
template void gemm_f8_selftest<bf16>(int, int, int, int, float*, float*, hipStream_t, int, float*);
template void gemm_f8_selftest<f16>(int, int, int, int, float*, float*, hipStream_t, int, float*);

}  // namespace ss
