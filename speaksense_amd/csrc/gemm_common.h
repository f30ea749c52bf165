// Shared by the MFMA GEMM translation units (kernels_gemm.hip: f16 / bf16 operands, kernels_gemm_fp8.hip: e4m3 operands):
// MFMA wrappers, the GELU used by the fused epilogues, the XCD-aware tile rasterisation, the LDS-DMA and counted-wait helpers.
#pragma once
#include <type_traits>

#include "kernels.h"

namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<bf16> {
    typedef bf16x8 V8; typedef bf16x4 V4;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma<f16> {
    typedef f16x8 V8; typedef f16x4 V4;
    static __device__ __forceinline__ f32x4 mma(V8 a, V8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)));  tanh(u) = 1 - 2/(exp(2u)+1)
    // == x * sigmoid(2u): one v_exp_f32 + one v_rcp_f32
    const float u = 0.79788456080286535588f * x * (1.0f + 0.044715f * x * x);
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
}
// The same GELU for the four accumulator values a lane owns, written on float2 so that hipcc emits packed-f32 VALU (v_pk_mul_f32 / v_pk_fma_f32 /
// v_pk_add_f32: two elements per full-rate instruction) and with the constants folded: per PAIR 1 add (bias, done by the caller) + 3 mul + 1 fma
// + 1 add packed, 2 v_exp_f32 + 2 v_rcp_f32, and -- ROUND_IN, ggml's f16 GELU table input -- one v_cvt_pk_f16_f32 (RNE) + two v_cvt_f32_f16.
// The scalar form above compiled to 6 v_mul + 2 v_add + 1 v_fma + 1 v_cndmask per ELEMENT beside the two transcendentals (r04: FC1 epilogue
// ISA), i.e. ~76 issue cycles per element against ~52 here.  exp(-2u) = 2^(x * (C1 + C3 x^2)), C1 = -2 sqrt(2/pi) log2(e), C3 = 0.044715 C1.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
template <bool ROUND_IN>
__device__ __forceinline__ f32x2 gelu_tanh_pk(f32x2 x) {
    if constexpr (ROUND_IN) x = __builtin_convertvector(__builtin_convertvector(x, h16x2), f32x2);
    constexpr float C1 = -2.0f * 0.79788456080286535588f * 1.44269504088896340736f, C3 = 0.044715f * C1;
    const f32x2 x2 = x * x;
    const f32x2 t = x2 * (f32x2){C3, C3} + (f32x2){C1, C1};
    const f32x2 z = x * t;
    f32x2 e = {__builtin_amdgcn_exp2f(z[0]), __builtin_amdgcn_exp2f(z[1])};
    e += (f32x2){1.0f, 1.0f};
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return x * r;
}
// four values -> four T (f16: packed RNE converts); round_in is warp-uniform (a kernel argument): a scalar branch, not a select per element
template <typename T>
__device__ __forceinline__ typename Mfma<T>::V4 gelu4(f32x4 v, int round_in) {
    f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    if (std::is_same<T, f16>::value && round_in) { a = gelu_tanh_pk<true>(a); b = gelu_tanh_pk<true>(b); }
    else { a = gelu_tanh_pk<false>(a); b = gelu_tanh_pk<false>(b); }
    typename Mfma<T>::V4 o;
    o[0] = (T)a[0]; o[1] = (T)a[1]; o[2] = (T)b[0]; o[3] = (T)b[1];
    return o;
}
template <typename T> __device__ __forceinline__ float gelu_in_round(float x, int on);
template <> __device__ __forceinline__ float gelu_in_round<bf16>(float x, int) { return x; }
template <> __device__ __forceinline__ float gelu_in_round<f16>(float x, int on) { return on ? (float)(f16)x : x; }

// Tile rasterisation.  The dispatcher places workgroup b on XCD b % 8 (observed; used for speed only), each XCD has a private
// 4 MB L2 and runs ~32 tiles at a time.  (1) bijective remap: every XCD gets a contiguous range of logical ids;
// (2) logical ids walk the grid in column groups of GN n-panels with m fastest inside a group, so the ~32 concurrent tiles of
// an XCD form a ~GN x (32/GN) patch: fabric traffic per launch ~ A_bytes*nbn/GN + W_bytes*nbm*GN/32, minimal near GN = sqrt(32).
// With plain n-fastest order an XCD swept all nbn weight panels (13 MB for FC1 >> L2) for every ~1.6 tile rows: rocprofv3
// FETCH_SIZE showed 460 MB per FC1 launch against 44 MB of operands, and the kernel sat on the ~10 B/clk/CU miss rate.
__device__ __forceinline__ void tile_of_block(int bid, int nbm, int nbn, int* mb, int* nb) {
    const int nwg = nbm * nbn;
    {
        const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int GN = nbn;
    if (nbn > 8) {
        GN = 5;
        if (nbn % 5 != 0) { if (nbn % 6 == 0) GN = 6; else if (nbn % 4 == 0) GN = 4; else if (nbn % 7 == 0) GN = 7; }
    }
    const int per_group = nbm * GN, full = nbn / GN;
    int g = bid / per_group, rem = bid - g * per_group, gn = GN;
    if (g >= full) { g = full; rem = bid - full * per_group; gn = nbn - full * GN; }
    *mb = rem / gn;
    *nb = g * GN + rem % gn;
}

// Element offset of operand / output row m.  rpb == 0 (every GEMM but the conv stem): plain rows, m * ld.  Otherwise the rows come in batches of
// rpb (a window of the padded time-major conv input): 32-bit division.  rpb is a kernel argument, so the test is a scalar branch.  (Written with
// 64-bit `/` and `%` on a "no batching" sentinel of 2^40 this cost ~120 instructions per row and 24 rows per tile and wave: 3 VALU per MFMA.)
__device__ __forceinline__ long row_off(long m, int rpb, long batch_stride, long ld) {
    if (rpb == 0) return m * ld;
    const unsigned b = (unsigned)m / (unsigned)rpb;
    return (long)b * batch_stride + (long)((unsigned)m - b * (unsigned)rpb) * ld;
}

template <typename T>
__device__ __forceinline__ void glds16(const T* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

}  // namespace ss
