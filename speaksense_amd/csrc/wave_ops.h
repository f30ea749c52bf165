// Wave64 cross-lane reductions on the VALU for gfx950.
// hipcc lowers every __shfl_xor to ds_bpermute_b32 -- a round trip through the LDS pipe (~100 clk in a dependency chain).  The decode
// kernels are latency-bound and reduce inside their inner loops, so these use DPP row operations (lane ^ 1, ^ 2 by quad_perm; ^ 4, ^ 8 by
// row rotations, which coincide with the xor once the higher bits are already reduced) and the gfx950 v_permlane16_swap /
// v_permlane32_swap for the two cross-row steps.  Each helper adds / maxes exactly the operand pairs of the __shfl_xor butterfly it
// replaces, in the same order, so results are bit-identical to the shuffle version.
#pragma once
#include <hip/hip_runtime.h>

namespace ss {

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppRor4 = 0x124, kDppRor8 = 0x128, kDppHalfMirror = 0x141;

struct SwapPair { float a, b; };
// v_permlane{16,32}_swap exchange the odd rows of the first register with the even rows of the second; fed two copies of v they leave
// {v of my row pair's even row, v of its odd row} in every lane.  Written as inline asm: with ROCm 7.2's hipcc the
// __builtin_amdgcn_permlane*_swap builtins return a pair whose second element is folded to the first (op(r[0], r[1]) became
// op(r[0], r[0]) in the generated code; tools/waveops_test.cpp).  The s_nop covers the VALU-write -> permlane-read hazard the
// compiler would otherwise pad itself.
__device__ __forceinline__ SwapPair swap16(float v) {
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return {x, y};
}
__device__ __forceinline__ SwapPair swap32(float v) {
    float x = v, y = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    return {x, y};
}

// all-reduce over the 64 lanes, butterfly order 32, 16, 8, 4, 2, 1
__device__ __forceinline__ float wave_sum(float v) {
    SwapPair p = swap32(v); v = p.a + p.b;
    p = swap16(v); v = p.a + p.b;
    v += dpp_mov<kDppRor8>(v);
    v += dpp_mov<kDppRor4>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppXor1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    SwapPair p = swap32(v); v = fmaxf(p.a, p.b);
    p = swap16(v); v = fmaxf(p.a, p.b);
    v = fmaxf(v, dpp_mov<kDppRor8>(v));
    v = fmaxf(v, dpp_mov<kDppRor4>(v));
    v = fmaxf(v, dpp_mov<kDppXor2>(v));
    v = fmaxf(v, dpp_mov<kDppXor1>(v));
    return v;
}
// sum over the 8 lanes that share lane >> 3 (butterfly 1, 2, 4); every lane of the group gets the total
__device__ __forceinline__ float sum_lanes8(float v) {
    v += dpp_mov<kDppXor1>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppHalfMirror>(v);   // lane i <- 7 - i: the other quad of the group, whose four lanes already agree
    return v;
}
// sum over the 8 lanes that share lane & 7 (butterfly 8, 16, 32)
__device__ __forceinline__ float sum_stride8(float v) {
    v += dpp_mov<kDppRor8>(v);
    SwapPair p = swap16(v); v = p.a + p.b;
    p = swap32(v); v = p.a + p.b;
    return v;
}
// lane ^ 16 then lane ^ 32 (the four 16-lane rows)
__device__ __forceinline__ float rows_max(float v) {
    SwapPair p = swap16(v); v = fmaxf(p.a, p.b);
    p = swap32(v); return fmaxf(p.a, p.b);
}
__device__ __forceinline__ float rows_sum(float v) {
    SwapPair p = swap16(v); v = p.a + p.b;
    p = swap32(v); return p.a + p.b;
}

}  // namespace ss
