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

// v_permlane{16,32}_swap exchange the odd rows of the first register with the even rows of the second; fed two copies of v they leave
// {v of my row pair's even row, v of its odd row} in every lane, which the same asm block then combines.  Inline asm for two reasons:
// with ROCm 7.2's hipcc the __builtin_amdgcn_permlane*_swap builtins return a pair whose second element is folded to the first
// (op(r[0], r[1]) became op(r[0], r[0]); tools/waveops_test.cpp), and fmaxf on values the compiler cannot see through costs two extra
// canonicalising v_max per step.  The s_nops cover the VALU-write -> permlane-read and permlane-write -> VALU-read hazards.
#define SS_SWAP_OP(NAME, SWAP, OP)                                                                                           \
    __device__ __forceinline__ float NAME(float v) {                                                                         \
        float y;                                                                                                             \
        asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\t" SWAP " %0, %1\n\ts_nop 0\n\t" OP " %0, %0, %1" : "+v"(v), "=&v"(y)); \
        return v;                                                                                                            \
    }
SS_SWAP_OP(swap16_sum, "v_permlane16_swap_b32", "v_add_f32")
SS_SWAP_OP(swap32_sum, "v_permlane32_swap_b32", "v_add_f32")
SS_SWAP_OP(swap16_max, "v_permlane16_swap_b32", "v_max_f32")
SS_SWAP_OP(swap32_max, "v_permlane32_swap_b32", "v_max_f32")
#undef SS_SWAP_OP

// all-reduce over the 64 lanes, butterfly order 32, 16, 8, 4, 2, 1
__device__ __forceinline__ float wave_sum(float v) {
    v = swap32_sum(v);
    v = swap16_sum(v);
    v += dpp_mov<kDppRor8>(v);
    v += dpp_mov<kDppRor4>(v);
    v += dpp_mov<kDppXor2>(v);
    v += dpp_mov<kDppXor1>(v);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = swap32_max(v);
    v = swap16_max(v);
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
    v = swap16_sum(v);
    return swap32_sum(v);
}
// lane ^ 16 then lane ^ 32 (the four 16-lane rows)
__device__ __forceinline__ float rows_max(float v) {
    return swap32_max(swap16_max(v));
}
__device__ __forceinline__ float rows_sum(float v) {
    return swap32_sum(swap16_sum(v));
}

}  // namespace ss
