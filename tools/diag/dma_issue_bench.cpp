// What does one LDS-DMA instruction (global_load_lds_dwordx4, 1 KB per wave) cost the issuing wave, as a function of how its 64 lanes' 16-byte
// pieces are laid out in global memory?  (dev tool, run on the GPU box; DESIGN.md section 8a: the staging waves' DMA issue is the critical path of
// the 256 x 256 GEMM's main loop)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/diag/dma_issue_bench.cpp -o tools/diag/dma_issue_bench.bin
// Shapes: lanes-per-row L in {4, 8, 16, 64}: a wave instruction touches 64 / L rows of 16 L contiguous bytes each (row pitch 2560 B = a K = 1280
// f16 operand).  L = 4 is what gemm256_kernel issues (TK = 32: 64-byte rows), L = 8 a TK = 64 layout, L = 64 one contiguous kilobyte.
// Reported: shader cycles per instruction between the first and the last issue of a burst of 32 (s_memtime), 1 / 2 / 4 / 8 waves per workgroup
// issuing at once, one workgroup per CU on all CUs, source L2-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int L>
__global__ __launch_bounds__(512) void k(const char* src, long long* out, int n_waves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= n_waves) return;
    const int row = lane / L, piece = lane % L;
    const char* p = src + ((long)blockIdx.x * 4096 + wave * 512 + row) * 2560 + piece * 16;   // distinct rows per wave and workgroup
    char* dst = smem + wave * 1024 * 16;
    long long t0 = 0, t1 = 0;
    for (int rep = 0; rep < 3; rep++) {     // the last repetition is the one reported (instruction cache, L2 warm)
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_sched_barrier(0);
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 32; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + (long)i * 64 * 2560),
                                             (__attribute__((address_space(3))) void*)(dst + (i & 15) * 1024), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int L>
static void run(const char* src, long long* d_out, int n_cu) {
    for (int nw : {1, 2, 4, 8}) {
        CK(hipFuncSetAttribute((const void*)k<L>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 1024 * 16));
        CK(hipMemset(d_out, 0, n_cu * 8 * 8));
        k<L><<<n_cu, 512, 8 * 1024 * 16, 0>>>(src, d_out, nw);
        CK(hipDeviceSynchronize());
        std::vector<long long> h(n_cu * 8);
        CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
        double s = 0; int n = 0; long long mx = 0;
        for (int b = 0; b < n_cu; b++) for (int w = 0; w < nw; w++) { s += h[b * 8 + w]; n++; if (h[b * 8 + w] > mx) mx = h[b * 8 + w]; }
        printf("  lanes per row %2d (%4d-byte rows), %d wave(s) per CU issuing: %6.1f cycles per instruction (slowest wave %6.1f)\n", L, 16 * L, nw, s / n / 32, mx / 32.0);
    }
}

int main() {
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int n_cu = pr.multiProcessorCount;
    char* src; long long* d_out;
    const size_t bytes = (size_t)n_cu * 4096 * 2560 + (size_t)33 * 64 * 2560;
    CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes)); CK(hipMalloc(&d_out, n_cu * 8 * 8));
    printf("%s, %d CUs; s_memtime ticks at the shader clock\n", pr.name, n_cu);
    run<4>(src, d_out, n_cu); run<8>(src, d_out, n_cu); run<16>(src, d_out, n_cu); run<64>(src, d_out, n_cu);
    return 0;
}
