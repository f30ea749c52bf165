// Dev tool (round 6): how fast can ONE CU get a 256 x 256 output tile out to memory, by request shape, row pitch, cache policy and number of CUs storing?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/diag/store_rate_bench.cpp -o tools/diag/store_rate_bench.bin
// The GEMM epilogues (kernels_gemm.hip epilogue256) move a 128 KB (f16) / 256 + 256 KB (f32 residual) tile per CU at 12 - 14 B/clk whatever was tried there
// (deeper operand prefetch, a start stagger: tools/experiments/r06_gemm_stagger/).  This isolates the store path: one 512-thread workgroup per CU (96 KB of dynamic
// LDS keeps a second one off the CU), every workgroup writes TILES tiles of 256 rows x 512 bytes into a matrix of row pitch `pitch`, s_memtime around the stores
// + a final s_waitcnt vmcnt(0).  Shapes (bytes per lane x lanes per row segment): 16 x 4 = the epilogue's (16 row segments of 64 B per wave instruction),
// 16 x 8 (8 segments of 128 B = whole lines), 16 x 32 (2 segments of 512 B), and the same with non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int LPR, bool NT>   // LPR = lanes per row segment (4: 64 B, 8: 128 B, 32: 512 B)
__global__ __launch_bounds__(512) void store_tiles(unsigned char* out, long pitch, int tiles_per_wg, int n_col_tiles, long long* cycles) {
    extern __shared__ char pad[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int RPI = 64 / LPR;                       // rows one wave instruction covers
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < tiles_per_wg; t++) {
        const long tile = (long)blockIdx.x + (long)t * gridDim.x;
        const long row0 = (tile / n_col_tiles) * 256, col0 = (tile % n_col_tiles) * 512;
        // a wave owns 32 rows x 512 B... as the epilogue: wave (wm, wn) = 128 rows x 128 B; here simply rows [wave * 32, +32), all 512 B, walked in instructions
        // of RPI rows x (LPR * 16) B
        for (int r = 0; r < 32; r += RPI)
            for (int c = 0; c < 512; c += LPR * 16) {
                unsigned char* p = out + (row0 + wave * 32 + r + lane / LPR) * pitch + col0 + c + (lane % LPR) * 16;
                if (NT) __builtin_nontemporal_store(v, (u32x4*)p); else *(u32x4*)p = v;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (pad[tid] == 77) out[0] = 1;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int kMaxCus = 256, kTiles = 8;
    long long* cyc; CK(hipMalloc(&cyc, 256 * 8));
    printf("%-28s %6s %8s | %10s %10s %12s\n", "request shape", "CUs", "pitch B", "us/launch", "GB/s chip", "B/clk per CU");
    for (long pitch : {10240L, 5120L, 2560L}) {
        const int n_col_tiles = (int)(pitch / 512);
        const long rows = ((long)kMaxCus * kTiles / n_col_tiles + 2) * 256;        // every (workgroup, tile) writes a tile of its own
        unsigned char* out; CK(hipMalloc(&out, (size_t)rows * pitch)); CK(hipMemset(out, 0, (size_t)rows * pitch));
        for (int cus : {256, 128, 32}) {
            const int tiles = kTiles;
            auto run = [&](auto kern, const char* name) {
                CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                kern<<<cus, 512, 96 * 1024, st>>>(out, pitch, tiles, n_col_tiles, cyc);
                CK(hipEventRecord(e0, st));
                const int reps = 10;
                for (int i = 0; i < reps; i++) kern<<<cus, 512, 96 * 1024, st>>>(out, pitch, tiles, n_col_tiles, cyc);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
                std::vector<long long> h(cus); CK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
                double c = 0; for (long long x : h) c += (double)x; c /= cus;
                const double bytes = (double)cus * tiles * 256 * 512;
                printf("%-28s %6d %8ld | %10.1f %10.0f %12.1f\n", name, cus, pitch, ms * 1e3, bytes / ms / 1e6, (double)tiles * 256 * 512 / c);
            };
            run(store_tiles<4, false>, "16 rows x 64 B");
            run(store_tiles<8, false>, "8 rows x 128 B");
            run(store_tiles<32, false>, "2 rows x 512 B");
            run(store_tiles<4, true>, "16 rows x 64 B, nt");
            run(store_tiles<32, true>, "2 rows x 512 B, nt");
        }
        CK(hipFree(out));
    }
    return 0;
}
