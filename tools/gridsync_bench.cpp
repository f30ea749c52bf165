// Dev tool: cost of a software grid barrier vs a kernel boundary on MI355X, and of barrier-separated weight-streaming phases.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gridsync_bench.cpp -o tools/gridsync_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { *fail = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

// flag-array barrier: WG i publishes flags[i] = epoch; every thread polls one flag (G <= 256 * k)
__device__ __forceinline__ bool flag_barrier(unsigned* flags, unsigned epoch, unsigned* fail) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    while (true) {
        bool ok = true;
        for (unsigned i = threadIdx.x; i < gridDim.x; i += blockDim.x)
            ok = ok && (__hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
        if (__syncthreads_and(ok)) break;
        if (++spins > (1u << 22)) { *fail = 2; return false; }
    }
    return true;
}
__global__ __launch_bounds__(256) void flag_barrier_only(unsigned* flags, int rounds, unsigned* fail) {
    for (int r = 0; r < rounds; r++)
        if (!flag_barrier(flags, (unsigned)(r + 1), fail)) return;
}
// two-level: 8 group counters (blockIdx % 8 = XCD) + polling of the 8 group counters
__device__ __forceinline__ bool xcd_barrier(unsigned* ctr8, unsigned epoch, unsigned* fail) {
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr8 + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned per = gridDim.x / 8;
    unsigned spins = 0;
    while (true) {
        bool ok = true;
        if (threadIdx.x < 8) ok = __hip_atomic_load(ctr8 + threadIdx.x * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= epoch * per;
        if (__syncthreads_and(ok)) break;
        if (++spins > (1u << 22)) { *fail = 3; return false; }
    }
    return true;
}
__global__ __launch_bounds__(256) void xcd_barrier_only(unsigned* ctr8, int rounds, unsigned* fail) {
    for (int r = 0; r < rounds; r++)
        if (!xcd_barrier(ctr8, (unsigned)(r + 1), fail)) return;
}
__global__ __launch_bounds__(256) void barrier_only(unsigned* ctr, int rounds, unsigned* fail) {
    for (int r = 0; r < rounds; r++)
        if (!grid_barrier(ctr, (unsigned)(r + 1) * gridDim.x, fail)) return;
}
// each phase: every WG streams `bytes_per_wg` of weights (float4 loads, summed), writes one value, then barrier; next phase reads a neighbour's value
__global__ __launch_bounds__(256) void phases(unsigned* ctr, int rounds, unsigned* fail, const float4* __restrict__ w, size_t f4_per_wg, size_t f4_total,
                                              float* xchg, int prefetch) {
    float acc = 0.f;
    size_t base = (size_t)blockIdx.x * f4_per_wg;
    for (int r = 0; r < rounds; r++) {
        const float4* p = w + (base + (size_t)r * gridDim.x * f4_per_wg) % (f4_total - f4_per_wg);
        float4 s = make_float4(0, 0, 0, 0);
        for (size_t i = threadIdx.x; i < f4_per_wg; i += 256) { float4 v = p[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
        acc += s.x + s.y + s.z + s.w + (r ? xchg[(blockIdx.x + 1) % gridDim.x] : 0.f);
        if (threadIdx.x == 0) xchg[blockIdx.x] = acc;
        __threadfence();
        if (!grid_barrier(ctr, (unsigned)(r + 1) * gridDim.x, fail)) return;
    }
    if (acc == 123.456f) xchg[0] = 0;
}
__global__ __launch_bounds__(256) void stream_only(const float4* __restrict__ w, size_t f4_per_wg, size_t off, size_t f4_total, float* xchg) {
    const float4* p = w + ((size_t)blockIdx.x * f4_per_wg + off) % (f4_total - f4_per_wg);
    float4 s = make_float4(0, 0, 0, 0);
    for (size_t i = threadIdx.x; i < f4_per_wg; i += 256) { float4 v = p[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (threadIdx.x == 0) xchg[blockIdx.x] = s.x + s.y + s.z + s.w;
}
__global__ void empty_kernel(float* x) { if (x == nullptr && threadIdx.x == 9999) x[0] = 0; }

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned *ctr, *fail; CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&fail, 4));
    float* xchg; CK(hipMalloc(&xchg, 4096 * 4));
    const size_t wbytes = (size_t)2 << 30;   // 2 GB of "weights": phases never hit L2/MALL
    float4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 0, wbytes));
    const size_t f4_total = wbytes / 16;
    float ms;
    for (int G : {256, 512}) {
        const int R = 2000;
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(fail, 0, 4));
        barrier_only<<<G, 256, 0, st>>>(ctr, 10, fail); CK(hipMemset(ctr, 0, 4));
        CK(hipEventRecord(e0, st));
        barrier_only<<<G, 256, 0, st>>>(ctr, R, fail);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("grid barrier G=%d: %.2f us per barrier (fail=%u)\n", G, ms * 1e3 / R, f);
    }
    unsigned* flags; CK(hipMalloc(&flags, 4096 * 4));
    for (int G : {256, 512}) {
        const int R = 2000;
        CK(hipMemset(flags, 0, 4096 * 4)); CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0, st));
        flag_barrier_only<<<G, 256, 0, st>>>(flags, R, fail);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("flag-array barrier G=%d: %.2f us per barrier (fail=%u)\n", G, ms * 1e3 / R, f);
        CK(hipMemset(flags, 0, 4096 * 4)); CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0, st));
        xcd_barrier_only<<<G, 256, 0, st>>>(flags, R, fail);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("8-counter barrier G=%d: %.2f us per barrier (fail=%u)\n", G, ms * 1e3 / R, f);
    }
    {
        const int R = 2000;
        for (int i = 0; i < 10; i++) empty_kernel<<<256, 256, 0, st>>>(xchg);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < R; i++) empty_kernel<<<256, 256, 0, st>>>(xchg);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel chain: %.2f us per launch\n", ms * 1e3 / R);
    }
    for (size_t kb : {16, 48, 128, 256}) {
        const int G = 256, R = 400;
        const size_t f4 = kb * 1024 / 16;
        CK(hipMemset(ctr, 0, 4)); CK(hipMemset(fail, 0, 4));
        CK(hipEventRecord(e0, st));
        phases<<<G, 256, 0, st>>>(ctr, R, fail, w, f4, f4_total, xchg, 0);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / R;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < R; i++) stream_only<<<G, 256, 0, st>>>(w, f4, (size_t)i * G * f4, f4_total, xchg);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        const double us2 = ms2 * 1e3 / R;
        unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
        printf("phase %3zu KB/WG (%.1f MB): persistent+barrier %.2f us (%.2f TB/s) | separate kernels %.2f us (%.2f TB/s) fail=%u\n", kb, kb * G / 1024.0, us,
               kb * 1024.0 * G / us / 1e6, us2, kb * 1024.0 * G / us2 / 1e6, f);
    }
    return 0;
}
