// Dev tool ("measure the flag path first with a 2-kernel microbenchmark"): what does one DEPENDENT hand-off between two groups of
// workgroups cost on MI355X -- as a kernel boundary inside a captured hipGraph (what the decoder pass uses today) and as a flag hand-off inside one
// persistent kernel (what a fused per-block kernel would use)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/handoff_bench.cpp -o tools/handoff_bench.bin
// Two groups A and B of G workgroups (256 threads) ping-pong: every workgroup of a group writes a 4 KB payload; a workgroup of the other group may
// start its turn when (p2p) ITS partner's payload or (all) EVERY payload of the previous turn is visible, reads 4 KB of it, and writes its own.
// One turn = one hand-off; the decoder pass is a chain of ~350 of them (split-K GEMV -> reduce + LayerNorm -> GEMV -> attention -> ...), all of the
// "all" kind.  Forms: kernel boundary (graph of alternating launches); flags with HIP's agent-scope release / acquire atomics (the compiler adds
// buffer_wbl2 / buffer_inv); flags and payload with sc1 stores / loads only ("sc1 payload + drained flag", no cache maintenance instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_sc1_b128(f32x4* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1_b32(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_sc1_b32(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ f32x4 ld_sc1_b128(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- kernel-boundary form: one launch = one turn of one group ----
__global__ __launch_bounds__(256) void turn_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, int all, int G) {
    const int src = all ? (blockIdx.x * 7 + 3) % G : blockIdx.x;      // "all": some other workgroup's payload (any may be needed)
    f32x4 v = in[(size_t)src * 256 + threadIdx.x];
    v += (f32x4){1.f, 1.f, 1.f, 1.f};
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

// ---- persistent form ----
// MODE 0: HIP agent-scope atomics (release store of the flag after the payload; acquire loads).  MODE 1: sc1 payload, drained, sc1 flag; sc1 polls and reads.
template <int MODE>
__global__ __launch_bounds__(256) void pingpong(f32x4* pay_a, f32x4* pay_b, unsigned* flag_a, unsigned* flag_b, int G, int turns, int all, unsigned* fail) {
    const bool is_a = blockIdx.x < (unsigned)G;
    const int me = is_a ? blockIdx.x : blockIdx.x - G;
    f32x4* my_pay = (is_a ? pay_a : pay_b) + (size_t)me * 256;
    const f32x4* their_pay = is_a ? pay_b : pay_a;
    unsigned* my_flag = (is_a ? flag_a : flag_b) + me * 16;           // one flag per 64-byte line
    const unsigned* their_flag = is_a ? flag_b : flag_a;
    const int src = all ? (me * 7 + 3) % G : me;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int t = 1; t <= turns; t++) {
        // A moves on odd turns, B on even turns; before moving (except A's first move) wait for the other group's previous turn
        const bool my_move = is_a ? (t & 1) : !(t & 1);
        if (!my_move) continue;
        if (t > 1) {
            const unsigned want = (unsigned)(t - 1);
            unsigned spins = 0;
            while (true) {
                bool ok = true;
                if (all) {
                    for (int i = threadIdx.x; i < G; i += 256) {
                        const unsigned f = MODE == 0 ? __hip_atomic_load(their_flag + i * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : ld_sc1_b32(their_flag + i * 16);
                        ok = ok && f >= want;
                    }
                } else if (threadIdx.x == 0) {
                    const unsigned f = MODE == 0 ? __hip_atomic_load(their_flag + src * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : ld_sc1_b32(their_flag + src * 16);
                    ok = f >= want;
                }
                if (__syncthreads_and(ok)) break;
                if (++spins > (1u << 20)) { if (threadIdx.x == 0) *fail = 1; return; }
            }
            if (MODE == 0) v = their_pay[(size_t)src * 256 + threadIdx.x];     // ordered behind the acquire loads above
            else v = ld_sc1_b128(their_pay + (size_t)src * 256 + threadIdx.x);
        }
        v += (f32x4){1.f, 1.f, 1.f, 1.f};
        if (MODE == 0) {
            my_pay[threadIdx.x] = v;
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(my_flag, (unsigned)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            st_sc1_b128(my_pay + threadIdx.x, v);
            drain();
            __syncthreads();
            if (threadIdx.x == 0) st_sc1_b32(my_flag, (unsigned)t);
        }
    }
    if (v[0] == -1.f) *fail = 2;
}

int main(int argc, char** argv) {
    const int turns = argc > 1 ? atoi(argv[1]) : 2000;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f32x4 *pa, *pb; unsigned *fa, *fb, *fail;
    const int GMAX = 256;
    CK(hipMalloc(&pa, (size_t)GMAX * 256 * 16)); CK(hipMalloc(&pb, (size_t)GMAX * 256 * 16));
    CK(hipMalloc(&fa, GMAX * 64)); CK(hipMalloc(&fb, GMAX * 64)); CK(hipMalloc(&fail, 4));
    printf("one dependent hand-off between two groups of G workgroups x 256 threads, 4 KB payload per workgroup (us per hand-off; %d turns)\n", turns);
    printf("%5s %5s | %14s | %22s | %22s\n", "G", "dep", "kernel boundary", "flags, HIP atomics", "flags, sc1 only");
    for (int G : {8, 32, 80, 256}) {
        for (int all = 0; all <= 1; all++) {
            float ms_k = 0, ms_f[2] = {0, 0};
            {   // graph of alternating launches
                CK(hipMemset(pa, 0, (size_t)GMAX * 256 * 16)); CK(hipMemset(pb, 0, (size_t)GMAX * 256 * 16));
                hipGraph_t g; hipGraphExec_t ge;
                const int per_graph = 200;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                for (int t = 0; t < per_graph; t++) {
                    if (t & 1) turn_kernel<<<G, 256, 0, st>>>(pa, pb, all, G);
                    else turn_kernel<<<G, 256, 0, st>>>(pb, pa, all, G);
                }
                CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
                CK(hipEventRecord(e0, st));
                const int reps = turns / per_graph;
                for (int r = 0; r < reps; r++) CK(hipGraphLaunch(ge, st));
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms_k, e0, e1)); ms_k /= (reps * per_graph);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            for (int mode = 0; mode < 2; mode++) {
                unsigned hf = 0;
                for (int rep = 0; rep < 2; rep++) {     // rep 0 warms up
                    CK(hipMemsetAsync(fa, 0, GMAX * 64, st)); CK(hipMemsetAsync(fb, 0, GMAX * 64, st)); CK(hipMemsetAsync(fail, 0, 4, st));
                    CK(hipEventRecord(e0, st));
                    if (mode == 0) pingpong<0><<<2 * G, 256, 0, st>>>(pa, pb, fa, fb, G, turns, all, fail);
                    else pingpong<1><<<2 * G, 256, 0, st>>>(pa, pb, fa, fb, G, turns, all, fail);
                    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms_f[mode], e0, e1));
                    CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
                    if (hf) break;
                }
                ms_f[mode] = hf ? -1.f : ms_f[mode] / turns;
            }
            // correctness of the sc1 form: after `turns` turns every payload element of the last mover must equal its chain length
            printf("%5d %5s | %11.2f    | %19.2f    | %19.2f\n", G, all ? "all" : "p2p", ms_k * 1e3, ms_f[0] * 1e3, ms_f[1] * 1e3);
        }
    }
    // value check of the sc1 form (p2p, G = 32): element = number of turns
    {
        const int G = 32, T = 101;
        CK(hipMemset(pa, 0, (size_t)GMAX * 256 * 16)); CK(hipMemset(pb, 0, (size_t)GMAX * 256 * 16));
        CK(hipMemset(fa, 0, GMAX * 64)); CK(hipMemset(fb, 0, GMAX * 64)); CK(hipMemset(fail, 0, 4));
        pingpong<1><<<2 * G, 256, 0, st>>>(pa, pb, fa, fb, G, T, 1, fail);
        CK(hipStreamSynchronize(st));
        std::vector<float> h((size_t)G * 256 * 4);
        CK(hipMemcpy(h.data(), pa, h.size() * 4, hipMemcpyDeviceToHost));
        bool ok = true; for (float x : h) ok = ok && x == (float)T;
        printf("sc1 form, all-dependency, G = 32: payload after %d turns %s\n", T, ok ? "correct" : "WRONG (a stale line was read)");
    }
    return 0;
}
