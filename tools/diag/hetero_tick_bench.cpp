// Premise check for a "heterogeneous tick" decoder pass (dev tool, run on the GPU box; DESIGN.md section 8a, "what next"):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/diag/hetero_tick_bench.cpp -o tools/diag/hetero_tick_bench.bin
// A decoder pass is a chain of ~10 latency-bound launches per layer plus one HBM-streaming cross-attention launch; lanes overlap the two kinds
// across independent row groups, but each kernel doubles in duration under the other lanes' traffic.  The alternative this tool
// prices: ONE launch per chain step that carries the chain kernel of row group B AND a tenth of the cross-attention key range of row group A (whose
// chain is one layer-half behind), so that the stream is spread over the whole layer and hides behind the chain instead of competing with it.
// Stand-ins with the memory / MFMA pattern of the product kernels (not their arithmetic): a 16-row weight tile x 32 token rows GEMV with every load
// issued before the first MFMA (dec_gemv_kernel<T,EPI,2,10>), and the cross-attention's K then V sweep over a key range (dec_cross_attn_q_kernel<T,1>).
// Reported per layer (10 chain steps + the cross-attention of 32 rows x 20 heads x 1500 keys), large-v3 sizes, one stream, graph replay:
//   serial      10 chain launches, then the whole cross-attention as one launch           (today's pass, one lane alone)
//   two lanes   two streams, each running `serial` on its own rows                        (today's overlap mechanism)
//   fused       10 launches, each = chain step of group B + keys [150 s, 150 s + 150) of group A   (the proposal; per layer it does ONE group's worth of work:
//               B's chain and A's stream, exactly what `serial` does for one group -- the question is whether the stream hides behind the chain)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 1280, H = 20, TN = 1500, M = 32, NFR = 10, NW = 4;   // one GEMV workgroup: 16 weight rows x K = 1280, 4 waves x 10 fragments

// ---- chain stand-in: N = n_tiles x 16 outputs for 32 token rows ---------------------------------------------------------------------------------------
__device__ __forceinline__ void gemv_part(int tile, const f16* __restrict__ W, const f16* __restrict__ X, float* __restrict__ out, float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fg = lane >> 4;
    const int kbeg = wave * (32 * NFR);
    const f16* wp = W + (long)(tile * 16 + frow) * D + kbeg;
    f16x8 wf[NFR], xf[2][NFR];
#pragma unroll
    for (int f = 0; f < NFR; f++) wf[f] = *(const f16x8*)(wp + (f >> 1) * 64 + fg * 16 + (f & 1) * 8);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int f = 0; f < NFR; f++) xf[ct][f] = *(const f16x8*)(X + (long)(ct * 16 + frow) * D + kbeg + (f >> 1) * 64 + fg * 16 + (f & 1) * 8);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int f = 0; f < NFR; f++) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f], xf[ct][f], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((wave * 2 + ct) * 16 + frow) * 17 + fg * 4 + r] = acc[ct][r];
    __syncthreads();
    for (int idx = tid; idx < 512; idx += 256) {
        const int m = idx >> 4, nn = idx & 15;
        float v = 0.f;
        for (int w = 0; w < NW; w++) v += red[((w * 2 + (m >> 4)) * 16 + (m & 15)) * 17 + nn];
        out[(long)m * 5120 + tile * 16 + nn] = v;
    }
}

// ---- stream stand-in: keys [k0, k1) of one (row, head): scores, then P.V, partial (max, sum, o[64]) out -----------------------------------------------
__device__ __forceinline__ void cross_part(int pair, int k0, int k1, const f16* __restrict__ KV, const f16* __restrict__ Q, float* __restrict__ part, float* s_sc) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane >> 3, c = lane & 7;
    const int m = pair / H, h = pair % H, nk = k1 - k0;
    const f16* K = KV + ((long)m * 2 * H + h) * TN * 64;
    const f16* V = K + (long)H * TN * 64;
    float qv[8];
    { const f16x8 t = *(const f16x8*)(Q + (long)m * D + h * 64 + c * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) qv[e] = (float)t[e]; }
    const int nit = (nk + 31) / 32;
    float mx = -1e30f;
    for (int it = 0; it < nit; it += 4) {
        f16x8 kv[4]; int ii[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { ii[u] = (it + u) * 32 + wave * 8 + r; kv[u] = *(const f16x8*)(K + (long)(k0 + (ii[u] < nk ? ii[u] : 0)) * 64 + c * 8); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) a += qv[e] * (float)kv[u][e];
            a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
            if (ii[u] < nk) { if (c == 0) s_sc[ii[u]] = a; mx = fmaxf(mx, a); }
        }
    }
    __syncthreads();
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < nit; it += 4) {
        f16x8 vv[4]; float pw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = (it + u) * 32 + wave * 8 + r; const bool ok = i < nk; vv[u] = *(const f16x8*)(V + (long)(k0 + (ok ? i : 0)) * 64 + c * 8); pw[u] = ok ? __expf(s_sc[i] - mx) : 0.f; }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int e = 0; e < 8; e++) acc[e] += pw[u] * (float)vv[u][e];
    }
    if (r == 0) {
#pragma unroll
        for (int e = 0; e < 8; e++) part[((long)pair * 16 + (k0 / 128) % 16) * 264 + wave * 66 + c * 8 + e] = acc[e] + mx;
    }
}

// the same with the weights pre-packed fragment-major: fragment (tile, k / 32) is ONE contiguous kilobyte in lane order, so a wave's load instruction
// touches 8 full 128-byte lines instead of 64 bytes in each of 16 lines (rows 2560 bytes apart)
__global__ __launch_bounds__(256) void k_gemv_packed(const f16* __restrict__ W, const f16* __restrict__ X, float* __restrict__ out) {
    __shared__ float red[NW * 2 * 16 * 17];
    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fg = lane >> 4;
    const int kbeg = wave * (32 * NFR);
    const f16* wp = W + ((long)tile * (D / 32) + kbeg / 32) * 512 + lane * 8;
    f16x8 wf[NFR], xf[2][NFR];
#pragma unroll
    for (int f = 0; f < NFR; f++) wf[f] = *(const f16x8*)(wp + f * 512);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int f = 0; f < NFR; f++) xf[ct][f] = *(const f16x8*)(X + (long)(ct * 16 + frow) * D + kbeg + f * 32 + fg * 8);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int f = 0; f < NFR; f++) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f], xf[ct][f], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((wave * 2 + ct) * 16 + frow) * 17 + fg * 4 + r] = acc[ct][r];
    __syncthreads();
    for (int idx = tid; idx < 512; idx += 256) {
        const int m = idx >> 4, nn = idx & 15;
        float v = 0.f;
        for (int w = 0; w < NW; w++) v += red[((w * 2 + (m >> 4)) * 16 + (m & 15)) * 17 + nn];
        out[(long)m * 5120 + tile * 16 + nn] = v;
    }
}
// ... and with the 32 activation rows stored fragment-major as well (what a producer kernel would have to write)
__global__ __launch_bounds__(256) void k_gemv_packed2(const f16* __restrict__ W, const f16* __restrict__ X, float* __restrict__ out) {
    __shared__ float red[NW * 2 * 16 * 17];
    const int tile = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frow = lane & 15, fg = lane >> 4;
    const int kbeg = wave * (32 * NFR);
    const f16* wp = W + ((long)tile * (D / 32) + kbeg / 32) * 512 + lane * 8;
    f16x8 wf[NFR], xf[2][NFR];
#pragma unroll
    for (int f = 0; f < NFR; f++) wf[f] = *(const f16x8*)(wp + f * 512);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int f = 0; f < NFR; f++) xf[ct][f] = *(const f16x8*)(X + (((long)ct * (D / 32) + kbeg / 32 + f) * 64 + lane) * 8);   // activations fragment-major too
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int f = 0; f < NFR; f++) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[f], xf[ct][f], acc[ct], 0, 0, 0);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((wave * 2 + ct) * 16 + frow) * 17 + fg * 4 + r] = acc[ct][r];
    __syncthreads();
    for (int idx = tid; idx < 512; idx += 256) {
        const int m = idx >> 4, nn = idx & 15;
        float v = 0.f;
        for (int w = 0; w < NW; w++) v += red[((w * 2 + (m >> 4)) * 16 + (m & 15)) * 17 + nn];
        out[(long)m * 5120 + tile * 16 + nn] = v;
    }
}
__global__ __launch_bounds__(256) void k_gemv(const f16* W, const f16* X, float* out) {
    __shared__ float red[NW * 2 * 16 * 17];
    gemv_part(blockIdx.x, W, X, out, red);
}
__global__ __launch_bounds__(256) void k_cross(const f16* KV, const f16* Q, float* part, int k0, int k1) {
    __shared__ float s_sc[1536];
    cross_part(blockIdx.x, k0, k1, KV, Q, part, s_sc);
}
__global__ __launch_bounds__(256) void k_fused(int n_tiles, const f16* W, const f16* X, float* out, const f16* KV, const f16* Q, float* part, int k0, int k1) {
    __shared__ float sm[2304];
    // interleave the two kinds over the block index so that both spread over all XCDs / CUs
    const int b = blockIdx.x, n_pairs = M * H, period = (n_tiles + n_pairs);
    // block b is a chain block iff it falls on the first n_tiles slots of an evenly interleaved sequence
    const long slot = (long)b * n_tiles / period, next = (long)(b + 1) * n_tiles / period;
    if (next > slot) gemv_part((int)slot, W, X, out, sm);
    else cross_part(b - (int)next, k0, k1, KV, Q, part, sm);
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 20;
    // chain steps of one layer: output widths in 16-row tiles (QKV 240, out 80 x 4 split = 320 workgroups modelled as 320 tiles of K/4 ... kept simple: K = 1280 everywhere)
    const int steps[10] = {240, 80, 80, 80, 80, 80, 320, 320, 80, 80};   // QKV, self-attn stand-in, out-proj, reduce, cross-q, reduce, FC1, FC2 (as 320), cross-out, reduce
    f16 *W[2], *X[2], *KV[2], *Q[2]; float *out[2], *part[2];
    hipStream_t st[2];
    const size_t wbytes = (size_t)5120 * D * 2 * 12, kvbytes = (size_t)M * 2 * H * TN * 64 * 2;
    for (int g = 0; g < 2; g++) {
        CK(hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking));
        CK(hipMalloc(&W[g], wbytes * L)); CK(hipMemset(W[g], 0x11, wbytes * L));                 // a pass's worth of distinct weights: every step starts cold, as in the engine
        CK(hipMalloc(&X[g], (size_t)M * D * 2)); CK(hipMemset(X[g], 0x11, (size_t)M * D * 2));
        CK(hipMalloc(&KV[g], kvbytes * L)); CK(hipMemset(KV[g], 0x11, kvbytes * L));
        CK(hipMalloc(&Q[g], (size_t)M * D * 2)); CK(hipMemset(Q[g], 0x11, (size_t)M * D * 2));
        CK(hipMalloc(&out[g], (size_t)M * 5120 * 4)); CK(hipMalloc(&part[g], (size_t)M * H * 16 * 264 * 4));
    }
    CK(hipDeviceSynchronize());
    auto wptr = [&](int g, int l, int s) { return W[g] + ((size_t)l * 12 + s) * 5120 * D; };
    auto kvptr = [&](int g, int l) { return KV[g] + (size_t)l * (kvbytes / 2); };
    auto serial_pass = [&](int g) {
        for (int l = 0; l < L; l++) {
            for (int s = 0; s < 5; s++) k_gemv<<<steps[s], 256, 0, st[g]>>>(wptr(g, l, s), X[g], out[g]);
            k_cross<<<M * H, 256, 0, st[g]>>>(kvptr(g, l), Q[g], part[g], 0, TN);
            for (int s = 5; s < 10; s++) k_gemv<<<steps[s], 256, 0, st[g]>>>(wptr(g, l, s), X[g], out[g]);
        }
    };
    auto fused_pass = [&]() {   // group 0's chain with group 1's stream riding on it: the same amount of work as serial_pass, differently packed
        for (int l = 0; l < L; l++)
            for (int s = 0; s < 10; s++)
                k_fused<<<steps[s] + M * H, 256, 0, st[0]>>>(steps[s], wptr(0, l, s), X[0], out[0], kvptr(1, l), Q[1], part[1], 150 * s, 150 * s + 150);
    };
    auto capture = [&](int g, auto fn) {
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st[g], hipStreamCaptureModeThreadLocal)); fn(); CK(hipStreamEndCapture(st[g], &gr));
        CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        return ex;
    };
    hipGraphExec_t g_serial0 = capture(0, [&] { serial_pass(0); }), g_serial1 = capture(1, [&] { serial_pass(1); }), g_fused = capture(0, fused_pass);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, int n_groups, auto body) {
        body(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < reps; i++) body();
        CK(hipStreamSynchronize(st[1]));
        CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-58s %8.3f ms per pass of %d layers = %6.1f us per layer, %5.1f rows / ms\n", name, ms / reps, L, 1e3 * ms / reps / L, n_groups * M / (ms / reps));
    };
    time_it("serial (one group, one stream)", 1, [&] { CK(hipGraphLaunch(g_serial0, st[0])); });
    time_it("two lanes (two groups, two streams)", 2, [&] { CK(hipGraphLaunch(g_serial0, st[0])); CK(hipGraphLaunch(g_serial1, st[1])); CK(hipStreamSynchronize(st[1])); });
    time_it("fused ticks (one group's chain + another's stream per launch)", 1, [&] { CK(hipGraphLaunch(g_fused, st[0])); });
    // ---- which pairing costs what?  chain-only and stream-only passes, alone and against each other (r04_ab) ----
    auto chain_pass = [&](int g) { for (int l = 0; l < L; l++) for (int s = 0; s < 10; s++) k_gemv<<<steps[s], 256, 0, st[g]>>>(wptr(g, l, s), X[g], out[g]); };
    auto cross_pass = [&](int g) { for (int l = 0; l < L; l++) k_cross<<<M * H, 256, 0, st[g]>>>(kvptr(g, l), Q[g], part[g], 0, TN); };
    hipGraphExec_t g_chain[2] = {capture(0, [&] { chain_pass(0); }), capture(1, [&] { chain_pass(1); })};
    hipGraphExec_t g_cross[2] = {capture(0, [&] { cross_pass(0); }), capture(1, [&] { cross_pass(1); })};
    hipEvent_t a0, a1, b0, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b0)); CK(hipEventCreate(&b1));
    auto pair = [&](const char* name, hipGraphExec_t ga, int ra, hipGraphExec_t gb, int rb) {   // ga x ra on stream 0 while gb x rb runs on stream 1 (rb = 0: alone)
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a0, st[0])); if (rb) CK(hipEventRecord(b0, st[1]));
        for (int i = 0; i < std::max(ra, rb); i++) { if (i < ra) CK(hipGraphLaunch(ga, st[0])); if (i < rb) CK(hipGraphLaunch(gb, st[1])); }
        CK(hipEventRecord(a1, st[0])); if (rb) CK(hipEventRecord(b1, st[1]));
        CK(hipDeviceSynchronize());
        float ma = 0, mb = 0; CK(hipEventElapsedTime(&ma, a0, a1)); if (rb) CK(hipEventElapsedTime(&mb, b0, b1));
        printf("%-46s stream 0: %7.3f ms per pass (%5.1f us per layer)", name, ma / ra, 1e3 * ma / ra / L);
        if (rb) printf("   stream 1: %7.3f ms per pass (%5.1f us per layer)", mb / rb, 1e3 * mb / rb / L);
        printf("\n");
    };
    auto chain_packed = [&](int g) { for (int l = 0; l < L; l++) for (int s = 0; s < 10; s++) k_gemv_packed<<<steps[s], 256, 0, st[g]>>>(wptr(g, l, s), X[g], out[g]); };
    hipGraphExec_t g_chainp[2] = {capture(0, [&] { chain_packed(0); }), capture(1, [&] { chain_packed(1); })};
    pair("chain alone", g_chain[0], reps, nullptr, 0);
    pair("chain alone, fragment-major weights", g_chainp[0], reps, nullptr, 0);
    pair("chain | chain, fragment-major weights", g_chainp[0], reps, g_chainp[1], reps);
    auto chain_packed2 = [&](int g) { for (int l = 0; l < L; l++) for (int s = 0; s < 10; s++) k_gemv_packed2<<<steps[s], 256, 0, st[g]>>>(wptr(g, l, s), X[g], out[g]); };
    hipGraphExec_t g_chainq[2] = {capture(0, [&] { chain_packed2(0); }), capture(1, [&] { chain_packed2(1); })};
    pair("chain alone, weights and activations fragment-major", g_chainq[0], reps, nullptr, 0);
    pair("chain | chain, weights and activations fragment-major", g_chainq[0], reps, g_chainq[1], reps);
    pair("chain (fragment-major) | stream", g_chainp[0], reps, g_cross[1], 2 * reps);
    pair("stream (cross-attention) alone", g_cross[0], reps, nullptr, 0);
    pair("chain | chain", g_chain[0], reps, g_chain[1], reps);
    pair("stream | stream", g_cross[0], reps, g_cross[1], reps);
    // equal wall time on both sides: a chain pass is ~2 x a stream pass
    pair("chain | stream (stream 1 runs 2 passes per chain pass)", g_chain[0], reps, g_cross[1], 2 * reps);
    pair("full | full", g_serial0, reps, g_serial1, reps);
    // ---- does limiting how many stream workgroups are resident per CU protect the other lane's chain?  (dynamic LDS as the limiter) ----
    for (int dyn : {0, 40 << 10, 72 << 10, 120 << 10}) {
        auto cross_lim = [&](int g) { for (int l = 0; l < L; l++) k_cross<<<M * H, 256, dyn, st[g]>>>(kvptr(g, l), Q[g], part[g], 0, TN); };
        CK(hipFuncSetAttribute((const void*)k_cross, hipFuncAttributeMaxDynamicSharedMemorySize, 120 << 10));
        hipGraphExec_t gl[2] = {capture(0, [&] { cross_lim(0); }), capture(1, [&] { cross_lim(1); })};
        char nm[96];
        snprintf(nm, sizeof nm, "stream alone, %3d KB dynamic LDS", dyn >> 10); pair(nm, gl[0], reps, nullptr, 0);
        snprintf(nm, sizeof nm, "chain | stream, %3d KB dynamic LDS", dyn >> 10); pair(nm, g_chain[0], reps, gl[1], 2 * reps);
    }
    // r04_r on MI355X: serial 3.475 ms (108.6 us per layer, 9.2 rows / ms -- the engine's real one-lane pass is 3.42 - 3.46 ms), two lanes 5.063 ms for two
    // groups (12.6 rows / ms), fused 3.897 ms for ONE group's worth (121.8 us per layer, 8.2 rows / ms): slower than serial.  A tenth of the key range
    // is 19 KB of K and 19 KB of V per (row, head) workgroup -- four dependent memory round trips and a barrier, i.e. as latency-bound as the chain
    // step it rides on -- so every fused launch lasts chain + slice instead of max(chain, slice).  The stream is only cheap as ONE long launch.
    return 0;
}
