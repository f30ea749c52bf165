// Decode GEMV micro-benchmark (dev tool): time dec_gemv_kernel for the large-v3 decode shapes over (S, NW) plans.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemv_bench.cpp speaksense_amd/csrc/kernels_decode.hip -Ispeaksense_amd/csrc -o tools/gemv_bench.bin
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
using namespace ss;
int main() {
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Shape { int N, K; const char* name; int epi; };
    Shape shapes[] = {{3840, 1280, "QKV", DEPI_PART}, {5120, 1280, "FC1", DEPI_GELU_T}, {1280, 5120, "FC2", DEPI_PART}, {1280, 1280, "O", DEPI_PART}, {51904, 1280, "logits", DEPI_LOGITS}};
    const int M = getenv("SS_GEMV_M") ? atoi(getenv("SS_GEMV_M")) : 8;     // token rows per launch (8 = batch8_strict, 32 = the headline's passes)
    // many distinct weight copies so every launch streams cold weights from HBM like a real step (32 layers)
    const int NCOPY = getenv("NCOPY") ? atoi(getenv("NCOPY")) : 24;
    for (auto& s : shapes) {
        f16* W; hipMalloc(&W, (size_t)NCOPY * s.N * s.K * 2); hipMemset(W, 0, (size_t)NCOPY * s.N * s.K * 2);
        f16* X; hipMalloc(&X, (size_t)128 * s.K * 2); hipMemset(X, 0, (size_t)128 * s.K * 2);
        float* part; hipMalloc(&part, (size_t)4 * 128 * s.N * 4);
        void* out; hipMalloc(&out, (size_t)128 * s.N * 4);
        float* bias; hipMalloc(&bias, s.N * 4); hipMemset(bias, 0, s.N * 4);
        for (int S = 1; S <= 4; S *= 2) for (int NW = 1; NW <= 4; NW *= 2) {
            if (s.K % S || (s.K / S) % NW) continue;
            const int kw = s.K / S / NW;
            if (kw % 32 || kw > 320) continue;
            if (s.epi != DEPI_PART && S != 1) continue;
            DecGemvDesc g{};
            g.pro = PRO_T; g.epi = s.epi; g.Xt = X; g.ldx = s.K; g.M = M; g.N = s.N; g.K = s.K; g.S = S; g.scale = 1.f; g.d = 1280;
            g.part_out = part; g.out = out; g.ldo = s.N; g.bias = bias; g.n_valid = s.N;
            auto run = [&](int i) { g.W = W + (size_t)(i % NCOPY) * s.N * s.K; launch_dec_gemv<f16>(g, NW, st); };
            for (int i = 0; i < NCOPY; i++) run(i);
            hipEventRecord(e0, st);
            const int reps = 96;
            for (int i = 0; i < reps; i++) run(i);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps, mb = (double)s.N * s.K * 2 / 1e6;
            printf("%-7s N=%5d K=%4d S=%d NW=%d kw=%3d blocks=%4d  %6.2f us  %6.2f TB/s\n", s.name, s.N, s.K, S, NW, kw, ((s.N + 15) / 16) * S, us, mb / us);
        }
        hipFree(W); hipFree(X); hipFree(part); hipFree(out); hipFree(bias);
    }
    return 0;
}
