// How much HBM bandwidth do S CONCURRENT decoder cross-attention launches get in total?  (dev tool, run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/diag/hbm_streams.cpp speaksense_amd/csrc/kernels_decode.hip -Ispeaksense_amd/csrc -o /tmp/hbm_streams
// Each stream owns a cross-KV region of 32 windows x (K, V) x 20 heads x 1500 x 64 f16 (246 MB: one large-v3 layer for a 32-row pass) and
// launches the unsplit kernel `reps` times back to back; wall time over all streams -> aggregate GB/s.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
using namespace ss;

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 200, max_s = argc > 3 ? atoi(argv[3]) : 4, L = argc > 4 ? atoi(argv[4]) : 1;   // L: layer regions a stream cycles through (32 = the engine's footprint)
    const int H = 20, d = 1280, Tn = 1500;
    const long per_win = 2L * H * Tn * 64;            // elements per window: K then V
    std::vector<hipStream_t> st(max_s);
    std::vector<f16*> kv(max_s), out(max_s);
    std::vector<float*> qp(max_s);
    float* qb; RowCtl* ctl;
    SS_HIP(hipMalloc(&qb, d * 4)); SS_HIP(hipMemset(qb, 0, d * 4));
    std::vector<RowCtl> hc(M);
    for (int m = 0; m < M; m++) { hc[m] = RowCtl{}; hc[m].cross = m; }
    SS_HIP(hipMalloc(&ctl, M * sizeof(RowCtl))); SS_HIP(hipMemcpy(ctl, hc.data(), M * sizeof(RowCtl), hipMemcpyHostToDevice));
    for (int s = 0; s < max_s; s++) {
        SS_HIP(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
        SS_HIP(hipMalloc(&kv[s], (size_t)L * M * per_win * 2)); SS_HIP(hipMemset(kv[s], 0x11, (size_t)L * M * per_win * 2));
        SS_HIP(hipMalloc(&out[s], (size_t)M * d * 2));
        SS_HIP(hipMalloc(&qp[s], (size_t)4 * kPartRows * d * 4)); SS_HIP(hipMemset(qp[s], 0, (size_t)4 * kPartRows * d * 4));
    }
    SS_HIP(hipDeviceSynchronize());
    const double bytes = (double)M * per_win * 2;
    for (int S = 1; S <= max_s; S++) {
        for (int round = 0; round < 2; round++) {      // round 0 warms up
            SS_HIP(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++)
                for (int s = 0; s < S; s++)
                {
                    const f16* base = kv[s] + (long)(i % L) * M * per_win;
                    launch_dec_cross_attention_direct<f16>(qp[s], 2, qb, 0.125f, base, base + (long)H * Tn * 64, per_win, d, H, Tn, ctl, M, out[s], st[s]);
                }
            SS_HIP(hipDeviceSynchronize());
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (round) printf("rows %d layers %d streams %d: %.1f us per launch-set, aggregate %.0f GB/s\n", M, L, S, sec / reps * 1e6, bytes * S * reps / sec / 1e9);
        }
    }
    return 0;
}
