// Standalone GEMM micro-benchmark / correctness harness for kernels_gemm.hip (dev tool, run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gemm_bench.cpp speaksense_amd/csrc/kernels_gemm.hip -Ispeaksense_amd/csrc -o /tmp/gemm_bench
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
using namespace ss;
namespace ss { extern int g_gemm_cu_cap; }

__global__ void ref_gemm(const f16* A, const f16* W, const float* bias, float* C, int M, int N, int K) {
    int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float acc = 0;
    for (int k = 0; k < K; k++) acc += (float)A[(long)m * K + k] * (float)W[(long)n * K + k];
    C[(long)m * N + n] = acc + bias[n];
}
__global__ void fill(f16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = (f16)(((x & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
int main(int argc, char** argv) {
    struct Shape { int M, N, K; const char* name; };
    Shape shapes[] = {{48000, 5120, 1280, "FC1x4"}, {48000, 1280, 5120, "FC2x4"}, {48000, 1280, 1280, "Ox4"}, {48000, 2560, 1280, "QKx4"}, {12000, 5120, 1280, "FC1"}, {12000, 1280, 5120, "FC2"}, {12000, 3840, 1280, "QKV"}, {12000, 4096, 1280, "N4096"}, {12000, 2560, 1280, "QK"}, {11776, 5120, 1280, "M46"}, {12288, 5120, 1280, "M48"}, {12000, 1280, 1280, "O"},
                      {12000, 81920, 1280, "crossKV"}, {4096, 4096, 4096, "sq4096"}, {8192, 8192, 8192, "sq8192"}};
    if (getenv("SS_GEMM_TINY")) { shapes[0] = {1024, 256, 128, "tiny2"}; shapes[1] = {1024, 256, 256, "tiny4"}; shapes[2] = {1024, 512, 1280, "tiny20"}; }
    const int n_shapes = getenv("SS_GEMM_TINY") ? 3 : (int)(sizeof(shapes) / sizeof(shapes[0]));
    int kinds[] = {EPI_STORE_T, EPI_GELU_T, EPI_RES_F32};
    const char* kn[] = {"store", "gelu", "res_f32"};
    if (getenv("SS_GEMM_CUS")) g_gemm_cu_cap = atoi(getenv("SS_GEMM_CUS"));   // how do the per-tile phases change when fewer CUs run tiles at once?
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int si = 0; si < n_shapes; si++) {
        auto& s = shapes[si];
        f16 *A, *W; float *bias, *Cf; void* out;
        hipMalloc(&A, (size_t)s.M * s.K * 2); hipMalloc(&W, (size_t)s.N * s.K * 2); hipMalloc(&bias, s.N * 4);
        hipMalloc(&out, (size_t)s.M * s.N * 4);
        // SS_GEMM_ZERO=1: all-zero operands -- the same instruction stream toggles far fewer bits, the chip's power management lets it clock higher
        // (MI355X_MICROARCH.md "DVFS give-back"): the A/B that separates what the schedule loses from what the power budget takes
        const float zs = getenv("SS_GEMM_ZERO") ? 0.0f : 1.0f;
        fill<<<1024, 256>>>(A, (size_t)s.M * s.K, 1, 1.0f * zs); fill<<<1024, 256>>>(W, (size_t)s.N * s.K, 2, 0.05f * zs);
        hipMemset(bias, 0, s.N * 4); hipMemset(out, 0, (size_t)s.M * s.N * 4);
        for (int ki = 0; ki < 3; ki++) {
            GemmDesc g{};
            g.A = A; g.lda = s.K; g.a_rows_per_batch = 0; g.W = W; g.M = s.M; g.N = s.N; g.K = s.K; g.kind = kinds[ki]; g.bias = bias;
            g.out = out; g.ldo = s.N; g.o_rows_per_batch = 0; g.res = (float*)out; g.scale = 1.0f; g.rows_per_batch = 1500; g.gelu_f16_in = 1;   // as the f16 engine runs it
            launch_gemm<f16>(g, st);
            hipEventRecord(e0, st);
            const int reps = getenv("SS_GEMM_REPS") ? atoi(getenv("SS_GEMM_REPS")) : 10;   // more repetitions = longer sustained load (the chip's power management reacts within milliseconds)
            for (int i = 0; i < reps; i++) launch_gemm<f16>(g, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("%-8s M=%5d N=%5d K=%4d %-8s %8.3f ms  %7.1f TF/s\n", s.name, s.M, s.N, s.K, kn[ki], ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        }
        if (s.N == 1280 && s.K == 1280 && s.M % 1500 == 0) {   // the V projection of the encoder: V^T layout [b][h][64][Tpad] (EPI_VT), what enc_attn_lds_kernel reads
            GemmDesc g{};
            g.A = A; g.lda = s.K; g.W = W; g.M = s.M; g.N = s.N; g.K = s.K; g.kind = EPI_VT; g.bias = bias; g.out = out; g.ldo = s.N; g.scale = 1.0f;
            g.rows_per_batch = 1500; g.d = 1280; g.Tpad = 1536; g.n_batch = s.M / 1500;
            launch_gemm<f16>(g, st);
            hipEventRecord(e0, st);
            for (int i = 0; i < 10; i++) launch_gemm<f16>(g, st);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
            printf("%-8s M=%5d N=%5d K=%4d %-8s %8.3f ms  %7.1f TF/s\n", s.name, s.M, s.N, s.K, "v^T", ms, 2.0 * s.M * s.N * s.K / ms / 1e9);
        }
        if (getenv("SS_TRACE") && s.N == 5120 && s.M == 12000) {   // per-tile phase breakdown of the gelu epilogue kernel
            long long* tr; hipMalloc(&tr, 256 * 8 * 4 * 8); hipMemset(tr, 0, 256 * 8 * 4 * 8);
            for (int ki = 0; ki < 3; ki++) {
                GemmDesc g{};
                g.A = A; g.lda = s.K; g.a_rows_per_batch = 0; g.W = W; g.M = s.M; g.N = s.N; g.K = s.K; g.kind = kinds[ki]; g.bias = bias;
                g.out = out; g.ldo = s.N; g.o_rows_per_batch = 0; g.res = (float*)out; g.scale = 1.0f; g.rows_per_batch = 1500; g.trace = tr; g.gelu_f16_in = 1;
                hipEventRecord(e0, st);
                launch_gemm<f16>(g, st);
                hipEventRecord(e1, st); hipDeviceSynchronize();
                float tms = 0; hipEventElapsedTime(&tms, e0, e1);
                std::vector<long long> h(256 * 8 * 4);
                hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
                long long tmin = 1LL << 62;
                for (int b = 0; b < 256; b++) if (h[(b * 8) * 4]) tmin = std::min(tmin, h[(b * 8) * 4]);
                double pro = 0, loop = 0, epi = 0, gap = 0; int n = 0, ng = 0;
                for (int b = 0; b < 256; b++) for (int t = 0; t < 4; t++) {
                    const long long* q = &h[(b * 8 + t) * 4];
                    if (!q[3]) continue;
                    pro += q[1] - q[0]; loop += q[2] - q[1]; epi += q[3] - q[2]; n++;
                    if (t > 0) { gap += q[0] - h[(b * 8 + t - 1) * 4 + 3]; ng++; }
                }
                printf("   trace %-8s: per tile (s_memtime ticks = shader cycles) prologue %.0f  loop %.0f  epilogue %.0f  (tiles %d)\n", kn[ki], pro / n, loop / n, epi / n, n);
                {   // effective shader clock of this launch: ticks a persistent workgroup spent from its first tile's start to its last tile's end / the launch's wall time
                    double span = 0; int nb = 0;
                    for (int b = 0; b < 256; b++) { long long a0 = h[(b * 8) * 4], z = 0; for (int t = 0; t < 8; t++) if (h[(b * 8 + t) * 4 + 3]) z = h[(b * 8 + t) * 4 + 3]; if (a0 && z) { span += (double)(z - a0); nb++; } }
                    if (nb) printf("   clock %-8s: mean workgroup span %.0f ticks over a launch of %.3f ms (traced) => >= %.2f GHz effective shader clock; MFMA issue alone (2 waves x 32 MFMAs per SIMD and 32-deep k-step, 16 cycles each) = %.0f ticks per tile\n",
                                   kn[ki], span / nb, tms, span / nb / (tms * 1e6), (double)(s.K / 32) * 32 * 2 * 16);
                }
                for (int b : {0, 100, 200}) { printf("     wg %3d:", b); for (int t = 0; t < 4; t++) { const long long* q = &h[(b * 8 + t) * 4]; if (q[3]) printf("  [%lld +%lld +%lld +%lld]", q[0] - tmin, q[1] - q[0], q[2] - q[1], q[3] - q[2]); } printf("\n"); }
            }
            hipFree(tr);
        }
        if (s.M * (long)s.N <= 12000L * 5120) {  // correctness vs naive reference (EPI_STORE_F32)
            hipMalloc(&Cf, (size_t)s.M * s.N * 4);
            ref_gemm<<<dim3((s.N + 255) / 256, s.M), 256>>>(A, W, bias, Cf, s.M, s.N, s.K);
            GemmDesc g{};
            g.A = A; g.lda = s.K; g.a_rows_per_batch = 0; g.W = W; g.M = s.M; g.N = s.N; g.K = s.K; g.kind = EPI_STORE_F32; g.bias = bias;
            g.out = out; g.ldo = s.N; g.o_rows_per_batch = 0; g.scale = 1.0f; g.rows_per_batch = 1500;
            launch_gemm<f16>(g, st); hipDeviceSynchronize();
            std::vector<float> a((size_t)s.M * s.N), b((size_t)s.M * s.N);
            hipMemcpy(a.data(), out, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), Cf, b.size() * 4, hipMemcpyDeviceToHost);
            double mx = 0, ref = 0; size_t bad = 0, first_bad = 0;
            for (size_t i = 0; i < a.size(); i++) { if (!std::isfinite(a[i])) { if (!bad) first_bad = i; bad++; continue; } mx = fmax(mx, fabs(a[i] - b[i])); ref = fmax(ref, fabs(b[i])); }
            printf("   check: max|diff| %.3e (max|ref| %.3f)%s\n", mx, ref, bad ? "  NON-FINITE OUTPUTS" : "");
            if (bad) printf("   %zu non-finite outputs, first at row %zu column %zu\n", bad, first_bad / s.N, first_bad % s.N);
            { size_t nbig = 0, fb = 0; for (size_t i = 0; i < a.size(); i++) if (std::isfinite(a[i]) && fabs(a[i] - b[i]) > 1e-2) { if (!nbig) fb = i; nbig++; }
              if (nbig) printf("   %zu outputs off by more than 1e-2, first at row %zu column %zu: got %f want %f\n", nbig, fb / s.N, fb % s.N, a[fb], b[fb]); }
            g.kind = EPI_STORE_T;    // the 16-byte-store epilogue (lane regrouping by v_permlane16_swap)
            launch_gemm<f16>(g, st); hipDeviceSynchronize();
            std::vector<f16> h16((size_t)s.M * s.N);
            hipMemcpy(h16.data(), out, h16.size() * 2, hipMemcpyDeviceToHost);
            mx = 0; for (size_t i = 0; i < h16.size(); i++) mx = fmax(mx, fabs((float)h16[i] - b[i]));
            printf("   check f16 out: max|diff| %.3e\n", mx);
            hipFree(Cf);
        }
        hipFree(A); hipFree(W); hipFree(bias); hipFree(out);
    }
    return 0;
}
