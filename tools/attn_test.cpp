// Dev tool: run launch_enc_attention on seeded random q/k/vT for several (B, H, Tn) and write the outputs; compare two runs made with
// different SS_ATTN_LDS settings.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/attn_test.cpp speaksense_amd/csrc/kernels_attn.hip -Ispeaksense_amd/csrc
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
using namespace ss;
__global__ void fillh(f16* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 2246822519u; x ^= x >> 13;
        p[i] = (f16)(((x & 0xffff) / 32768.0f - 1.0f) * scale);
    }
}
int main(int argc, char** argv) {
    const char* outp = argc > 1 ? argv[1] : "/tmp/attn_out.bin";
    FILE* f = fopen(outp, "wb");
    struct Cfg { int B, H, Tn; };
    Cfg cfgs[] = {{1, 2, 1500}, {3, 2, 1500}, {2, 20, 1500}, {1, 2, 100}, {3, 6, 1471}};
    hipStream_t st; hipStreamCreate(&st);
    for (auto& c : cfgs) {
        const int d = c.H * 64, ld = 2 * d, Tpad = (c.Tn + 63) / 64 * 64;
        const size_t nqk = (size_t)c.B * c.Tn * ld, nv = (size_t)c.B * c.H * 64 * Tpad, no = (size_t)c.B * c.Tn * d;
        f16 *qk, *vT, *out;
        hipMalloc(&qk, nqk * 2); hipMalloc(&vT, nv * 2); hipMalloc(&out, no * 2);
        hipMemset(vT, 0, nv * 2); hipMemset(out, 0, no * 2);
        fillh<<<512, 256>>>(qk, nqk, 7, 2.0f);
        // V^T: valid columns only (the pad columns stay zero as in the engine)
        std::vector<f16> hv(nv, (f16)0.f);
        for (size_t r = 0; r < (size_t)c.B * c.H * 64; r++) for (int t = 0; t < c.Tn; t++) { unsigned x = (unsigned)(r * 1543 + t) * 2654435761u; x ^= x >> 15; hv[r * Tpad + t] = (f16)(((x & 0xffff) / 32768.0f - 1.0f)); }
        hipMemcpy(vT, hv.data(), nv * 2, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        launch_enc_attention<f16>(qk, qk + d, ld, vT, Tpad, out, d, c.B, c.H, c.Tn, st);
        hipStreamSynchronize(st);
        std::vector<f16> ho(no);
        hipMemcpy(ho.data(), out, no * 2, hipMemcpyDeviceToHost);
        fwrite(ho.data(), 2, no, f);
        // also: batch element 0 of this config computed alone must equal element 0 computed in the batch
        if (c.B > 1) {
            hipMemset(out, 0, no * 2);
            launch_enc_attention<f16>(qk, qk + d, ld, vT, Tpad, out, d, 1, c.H, c.Tn, st);
            hipStreamSynchronize(st);
            std::vector<f16> h1((size_t)c.Tn * d);
            hipMemcpy(h1.data(), out, h1.size() * 2, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < h1.size(); i++) if ((float)h1[i] != (float)ho[i]) bad++;
            printf("B=%d H=%d Tn=%d: batch element 0 alone vs in batch: %zu differing values\n", c.B, c.H, c.Tn, bad);
            for (int bb = 1; bb < c.B; bb++) {   // every other batch element alone (pointers offset) vs in the batch
                f16* o1; hipMalloc(&o1, (size_t)c.Tn * d * 2); hipMemset(o1, 0, (size_t)c.Tn * d * 2);
                launch_enc_attention<f16>(qk + (size_t)bb * c.Tn * ld, qk + (size_t)bb * c.Tn * ld + d, ld, vT + (size_t)bb * c.H * 64 * Tpad, Tpad, o1, d, 1, c.H, c.Tn, st);
                hipStreamSynchronize(st);
                hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost);
                bad = 0;
                for (size_t i = 0; i < h1.size(); i++) if ((float)h1[i] != (float)ho[(size_t)bb * c.Tn * d + i]) bad++;
                printf("   batch element %d alone vs in batch: %zu differing values\n", bb, bad);
                hipFree(o1);
            }
            // run the batch again: determinism
            hipMemset(out, 0, no * 2);
            launch_enc_attention<f16>(qk, qk + d, ld, vT, Tpad, out, d, c.B, c.H, c.Tn, st);
            hipStreamSynchronize(st);
            std::vector<f16> h2(no);
            hipMemcpy(h2.data(), out, no * 2, hipMemcpyDeviceToHost);
            bad = 0;
            for (size_t i = 0; i < no; i++) if ((float)h2[i] != (float)ho[i]) bad++;
            printf("   same launch twice: %zu differing values\n", bad);
        }
        hipFree(qk); hipFree(vT); hipFree(out);
    }
    fclose(f);
    return 0;
}
