// Dev tool: every helper of wave_ops.h against the __shfl_xor butterfly it replaces.  hipcc --offload-arch=gfx950 -O3 -Ispeaksense_amd/csrc tools/waveops_test.cpp
#include <cstdio>
#include <vector>
#include "wave_ops.h"
using namespace ss;
__global__ void k(const float* in, float* out) {
    const int t = threadIdx.x;
    float v = in[t];
    float a = v; for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    float b = v; for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o));
    float c = v; c += __shfl_xor(c, 1); c += __shfl_xor(c, 2); c += __shfl_xor(c, 4);
    float d = v; d += __shfl_xor(d, 8); d += __shfl_xor(d, 16); d += __shfl_xor(d, 32);
    float e = v; e = fmaxf(e, __shfl_xor(e, 16)); e = fmaxf(e, __shfl_xor(e, 32));
    float f = v; f += __shfl_xor(f, 16); f += __shfl_xor(f, 32);
    out[t * 12 + 0] = a; out[t * 12 + 1] = wave_sum(v);
    out[t * 12 + 2] = b; out[t * 12 + 3] = wave_max(v);
    out[t * 12 + 4] = c; out[t * 12 + 5] = sum_lanes8(v);
    out[t * 12 + 6] = d; out[t * 12 + 7] = sum_stride8(v);
    out[t * 12 + 8] = e; out[t * 12 + 9] = rows_max(v);
    out[t * 12 + 10] = f; out[t * 12 + 11] = rows_sum(v);
}
int main() {
    std::vector<float> h(64), o(64 * 12);
    for (int i = 0; i < 64; i++) h[i] = (float)((i * 37 + 11) % 101) - 50.25f + i * 0.001f;
    float *di, *dout; hipMalloc(&di, 256); hipMalloc(&dout, 64 * 12 * 4);
    hipMemcpy(di, h.data(), 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout); hipDeviceSynchronize();
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    const char* names[] = {"wave_sum", "wave_max", "sum_lanes8", "sum_stride8", "rows_max", "rows_sum"};
    for (int j = 0; j < 6; j++) {
        int bad = 0, first = -1;
        for (int t = 0; t < 64; t++) if (o[t * 12 + 2 * j] != o[t * 12 + 2 * j + 1]) { if (first < 0) first = t; bad++; }
        printf("%-12s mismatching lanes %d", names[j], bad);
        if (bad) printf("  (lane %d: shfl %.6f vs %.6f)", first, o[first * 12 + 2 * j], o[first * 12 + 2 * j + 1]);
        printf("\n");
    }
    return 0;
}
