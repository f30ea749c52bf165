// Sinc resampler to 16 kHz on gfx950 -- SURVEY.md §8f "next" #4.
// Replaces `create_resampler` + `resample_chunk` of /root/reference/src/audio/mod.rs:235-257: rubato 0.16.0 SincFixedIn<f32>
// (sinc_len 256, cutoff 0.95, linear blend of the two nearest of 256 sub-phases, BlackmanHarris2 window, 4096-sample chunks).
// The output instants are a sequential f64 recurrence (idx += 1/ratio) that also decides how many samples each chunk yields; the host
// runs it (a few thousand additions per second of audio) and hands the instants over, so lengths and tap positions are exact.
// One thread per output sample: two 256-tap dot products with the crate's 8-lane accumulation order.  HBM-light (input read once,
// the 256 KB phase table lives in L2), so no LDS staging.
#include "kernels.h"

namespace ss {
namespace {
constexpr int kTaps = 256, kPhases = 256, kChunk = 4096;

__device__ __forceinline__ float dot8(const float* __restrict__ x, long g0, const float* __restrict__ sinc) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < kTaps; j += 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const long g = g0 + j + u;
            const float v = g >= 0 ? x[g] : 0.0f;     // 2*sinc_len samples of zero history before the stream
            acc[u] = __fadd_rn(acc[u], __fmul_rn(v, sinc[j + u]));
        }
    }
    float t = 0.0f;
#pragma unroll
    for (int u = 0; u < 8; u++) t = __fadd_rn(t, acc[u]);
    return t;
}

__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, const double* __restrict__ idx_rel, const int* __restrict__ chunk_of,
                                                       long n_out, const float* __restrict__ sincs, float* __restrict__ y) {
    const long k = (long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_out) return;
    const double idx = idx_rel[k];
    const double fl = floor(idx);
    int sub0 = (int)floor((idx - fl) * (double)kPhases);
    long i0 = (long)fl, i1 = i0;
    int sub1 = sub0 + 1;
    if (sub1 >= kPhases) { sub1 -= kPhases; i1 += 1; }
    const double ov = idx * (double)kPhases;
    const float frac = (float)(ov - floor(ov));
    const long base = (long)chunk_of[k] * kChunk;
    const float p0 = dot8(x, base + i0, sincs + (size_t)sub0 * kTaps);
    const float p1 = dot8(x, base + i1, sincs + (size_t)sub1 * kTaps);
    y[k] = __fadd_rn(p0, __fmul_rn(frac, __fsub_rn(p1, p0)));
}
}  // namespace

void launch_resample(const float* x, const double* idx_rel, const int* chunk_of, long n_out, const float* sincs, float* y, hipStream_t st) {
    if (n_out <= 0) return;
    resample_kernel<<<(unsigned)((n_out + 255) / 256), 256, 0, st>>>(x, idx_rel, chunk_of, n_out, sincs, y); SS_LAUNCH_CHECK();
}

}  // namespace ss
