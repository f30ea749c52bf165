// STFT denoiser on gfx950 -- SURVEY.md §8f "next" #1.
// Replaces the reference's CPU pre-stage `denoise_audio` (/root/reference/src/audio/mod.rs:507-735: noise-type analysis,
// spectral subtraction, Wiener filter, overlap-add with Hann^2 normalisation and the hard-coded x10 gain), which the gRPC
// handler runs in front of every 5 s chunk (src/grpc/handlers/asr.rs:196) and the REST stream pre-processor per 2048-sample
// frame (mod.rs:133-134).  rustfft's transforms are unnormalised; so are these.
// HBM/latency-bound: one workgroup per 2048-sample frame; window, radix-2 FFT, per-bin gain and inverse FFT stay in LDS,
// overlap-add is a gather (each output sample sums its <= 4 covering frames in frame order: deterministic, no atomics).
#include "kernels.h"

namespace ss {

namespace {
constexpr int kFs = 2048, kLog = 11, kThreads = 256;

__device__ __forceinline__ float hann_dn(int i) {   // mod.rs:503-505 (f32 arithmetic)
    return 0.5f * (1.0f - cosf(2.0f * 3.14159265358979323846f * (float)i / (float)(kFs - 1)));
}

// in-place radix-2 decimation-in-time FFT of 2048 complex values in LDS (input already in bit-reversed order).
// tw[k] = exp(-2 pi i k / 2048), k < 1024; inverse uses the conjugate.  Unnormalised in both directions.
__device__ void fft2048(float2* s, const float2* __restrict__ tw, bool inverse) {
    const int tid = threadIdx.x;
    for (int st = 0; st < kLog; st++) {
        const int half = 1 << st, shift = kLog - 1 - st;
#pragma unroll
        for (int b = 0; b < kFs / 2 / kThreads; b++) {
            const int idx = b * kThreads + tid;
            const int j = idx & (half - 1), blk = idx >> st;
            const int i0 = blk * (half << 1) + j, i1 = i0 + half;
            float2 w = tw[j << shift];
            if (inverse) w.y = -w.y;
            const float2 a = s[i0], c = s[i1];
            const float2 t = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
            s[i0] = make_float2(a.x + t.x, a.y + t.y);
            s[i1] = make_float2(a.x - t.x, a.y - t.y);
        }
        __syncthreads();
    }
}
__device__ __forceinline__ int bitrev11(int i) { return (int)(__brev((unsigned)i) >> (32 - kLog)); }

// power spectrum of the non-overlapping 2048-sample chunks (analysis / noise / signal estimates)
__global__ __launch_bounds__(kThreads) void dn_chunk_power_kernel(const float* __restrict__ x, const float2* __restrict__ tw, float* __restrict__ power) {
    __shared__ float2 s[kFs];
    const float* fr = x + (size_t)blockIdx.x * kFs;
    for (int i = threadIdx.x; i < kFs; i += kThreads) s[bitrev11(i)] = make_float2(fr[i] * hann_dn(i), 0.0f);
    __syncthreads();
    fft2048(s, tw, false);
    for (int i = threadIdx.x; i < kFs; i += kThreads) power[(size_t)blockIdx.x * kFs + i] = s[i].x * s[i].x + s[i].y * s[i].y;
}

// per-bin noise (first <=20 chunks, each /20) and signal (all chunks, each /n_chunks) spectra; per-pair spectral variance
__global__ void dn_spectra_kernel(const float* __restrict__ power, int n_chunks, float* __restrict__ noise, float* __restrict__ signal) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kFs) return;
    float nz = 0.f, sg = 0.f;
    const float nf = (float)n_chunks;
    for (int c = 0; c < n_chunks; c++) {
        const float p = power[(size_t)c * kFs + i];
        if (c < 20) nz += p / 20.0f;
        sg += p / nf;
    }
    noise[i] = nz;
    signal[i] = sg;
}
__global__ __launch_bounds__(kThreads) void dn_variance_kernel(const float* __restrict__ power, float* __restrict__ var_out) {
    __shared__ float red[kThreads / 64];
    const float* a = power + (size_t)blockIdx.x * kFs;
    const float* b = a + kFs;
    float acc = 0.f;
    for (int i = threadIdx.x; i < kFs; i += kThreads) { const float d = b[i] - a[i]; acc += d * d; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) var_out[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) / (float)kFs;
}

// one sliding frame: window -> FFT -> gain -> inverse FFT -> real part * window
template <int MODE>   // 0 spectral subtraction (mod.rs:581-624), 1 Wiener (mod.rs:626-662)
__global__ __launch_bounds__(kThreads) void dn_frame_kernel(const float* __restrict__ x, int step, const float2* __restrict__ tw,
                                                            const float* __restrict__ noise, const float* __restrict__ signal, float strength,
                                                            float* __restrict__ frames_out) {
    __shared__ float2 s[kFs];
    __shared__ float2 t[kFs];
    const float* fr = x + (size_t)blockIdx.x * step;
    for (int i = threadIdx.x; i < kFs; i += kThreads) s[bitrev11(i)] = make_float2(fr[i] * hann_dn(i), 0.0f);
    __syncthreads();
    fft2048(s, tw, false);
    for (int i = threadIdx.x; i < kFs; i += kThreads) {
        const float2 c = s[i];
        float gain;
        if (MODE == 0) {
            const float power = c.x * c.x + c.y * c.y;
            const float freq_factor = fminf((float)i / (float)kFs, 1.0f);
            const float freq_strength = strength * (1.0f - 0.3f * freq_factor);
            gain = sqrtf(fmaxf(1.0f - 1.0f * powf(noise[i] / (power + 1e-6f), freq_strength), 0.1f));
        } else {
            const float snr = signal[i] / (noise[i] + 1e-6f);
            gain = powf(snr / (1.0f + snr), strength * 0.7f);
        }
        t[bitrev11(i)] = make_float2(c.x * gain, c.y * gain);
    }
    __syncthreads();
    fft2048(t, tw, true);
    for (int i = threadIdx.x; i < kFs; i += kThreads) frames_out[(size_t)blockIdx.x * kFs + i] = t[i].x * hann_dn(i);
}

// overlap-add as a gather + normalise + x10 (mod.rs:711-735)
__global__ void dn_overlap_add_kernel(const float* __restrict__ frames, int n_frames, int step, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int f0 = (i - (kFs - 1) + step - 1) / step;
    if (i - (kFs - 1) < 0) f0 = 0;
    int f1 = i / step;
    if (f1 > n_frames - 1) f1 = n_frames - 1;
    float acc = 0.f, norm = 0.f;
    for (int f = f0; f <= f1; f++) {
        const int j = i - f * step;
        const float w = hann_dn(j);
        acc += frames[(size_t)f * kFs + j];
        norm += w * w;
    }
    out[i] = norm > 1e-10f ? (acc / norm) * 10.0f : acc;
}

__global__ void dn_noise_gate_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float gate) {   // mod.rs:495-500
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = in[i]; out[i] = fabsf(v) < gate ? 0.0f : v; }
}
}  // namespace

void launch_dn_chunk_power(const float* x, int n_chunks, const float2* tw, float* power, hipStream_t st) {
    if (n_chunks > 0) { dn_chunk_power_kernel<<<n_chunks, kThreads, 0, st>>>(x, tw, power); SS_LAUNCH_CHECK(); }
}
void launch_dn_spectra(const float* power, int n_chunks, float* noise, float* signal, float* var_out, hipStream_t st) {
    dn_spectra_kernel<<<kFs / 256, 256, 0, st>>>(power, n_chunks, noise, signal); SS_LAUNCH_CHECK();
    if (n_chunks > 1) { dn_variance_kernel<<<n_chunks - 1, kThreads, 0, st>>>(power, var_out); SS_LAUNCH_CHECK(); }
}
void launch_dn_frames(int mode, const float* x, int n_frames, int step, const float2* tw, const float* noise, const float* signal, float strength,
                      float* frames_out, hipStream_t st) {
    if (mode == 0) { dn_frame_kernel<0><<<n_frames, kThreads, 0, st>>>(x, step, tw, noise, signal, strength, frames_out); SS_LAUNCH_CHECK(); }
    else { dn_frame_kernel<1><<<n_frames, kThreads, 0, st>>>(x, step, tw, noise, signal, strength, frames_out); SS_LAUNCH_CHECK(); }
}
void launch_dn_overlap_add(const float* frames, int n_frames, int step, int n, float* out, hipStream_t st) {
    dn_overlap_add_kernel<<<(n + 255) / 256, 256, 0, st>>>(frames, n_frames, step, n, out); SS_LAUNCH_CHECK();
}
void launch_dn_noise_gate(const float* in, float* out, int n, float gate, hipStream_t st) {
    dn_noise_gate_kernel<<<(n + 255) / 256, 256, 0, st>>>(in, out, n, gate); SS_LAUNCH_CHECK();
}

}  // namespace ss
