// STFT denoiser on gfx950 -- SURVEY.md §8f "next" #1.
// Replaces the reference's CPU pre-stage `denoise_audio` (/root/reference/src/audio/mod.rs:507-735: noise-type analysis,
// spectral subtraction, Wiener filter, overlap-add with Hann^2 normalisation and the hard-coded x10 gain), which the gRPC
// handler runs in front of every 5 s chunk (src/grpc/handlers/asr.rs:196) and the REST stream pre-processor per 2048-sample
// frame (mod.rs:133-134).  rustfft's transforms are unnormalised; so are these.
// HBM/latency-bound: one workgroup per 2048-sample frame; window, radix-2 FFT, per-bin gain and inverse FFT stay in LDS,
// overlap-add is a gather (each output sample sums its <= 4 covering frames in frame order: deterministic, no atomics).
#include "kernels.h"
#include "wave_ops.h"

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
    acc = wave_sum(acc);
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
// ------------------------------------------------------------------------------------------------
// stream pre-processor (StreamAudioProcessor, /root/reference/src/audio/mod.rs:67-155) for a whole 16 kHz mono stream at once:
// per-read-chunk peak normalisation -> per-2048-frame pre-emphasis energy -> the (sequential, scalar) gain recurrence ->
// one fused kernel per frame: gain, single-frame spectral subtraction, Hann^2 normalisation x10, noise gate.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    v = is_max ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < kThreads / 64; i++) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}
// normalize_audio (mod.rs:408-411) per read chunk: x / max|x| (an all-zero chunk becomes NaN, as in the reference)
__global__ __launch_bounds__(kThreads) void pp_normalize_kernel(const float* __restrict__ x, long n, const long* __restrict__ chunk_off, int chunk_len,
                                                                float* __restrict__ y) {
    __shared__ float red[kThreads / 64];
    const long b = chunk_off ? chunk_off[blockIdx.x] : (long)blockIdx.x * chunk_len;
    long e = chunk_off ? chunk_off[blockIdx.x + 1] : b + chunk_len;
    if (e > n) e = n;
    float m = 0.0f;
    for (long i = b + threadIdx.x; i < e; i += kThreads) m = fmaxf(m, fabsf(x[i]));
    m = block_reduce(m, red, true);
    for (long i = b + threadIdx.x; i < e; i += kThreads) y[i] = x[i] / m;
}
// per frame: mean of the squared pre-emphasised samples (mod.rs:110-113), and the two 1024-sample energies estimate_noise_floor looks at
__global__ __launch_bounds__(kThreads) void pp_energy_kernel(const float* __restrict__ y, long n, float* __restrict__ energy, float* __restrict__ sub_energy) {
    __shared__ float red[kThreads / 64];
    const long base = (long)blockIdx.x * kFs;
    float acc = 0.f, lo = 0.f, hi = 0.f;
    for (int i = threadIdx.x; i < kFs; i += kThreads) {
        const float cur = base + i < n ? y[base + i] : 0.0f;
        const float prev = (i > 0 && base + i - 1 < n) ? y[base + i - 1] : 0.0f;
        const float p = i == 0 ? cur : cur - 0.97f * prev;
        acc += p * p;
        if (i < 1024) lo += cur * cur; else hi += cur * cur;
    }
    acc = block_reduce(acc, red, false);
    lo = block_reduce(lo, red, false);
    hi = block_reduce(hi, red, false);
    if (threadIdx.x == 0) { energy[blockIdx.x] = acc / (float)kFs; sub_energy[2 * blockIdx.x] = lo / 1024.0f; sub_energy[2 * blockIdx.x + 1] = hi / 1024.0f; }
}
// the scalar recurrence over frames (mod.rs:97-99,114-127).  fmaxf/fminf ignore NaN exactly as Rust's f32::max/min do, so the
// reference's NaN noise floor (estimate_noise_floor keeps (2 as f32 * 0.1) as usize == 0 frames -> 0/0) yields gain 0.1 here too.
__global__ void pp_gain_kernel(const float* __restrict__ energy, const float* __restrict__ sub_energy, int n_frames, int n_full_frames,
                               float* __restrict__ gain) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float noise_floor = 0.0f, prev_energy = 0.0f;
    for (int f = 0; f < n_frames; f++) {
        if (noise_floor == 0.0f && f < n_full_frames) {   // only process_chunk initialises the floor, finish() (the padded frame) does not
            float e0 = sub_energy[2 * f], e1 = sub_energy[2 * f + 1];
            if (e1 < e0) { const float t = e0; e0 = e1; e1 = t; }
            const int cnt = (int)(2.0f * 0.1f);
            float s = 0.0f;
            if (cnt > 0) s += e0;
            if (cnt > 1) s += e1;
            volatile float den = (float)cnt;
            noise_floor = s / den;
        }
        const float e = energy[f];
        const float threshold = noise_floor * 1.2f + prev_energy * 0.1f;
        gain[f] = e > threshold ? 1.0f : fmaxf(e / threshold, 0.1f);
        prev_energy = e;
        noise_floor = noise_floor * 0.95f + fminf(e, noise_floor) * 0.05f;
    }
}
// gain -> denoise_audio on the single frame (noise type is Stationary for one chunk: variance 0) -> noise gate
__global__ __launch_bounds__(kThreads) void pp_frame_kernel(const float* __restrict__ y, long n, const float* __restrict__ gain, const float2* __restrict__ tw,
                                                            float strength, float gate, int denoise, float* __restrict__ out) {
    __shared__ float2 s[kFs];
    __shared__ float2 t[kFs];
    const long base = (long)blockIdx.x * kFs;
    const float g = gain[blockIdx.x];
    if (!denoise) {
        for (int i = threadIdx.x; i < kFs; i += kThreads) {
            const float v = (base + i < n ? y[base + i] : 0.0f) * g;
            out[base + i] = fabsf(v) < gate ? 0.0f : v;
        }
        return;
    }
    for (int i = threadIdx.x; i < kFs; i += kThreads) s[bitrev11(i)] = make_float2(((base + i < n ? y[base + i] : 0.0f) * g) * hann_dn(i), 0.0f);
    __syncthreads();
    fft2048(s, tw, false);
    for (int i = threadIdx.x; i < kFs; i += kThreads) {
        const float2 c = s[i];
        const float power = c.x * c.x + c.y * c.y;
        const float noise = 0.0f + power / 20.0f;    // estimate_noise_spectrum over this one chunk (mod.rs:664-686)
        const float freq_factor = fminf((float)i / (float)kFs, 1.0f);
        const float freq_strength = strength * (1.0f - 0.3f * freq_factor);
        const float gn = sqrtf(fmaxf(1.0f - 1.0f * powf(noise / (power + 1e-6f), freq_strength), 0.1f));
        t[bitrev11(i)] = make_float2(c.x * gn, c.y * gn);
    }
    __syncthreads();
    fft2048(t, tw, true);
    for (int i = threadIdx.x; i < kFs; i += kThreads) {
        const float w = hann_dn(i);
        const float acc = t[i].x * w, norm = w * w;
        const float v = norm > 1e-10f ? (acc / norm) * 10.0f : acc;
        out[base + i] = fabsf(v) < gate ? 0.0f : v;
    }
}
}  // namespace

void launch_pp_stream(const float* x, long n, const long* chunk_off, int n_chunks, int chunk_len, int n_frames, const float2* tw, float strength, float gate,
                      int denoise, float* y, float* energy, float* sub_energy, float* gain, float* out, hipStream_t st) {
    pp_normalize_kernel<<<n_chunks, kThreads, 0, st>>>(x, n, chunk_off, chunk_len, y); SS_LAUNCH_CHECK();
    pp_energy_kernel<<<n_frames, kThreads, 0, st>>>(y, n, energy, sub_energy); SS_LAUNCH_CHECK();
    pp_gain_kernel<<<1, 64, 0, st>>>(energy, sub_energy, n_frames, (int)(n / kFs), gain); SS_LAUNCH_CHECK();
    pp_frame_kernel<<<n_frames, kThreads, 0, st>>>(y, n, gain, tw, strength, gate, denoise, out); SS_LAUNCH_CHECK();
}

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
