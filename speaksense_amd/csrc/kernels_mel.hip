// log-mel spectrogram on gfx950.
// Replaces whisper.cpp's log_mel_spectrogram (reached from /root/reference/src/asr/whisper.rs:75 via whisper_full;
// spec in SURVEY.md §8 a-3).  HBM/latency-bound, ~0.2 GFLOP per 30 s chunk: one workgroup per 25 ms frame, the whole
// frame pipeline (window, FFT-400, power, mel dot, log10) stays in LDS; one extra pass applies the global-max clamp.
//
// To stay within 1e-4 of the CPU path in low-energy bins the FFT keeps whisper.cpp's structure and op order:
// radix-2 decimation in time down to 16 DFT-25 leaves, f32, un-fused multiply/add, sin/cos from a 400-entry table.
#include "kernels.h"
#include "wave_ops.h"

namespace ss {

namespace {

constexpr int kFrameThreads = 256;

__device__ __forceinline__ float padded_sample(const float* pcm, int n, long i) {
    // sp[i]: 200 reflected samples, the audio, then zeros (30 s + 200)
    if (i < 200) return n > 200 ? pcm[200 - i] : 0.0f;
    long j = i - 200;
    return j < n ? pcm[j] : 0.0f;
}

__global__ __launch_bounds__(kFrameThreads) void mel_frame_kernel(MelTables mt, const float* __restrict__ pcm, int n_samples,
                                                                   float* __restrict__ mel, int n_len, int n_frames_active,
                                                                   float* __restrict__ frame_max) {
    __shared__ float s_sin[400], s_cos[400];
    __shared__ float s_in[400];
    __shared__ float s_a[800], s_b[800];  // ping-pong complex buffers
    __shared__ float s_red[kFrameThreads / 64];
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const int n_mel = mt.n_mel;

    if (frame >= n_frames_active) {  // frame lies entirely in the zero padding: log10(1e-10)
        for (int j = tid; j < n_mel; j += kFrameThreads) mel[(size_t)j * n_len + frame] = -10.0f;
        if (tid == 0) frame_max[frame] = -10.0f;
        return;
    }
    for (int i = tid; i < 400; i += kFrameThreads) {
        s_sin[i] = mt.sin_t[i];
        s_cos[i] = mt.cos_t[i];
        s_in[i] = __fmul_rn(mt.hann[i], padded_sample(pcm, n_samples, (long)frame * kHop + i));
    }
    __syncthreads();
    // stage 0: 16 DFT-25 leaves; leaf r holds x[16 n + r]; F0[r][k] at s_a[(r*25 + k)*2]
    for (int t = tid; t < 400; t += kFrameThreads) {
        const int r = t / 25, k = t % 25;
        float re = 0.0f, im = 0.0f;
        for (int n = 0; n < 25; n++) {
            const int idx = (k * n * 16) % 400;
            const float x = s_in[16 * n + r];
            re = __fadd_rn(re, __fmul_rn(x, s_cos[idx]));
            im = __fsub_rn(im, __fmul_rn(x, s_sin[idx]));
        }
        s_a[2 * t] = re;
        s_a[2 * t + 1] = im;
    }
    __syncthreads();
    // stages 1..4: F_s[rho] (size 25*2^s) = combine(F_{s-1}[rho], F_{s-1}[rho + 16/2^s])
    float* src = s_a;
    float* dst = s_b;
#pragma unroll
    for (int s = 1; s <= 4; s++) {
        const int N = 25 << s, h = N >> 1, nsub = 16 >> s, step = 400 / N;
        if (tid < 200) {
            const int rho = tid / h, k = tid % h;
            const float* E = src + 2 * (rho * h + k);
            const float* O = src + 2 * ((rho + nsub) * h + k);
            const int idx = k * step;
            const float re = s_cos[idx], im = -s_sin[idx];
            const float er = E[0], ei = E[1], re_odd = O[0], im_odd = O[1];
            float* o0 = dst + 2 * (rho * N + k);
            float* o1 = dst + 2 * (rho * N + k + h);
            o0[0] = __fsub_rn(__fadd_rn(er, __fmul_rn(re, re_odd)), __fmul_rn(im, im_odd));
            o0[1] = __fadd_rn(__fadd_rn(ei, __fmul_rn(re, im_odd)), __fmul_rn(im, re_odd));
            o1[0] = __fadd_rn(__fsub_rn(er, __fmul_rn(re, re_odd)), __fmul_rn(im, im_odd));
            o1[1] = __fsub_rn(__fsub_rn(ei, __fmul_rn(re, im_odd)), __fmul_rn(im, re_odd));
        }
        __syncthreads();
        float* t2 = src; src = dst; dst = t2;
    }
    // power spectrum for bins 0..200 into dst (reuse as real array)
    for (int k = tid; k < kNBins; k += kFrameThreads) {
        const float a = src[2 * k], b = src[2 * k + 1];
        dst[k] = __fadd_rn(__fmul_rn(a, a), __fmul_rn(b, b));
    }
    __syncthreads();
    float vmax = -1e30f;
    for (int j = tid; j < n_mel; j += kFrameThreads) {
        const float* fl = mt.filt + (size_t)j * kNBins;
        double sum = 0.0;
        int k = 0;
        for (; k < kNBins - 3; k += 4) {
            float part = __fmul_rn(dst[k], fl[k]);
            part = __fadd_rn(part, __fmul_rn(dst[k + 1], fl[k + 1]));
            part = __fadd_rn(part, __fmul_rn(dst[k + 2], fl[k + 2]));
            part = __fadd_rn(part, __fmul_rn(dst[k + 3], fl[k + 3]));
            sum += (double)part;
        }
        for (; k < kNBins; k++) sum += (double)__fmul_rn(dst[k], fl[k]);
        sum = log10(sum > 1e-10 ? sum : 1e-10);
        const float v = (float)sum;
        mel[(size_t)j * n_len + frame] = v;
        vmax = fmaxf(vmax, v);
    }
    vmax = wave_max(vmax);
    if ((tid & 63) == 0) s_red[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) {
        float m = s_red[0];
        for (int i = 1; i < kFrameThreads / 64; i++) m = fmaxf(m, s_red[i]);
        frame_max[frame] = m;
    }
}

// x = (max(x, gmax - 8) + 4) / 4 in double, as whisper.cpp does after the worker threads join
__global__ __launch_bounds__(256) void mel_norm_kernel(float* __restrict__ mel, size_t n, const float* __restrict__ frame_max, int n_frames) {
    __shared__ float s_red[4];
    __shared__ double s_mmax;
    float m = -1e30f;
    for (int i = threadIdx.x; i < n_frames; i += 256) m = fmaxf(m, frame_max[i]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) s_mmax = (double)fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])) - 8.0;
    __syncthreads();
    const double mmax = s_mmax;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double v = (double)mel[i];
        if (v < mmax) v = (double)(float)mmax;   // the clamp stores a float
        mel[i] = (float)((v + 4.0) / 4.0);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mel_window_kernel(const float* __restrict__ mel, int n_mel, int n_len, int seek, int T2, T* __restrict__ x0) {
    // 32 frames x 32 mels tile transposed through LDS: reads coalesced along time, writes along mel
    __shared__ float tile[32][33];
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        float v = 0.0f;
        if (c < n_mel && t < T2 && seek + t < n_len) v = mel[(size_t)c * n_len + seek + t];
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T2 && c < n_mel) x0[(size_t)(t + 1) * n_mel + c] = (T)tile[tx][i];
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {  // zero pad rows (conv padding)
        const int c = c0 + threadIdx.x;
        if (c < n_mel) { x0[c] = (T)0.0f; x0[(size_t)(T2 + 1) * n_mel + c] = (T)0.0f; }
    }
}

// whisper.cpp get_signal_energy(samples, n, 32) (token-level timestamps, SURVEY.md section 8 a-8 / whisper.rs:160): e[i] = mean |x| over the centred
// window [i - 32, i + 32] clipped to the signal, summed in f32 from left to right and divided by 65 -- the reference's order, so the values are the
// CPU's bit for bit (terms outside the signal are skipped there; adding +0 to a non-negative sum is the same thing).
constexpr int kEnergyHw = 32, kEnergyBlock = 256;
__global__ __launch_bounds__(kEnergyBlock) void signal_energy_kernel(const float* __restrict__ x, int n, float* __restrict__ e) {
    __shared__ float ax[kEnergyBlock + 2 * kEnergyHw];
    const int i0 = blockIdx.x * kEnergyBlock;
    for (int t = threadIdx.x; t < kEnergyBlock + 2 * kEnergyHw; t += kEnergyBlock) {
        const int k = i0 - kEnergyHw + t;
        ax[t] = (k >= 0 && k < n) ? fabsf(x[k]) : 0.0f;
    }
    __syncthreads();
    const int i = i0 + threadIdx.x;
    if (i >= n) return;
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j <= 2 * kEnergyHw; j++) sum = __fadd_rn(sum, ax[threadIdx.x + j]);
    e[i] = __fdiv_rn(sum, (float)(2 * kEnergyHw + 1));
}

}  // namespace

void launch_signal_energy(const float* pcm, int n_samples, float* energy, hipStream_t st) {
    if (n_samples <= 0) return;
    signal_energy_kernel<<<(n_samples + kEnergyBlock - 1) / kEnergyBlock, kEnergyBlock, 0, st>>>(pcm, n_samples, energy); SS_LAUNCH_CHECK();
}

void launch_log_mel(const MelTables& mt, const float* pcm, int n_samples, float* mel_out, int n_len, float* scratch, hipStream_t st) {
    // frames whose 400-sample window starts at or beyond the end of the audio (offset >= 200 + n) are all-zero
    int n_active = (200 + n_samples + kHop - 1) / kHop;
    if (n_active > n_len) n_active = n_len;
    mel_frame_kernel<<<n_len, kFrameThreads, 0, st>>>(mt, pcm, n_samples, mel_out, n_len, n_active, scratch); SS_LAUNCH_CHECK();
    const size_t n = (size_t)mt.n_mel * n_len;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    mel_norm_kernel<<<blocks, 256, 0, st>>>(mel_out, n, scratch, n_len); SS_LAUNCH_CHECK();
}

template <typename T>
void launch_mel_window(const float* mel, int n_mel, int n_len, int seek, int T2, T* x0, hipStream_t st) {
    dim3 grid((T2 + 31) / 32, (n_mel + 31) / 32);
    mel_window_kernel<T><<<grid, 256, 0, st>>>(mel, n_mel, n_len, seek, T2, x0); SS_LAUNCH_CHECK();
}
template void launch_mel_window<bf16>(const float*, int, int, int, int, bf16*, hipStream_t);
template void launch_mel_window<f16>(const float*, int, int, int, int, f16*, hipStream_t);

}  // namespace ss
