"""ORACLE (test infrastructure only): numpy restatement of the reference's STFT denoiser.

Follows /root/reference/src/audio/mod.rs function by function, f32 arithmetic where the Rust is f32:
  hann_window :503-505, denoise_audio :507-523, analyze_noise_characteristics :533-579, spectral_subtraction :581-624,
  wiener_filter :626-662, estimate_noise_spectrum :664-686, estimate_signal_spectrum :688-709, overlap_add :711-735,
  apply_noise_gate :495-500.
rustfft's forward/inverse transforms are unnormalised (the inverse does NOT divide by N) -- kept, together with the
hard-coded x10 gain at :730, so the output is ~20480x the input scale exactly as in the reference.
The reference's own tests hold no vectors for this function (test_audio_processing needs an absent ./test/a.wav), so
this restatement is pinned by construction (the Rust source is in /root/reference) and by properties in tests/.
Differences from the Rust that remain: FFT butterfly order (pocketfft vs rustfft) and pairwise vs sequential f32 sums.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import scipy.fft as sfft

STATIONARY, NON_STATIONARY, MIXED = 0, 1, 2


@dataclass
class DenoiseConfig:  # mod.rs:41-61
    frame_size: int = 2048
    overlap: float = 0.75
    strength: float = 0.2
    noise_gate: float = 0.003
    enable_noise_reduction: bool = True
    threshold: float = 0.002


def hann_window(size: int) -> np.ndarray:
    i = np.arange(size, dtype=np.float32)
    return (np.float32(0.5) * (np.float32(1.0) - np.cos(np.float32(2.0) * np.float32(np.pi) * i / np.float32(size - 1), dtype=np.float32))).astype(np.float32)


def _fft(frame_windowed: np.ndarray) -> np.ndarray:
    return sfft.fft(frame_windowed.astype(np.complex64))


def _ifft_unnormalised(spec: np.ndarray) -> np.ndarray:
    n = spec.shape[-1]
    return (sfft.ifft(spec.astype(np.complex64)) * np.float32(n)).astype(np.complex64)


def _norm_sqr(c: np.ndarray) -> np.ndarray:
    return (c.real.astype(np.float32) ** 2 + c.imag.astype(np.float32) ** 2).astype(np.float32)


def _chunk_powers(samples: np.ndarray, frame_size: int) -> np.ndarray:
    n = len(samples) // frame_size
    w = hann_window(frame_size)
    if n == 0:
        return np.zeros((0, frame_size), np.float32)
    fr = samples[: n * frame_size].reshape(n, frame_size).astype(np.float32) * w
    return _norm_sqr(_fft(fr))


def analyze_noise_characteristics(samples: np.ndarray, frame_size: int):
    p = _chunk_powers(samples, frame_size)
    var = np.float32(0.0)
    for k in range(1, len(p)):
        var = np.float32(var + np.float32(np.sum((p[k] - p[k - 1]) ** 2, dtype=np.float32) / np.float32(frame_size)))
    nv = np.float32(var / np.float32(len(samples)))
    if nv < 0.1:
        return STATIONARY, float(nv)
    if nv > 0.5:
        return NON_STATIONARY, float(nv)
    return MIXED, float(nv)


def estimate_noise_spectrum(samples: np.ndarray, frame_size: int) -> np.ndarray:
    p = _chunk_powers(samples, frame_size)[:20]
    out = np.zeros(frame_size, np.float32)
    for row in p:
        out = (out + row / np.float32(20)).astype(np.float32)
    return out


def estimate_signal_spectrum(samples: np.ndarray, frame_size: int) -> np.ndarray:
    p = _chunk_powers(samples, frame_size)
    nf = np.float32(len(samples) // frame_size)
    out = np.zeros(frame_size, np.float32)
    for row in p:
        out = (out + row / nf).astype(np.float32)
    return out


def _frames(samples: np.ndarray, frame_size: int, step: int) -> np.ndarray:
    n = len(samples)
    if n < frame_size:
        raise ValueError("denoise: fewer samples than one frame (the reference panics in overlap_add on frames[0])")
    idx = np.arange(0, n - frame_size + 1, step)
    return np.stack([samples[i : i + frame_size] for i in idx]).astype(np.float32)


def overlap_add(frames_c: np.ndarray, output_len: int, step: int) -> np.ndarray:
    frame_size = frames_c.shape[1]
    w = hann_window(frame_size)
    out = np.zeros(output_len, np.float32)
    norm = np.zeros(output_len, np.float32)
    for i, fr in enumerate(frames_c):
        s = i * step
        m = min(frame_size, output_len - s)
        if m <= 0:
            continue
        out[s : s + m] += (fr.real[:m].astype(np.float32) * w[:m]).astype(np.float32)
        norm[s : s + m] += (w[:m] * w[:m]).astype(np.float32)
    ok = norm > np.float32(1e-10)
    out[ok] = (out[ok] / norm[ok]) * np.float32(10.0)
    return out


def spectral_subtraction(samples, frame_size, overlap, strength):
    step = int(np.float32(frame_size) * (np.float32(1.0) - np.float32(overlap)))
    noise = estimate_noise_spectrum(samples, frame_size)
    w = hann_window(frame_size)
    spec = _fft(_frames(samples, frame_size, step) * w)
    power = _norm_sqr(spec)
    i = np.arange(frame_size, dtype=np.float32)
    freq_factor = np.minimum(i / np.float32(frame_size), np.float32(1.0)).astype(np.float32)
    freq_strength = (np.float32(strength) * (np.float32(1.0) - np.float32(0.3) * freq_factor)).astype(np.float32)
    ratio = (noise / (power + np.float32(1e-6))).astype(np.float32)
    g = np.sqrt(np.maximum(np.float32(1.0) - np.float32(1.0) * np.power(ratio, freq_strength, dtype=np.float32), np.float32(0.1)), dtype=np.float32)
    return overlap_add(_ifft_unnormalised(spec * g), len(samples), step)


def wiener_filter(samples, frame_size, overlap, strength):
    step = int(np.float32(frame_size) * (np.float32(1.0) - np.float32(overlap)))
    noise = estimate_noise_spectrum(samples, frame_size)
    signal = estimate_signal_spectrum(samples, frame_size)
    w = hann_window(frame_size)
    spec = _fft(_frames(samples, frame_size, step) * w)
    snr = (signal / (noise + np.float32(1e-6))).astype(np.float32)
    g = np.power(snr / (np.float32(1.0) + snr), np.float32(strength) * np.float32(0.7), dtype=np.float32)
    return overlap_add(_ifft_unnormalised(spec * g), len(samples), step)


def denoise_audio(samples: np.ndarray, config: DenoiseConfig = DenoiseConfig(), force_type=None):
    samples = np.ascontiguousarray(samples, np.float32)
    nt, nv = analyze_noise_characteristics(samples, config.frame_size)
    if force_type is not None:
        nt = force_type
    if nt == STATIONARY:
        out = spectral_subtraction(samples, config.frame_size, config.overlap, config.strength)
    elif nt == NON_STATIONARY:
        out = wiener_filter(samples, config.frame_size, config.overlap, config.strength)
    else:
        out = wiener_filter(spectral_subtraction(samples, config.frame_size, config.overlap, config.strength), config.frame_size, config.overlap,
                            config.strength)
    return out, nt, nv


def apply_noise_gate(samples: np.ndarray, noise_gate: float) -> np.ndarray:
    s = np.asarray(samples, np.float32)
    return np.where(np.abs(s) < np.float32(noise_gate), np.float32(0.0), s).astype(np.float32)
