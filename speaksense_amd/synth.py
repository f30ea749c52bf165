"""Synthetic 16 kHz mono f32 audio: the bench and parity inputs.  numpy only."""
from __future__ import annotations

import numpy as np

SAMPLE_RATE = 16000
CHUNK_SAMPLES = SAMPLE_RATE * 30  # the REST chunker's BUFFER_SIZE (/root/reference/src/schedule/processors/transcribe.rs:105)


def speech_like(seed: int, n: int = CHUNK_SAMPLES) -> np.ndarray:
    """3-5 harmonics of a 90-250 Hz f0, 4 Hz syllabic AM, pink-ish noise at -30 dB, peak 0.5."""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / SAMPLE_RATE
    f0 = rng.uniform(90, 250)
    x = np.zeros(n)
    for h in range(1, int(rng.integers(3, 6)) + 1):
        x += rng.uniform(0.3, 1.0) / h * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 2 * np.pi))
    x *= 0.55 + 0.45 * np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi))
    w = rng.standard_normal(n)
    spec = np.fft.rfft(w)
    spec /= np.sqrt(np.maximum(np.arange(len(spec)), 1.0))
    pink = np.fft.irfft(spec, n)
    pink /= np.abs(pink).max() + 1e-12
    x = x / (np.abs(x).max() + 1e-12) + 10 ** (-30 / 20) * pink
    return (0.5 * x / np.abs(x).max()).astype(np.float32)


def noise(seed: int, n: int = CHUNK_SAMPLES, amp: float = 0.3) -> np.ndarray:
    return (amp * np.random.default_rng(seed).uniform(-1, 1, n)).astype(np.float32)


def silence(n: int = CHUNK_SAMPLES) -> np.ndarray:
    return np.zeros(n, np.float32)
