"""Host-side mirror of the reference's audio pre-stage entry points that have a GPU implementation here
(/root/reference/src/audio/mod.rs): `DenoiseConfig` (:41-61), `denoise_audio` (:507-523), `apply_noise_gate` (:495-500).
Same names and argument meaning; runs on the engine's GPU through the C ABI (`ss_denoise_audio`)."""
from __future__ import annotations

import numpy as np

from . import binding

STATIONARY, NON_STATIONARY, MIXED = 0, 1, 2


def DenoiseConfig(frame_size=2048, overlap=0.75, strength=0.2, noise_gate=0.003, enable_noise_reduction=True, threshold=0.002):
    return binding.DenoiseConfig(frame_size, overlap, strength, noise_gate, int(enable_noise_reduction), threshold)


def denoise_audio(engine: binding.Engine, samples, config=None) -> np.ndarray:
    """`denoise_audio(samples, &config)`; raises for fewer than frame_size samples (the reference panics there)."""
    out, _, _, _ = engine.denoise_audio(samples, config)
    return out


def apply_noise_gate(samples, noise_gate: float) -> np.ndarray:
    s = np.asarray(samples, np.float32)
    return np.where(np.abs(s) < np.float32(noise_gate), np.float32(0.0), s).astype(np.float32)
