"""CPU: properties of the denoiser oracle (numpy restatement of /root/reference/src/audio/mod.rs:495-735)."""
import numpy as np
import pytest

from oracle import denoise_oracle as d
from speaksense_amd import synth


def test_window_and_step():
    w = d.hann_window(2048)
    assert w[0] == 0.0 and abs(w[1023] - 1.0) < 1e-5 and abs(w[-1]) < 1e-6   # symmetric Hann, (size-1) in the denominator (mod.rs:503-505)
    assert int(np.float32(2048) * (np.float32(1.0) - np.float32(0.75))) == 512


def test_noise_type_thresholds():
    assert d.analyze_noise_characteristics(np.zeros(80000, np.float32), 2048)[0] == d.STATIONARY
    assert d.analyze_noise_characteristics(synth.speech_like(1, 80000), 2048)[0] == d.NON_STATIONARY
    # fewer than two full frames: no spectral difference at all
    assert d.analyze_noise_characteristics(synth.speech_like(1, 3000), 2048) == (d.STATIONARY, 0.0)


def test_scale_and_support():
    pcm = synth.speech_like(2, 40000)
    out, nt, nv = d.denoise_audio(pcm)
    n_frames = (len(pcm) - 2048) // 512 + 1
    assert np.all(out[(n_frames - 1) * 512 + 2048:] == 0)          # nothing past the last full frame
    mid = slice(4096, 30000)
    ratio = np.abs(out[mid]).mean() / np.abs(pcm[mid]).mean()
    assert 2048 * 10 * 0.2 < ratio < 2048 * 10 * 1.2               # unnormalised inverse FFT x the hard-coded x10
    # linear in the input scale?  No: the gains depend on power ratios only, so scaling the input scales the output
    out2, _, _ = d.denoise_audio((0.5 * pcm).astype(np.float32), force_type=nt)
    assert np.allclose(out2[mid], 0.5 * out[mid], rtol=2e-3, atol=1e-3 * np.abs(out[mid]).max())


def test_too_short_raises():
    with pytest.raises(ValueError):
        d.denoise_audio(np.zeros(2047, np.float32))


def test_noise_gate():
    x = np.array([0.0029, -0.0031, 0.003, -0.0005], np.float32)
    assert np.array_equal(d.apply_noise_gate(x, 0.003), np.array([0.0, -0.0031, 0.003, 0.0], np.float32))
