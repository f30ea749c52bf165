"""GPU parity for the STFT denoiser against the numpy restatement of src/audio/mod.rs:495-735."""
import numpy as np
import pytest

from speaksense_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(toy_ml_path):
    from speaksense_amd import binding
    e = binding.Engine(toy_ml_path, max_batch=1)
    yield e
    e.close()


def _signals():
    rng = np.random.default_rng(0)
    burst = (0.01 * synth.noise(4, 80000)).astype(np.float32)
    burst[30000:30400] += 0.8 * rng.standard_normal(400).astype(np.float32)
    return {
        "grpc_chunk_speech": synth.speech_like(1, 80000),              # what the gRPC handler hands over: 5 s (asr.rs:13-18,196)
        "quiet_stationary": (0.004 * synth.noise(2, 80000)).astype(np.float32),
        "burst": burst,
        "full_30s": synth.speech_like(5),
        "one_frame": synth.speech_like(6, 2048),                       # the REST stream pre-processor's case (mod.rs:133-134)
        "ragged": synth.speech_like(7, 2048 * 3 + 777),
    }


def _conditioned_err(got, ref, n, step=512, fs=2048):
    """Overlap-add divides by sum(w^2), which is ~1e-8 for the first/last samples of the signal (one covering frame, Hann -> 0):
    there the reference's output is the FFT round-off amplified by up to 1/w.  Parity is therefore measured on the numerator the
    kernels actually compute: |d out| * sum(w^2) / sum(w), i.e. in units of the inverse-FFT output, relative to its scale."""
    from oracle import denoise_oracle as d
    w = d.hann_window(fs).astype(np.float64)
    w1 = np.zeros(n); w2 = np.zeros(n)
    for f in range((n - fs) // step + 1):
        w1[f * step:f * step + fs] += w
        w2[f * step:f * step + fs] += w * w
    ok = w2 > 1e-10
    num_err = np.abs(got.astype(np.float64) - ref)[ok] * w2[ok] / np.maximum(w1[ok], 1e-30) / 10.0
    num_ref = np.abs(ref.astype(np.float64))[ok] * w2[ok] / np.maximum(w1[ok], 1e-30) / 10.0
    return num_err.max() / num_ref.max()


@pytest.mark.parametrize("name", sorted(_signals()))
def test_denoise_matches_oracle(eng, name):
    from oracle import denoise_oracle as d
    pcm = _signals()[name]
    ref, nt_ref, nv_ref = d.denoise_audio(pcm)
    got, nt, nv, ms = eng.denoise_audio(pcm)
    assert nt == nt_ref, (nv, nv_ref)
    assert abs(nv - nv_ref) <= 1e-3 * max(abs(nv_ref), 1e-12)
    assert got.shape == ref.shape
    # f32 FFTs with different butterfly orders (radix-2 here, pocketfft in the oracle, rustfft in the reference): 2e-5 of full scale
    assert _conditioned_err(got, ref, len(pcm)) <= 2e-5


@pytest.mark.parametrize("force", [0, 1, 2])
def test_denoise_each_algorithm(eng, force):
    """Stationary -> spectral subtraction, NonStationary -> Wiener, Mixed -> both in sequence (mod.rs:510-522)."""
    from oracle import denoise_oracle as d
    pcm = synth.speech_like(11, 48000)
    ref, _, _ = d.denoise_audio(pcm, force_type=force)
    got, nt, _, _ = eng.denoise_audio(pcm, force_type=force)
    assert nt == force
    assert _conditioned_err(got, ref, len(pcm)) <= (2e-5 if force != 2 else 1e-4)   # Mixed chains two passes


def test_denoise_properties_and_errors(eng):
    from speaksense_amd import binding
    pcm = synth.speech_like(3, 80000)
    out, nt, nv, ms = eng.denoise_audio(pcm)
    # unnormalised inverse FFT (x2048) and the hard-coded x10 gain are part of the reference's behaviour
    mid = slice(4096, 70000)
    ratio = np.abs(out[mid]).mean() / np.abs(pcm[mid]).mean()
    assert 2048 * 10 * 0.2 < ratio < 2048 * 10 * 1.2
    # samples past the last full frame get no contribution
    n_frames = (len(pcm) - 2048) // 512 + 1
    assert np.all(out[(n_frames - 1) * 512 + 2048:] == 0)
    with pytest.raises(binding.SpeakSenseError) as e:
        eng.denoise_audio(synth.speech_like(1, 2047))
    assert e.value.code == -1
    with pytest.raises(binding.SpeakSenseError) as e:
        eng.denoise_audio(pcm, binding.DenoiseConfig(1024, 0.75, 0.2, 0.003, 1, 0.002))     # DenoiseConfig { frame_size: 1024, .. }
    assert e.value.code == -9
