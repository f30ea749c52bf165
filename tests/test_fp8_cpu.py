"""CPU: the FP8-mode primitives.  There is no reference arithmetic for fp8 (whisper.cpp has none), so the anchor is the OCP e4m3 format itself
as torch.float8_e4m3fn implements it: the oracle's table-based rounding and the engine's host converter (the one that quantises weights at
load) must both agree with it bit for bit; the device conversion is then held to the oracle by the GPU tests."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _samples():
    rng = np.random.default_rng(7)
    parts = [rng.standard_normal(120000).astype(np.float32) * s for s in (1e-3, 0.02, 1.0, 30.0, 200.0)]
    edge = np.array([0.0, -0.0, 448.0, -448.0, 447.9, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10, 2.0 ** -6, 0.97 * 2.0 ** -6, 0.0175, 240.0, 232.0, 248.0], np.float32)
    halfway = np.array([(a + b) / 2 for a, b in zip(np.arange(16, 32), np.arange(17, 33))], np.float32)       # exact ties between neighbouring codes
    return np.clip(np.concatenate(parts + [edge, halfway, -halfway]), -448.0, 448.0)


def test_oracle_e4m3_rounding_is_ocp_e4m3fn():
    from oracle import binding as orc
    x = _samples()
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).to(torch.float32).numpy()
    got = orc.e4m3_round(x)
    assert np.array_equal(got, ref)
    assert orc.e4m3_round(np.array([1e6, -1e6], np.float32)).tolist() == [448.0, -448.0]      # saturating


def test_engine_host_converter_is_ocp_e4m3fn():
    from speaksense_amd import binding
    x = _samples()
    ref = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    got = binding.e4m3_from_f32(x)
    same = got == ref
    zero = (x == 0)                       # +-0: the sign bit of a zero code carries no value
    assert np.all(same | zero)
    assert binding.e4m3_from_f32(np.array([1e6, -1e6, 464.0], np.float32)).tolist() == [0x7e, 0xfe, 0x7e]


def test_block_exponent_is_smallest_power_of_two_that_fits():
    from oracle import binding as orc
    for amax in (448.0, 449.0, 1.0, 3.5, 3.51, 1e-3, 7.0, 0.4375, 0.43751, 57344.0, 1e-30):
        e = orc.e8m0_exponent(amax)
        s = 2.0 ** (e - 127)
        if e > 1:
            assert amax / s <= 448.0 and amax / (s / 2) > 448.0, amax
    assert orc.e8m0_exponent(0.0) == 1 and orc.e8m0_exponent(448.0) == 127 and orc.e8m0_exponent(448.0 * 2 ** 100) == 227


def test_quantize_rows_properties():
    """Per (row, 64 columns): error below half an e4m3 step of each element's binade, idempotent, the block maximum survives within one step,
    blocks are independent."""
    from oracle import binding as orc
    rng = np.random.default_rng(3)
    a = (rng.standard_normal((37, 256)) * np.exp(rng.uniform(-6, 6, (37, 4)).repeat(64, axis=1))).astype(np.float32)
    q = orc.quantize_rows_f8(a)
    assert np.array_equal(orc.quantize_rows_f8(q), q)
    blk_max = np.abs(a).reshape(37, 4, 64).max(axis=2).repeat(64, axis=1)
    err = np.abs(q - a)
    assert np.all(err <= 2.0 ** -4 * np.abs(a) + 2.0 ** -10 * blk_max / 224.0 + 1e-30)
    b = a.copy(); b[:, 64:128] *= 1000.0
    qb = orc.quantize_rows_f8(b)
    assert np.array_equal(qb[:, :64], q[:, :64]) and np.array_equal(qb[:, 128:], q[:, 128:])


def test_oracle_fp8_mode_is_close_to_f16_mode_but_not_equal(toy256_path):
    from oracle import binding as orc
    from speaksense_amd import synth
    om = orc.OracleModel(toy256_path)
    mel = om.log_mel(synth.speech_like(5, 16000 * 4))
    a = om.encode(mel, 0, orc.MODE_GGML_F16)
    b = om.encode(mel, 0, orc.MODE_FP8)
    s = np.abs(a).max()
    rms = float(np.sqrt(np.mean((a - b) ** 2))) / s
    assert 1e-4 < rms < 5e-2, rms
    om.close()
