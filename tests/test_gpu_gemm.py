"""GPU: the tiled MFMA GEMM against a one-thread-per-output device reference at the large-v3 encoder shapes (ss_engine_selftest_gemm).
The model-level parity tests run small models whose GEMMs are one tile per workgroup; the persistent multi-tile machinery (early
prologue with store-aware vmcnt, DMA by half the waves, grouped rasterisation, partial last tiles) only runs at these sizes."""
import pytest

pytestmark = pytest.mark.gpu

STORE, GELU, RES, STORE_F32 = 0, 1, 2, 6


@pytest.fixture(scope="module", params=["f16", "bf16"])
def eng(request, toy_ml_path):
    from speaksense_amd import binding
    e = binding.Engine(toy_ml_path, dtype=binding.DTYPE_F16 if request.param == "f16" else binding.DTYPE_BF16, max_batch=1)
    e.dtype_name = request.param
    yield e
    e.close()


@pytest.mark.parametrize("M,N,K,kind", [
    (12000, 5120, 1280, GELU),       # FC1 at batch 8: 940 tiles on 256 workgroups, partial last row tile, DMA4 + early prologue
    (12000, 5120, 1280, STORE),
    (12000, 3840, 1280, STORE),      # QKV
    (12000, 1280, 5120, RES),        # FC2: f32 residual in place, 160 k-steps
    (12000, 1280, 1280, RES),        # attention out-projection
    (12000, 5120, 1280, STORE_F32),
    (1500, 1280, 1280, STORE),       # one window: fewer tiles than workgroups
    (24000, 5120, 1280, GELU),       # batch 16: more rounds per workgroup
    (3000, 384, 1152, STORE),        # N not a multiple of 256: the 128 x 128 kernel
    (777, 256, 128, RES),            # small M: the 128 x 128 kernel, ragged rows
])
def test_gemm_matches_reference(eng, M, N, K, kind):
    err, ref = eng.selftest_gemm(M, N, K, kind)
    assert ref > 0.5
    f32_out = kind in (RES, STORE_F32)
    if f32_out:
        tol = 2e-5 * ref if eng.dtype_name == "f16" else 2e-5 * ref     # same operands, only the accumulation order differs
    else:
        tol = (1.2e-3 if eng.dtype_name == "f16" else 9e-3) * ref        # + one rounding of the output to f16 / bf16
    assert err <= tol, f"max |diff| {err} vs max |ref| {ref}"


# ---- the e4m3 GEMM on the MX-scaled MFMA (kernels_gemm_fp8.hip) ----
F8_STORE, F8_GELU, F8_RES, F8_STORE_F32 = 0, 1, 2, 5


@pytest.mark.parametrize("M,N,K,kind", [
    (12000, 5120, 1280, F8_STORE_F32),   # FC1 shape: 940 tiles on 256 workgroups, partial last row tile, every group kind of the k loop
    (12000, 5120, 1280, F8_GELU),        # e4m3 output + its own exponent bytes
    (12000, 2560, 1280, F8_STORE),       # QK projection -> T
    (12000, 1280, 5120, F8_RES),         # FC2: 80 k-steps, f32 residual in place
    (12000, 1280, 1280, F8_RES),         # attention out-projection
    (1500, 1280, 1280, F8_STORE),        # one window: fewer tiles than workgroups (no early prologue)
    (24000, 5120, 1280, F8_GELU),        # batch 16
    (777, 256, 256, F8_STORE_F32),       # K = 256: the single (last) k group only, ragged rows
    (3000, 512, 512, F8_RES),            # two k groups
])
def test_gemm_fp8_matches_reference(eng, M, N, K, kind):
    """Operands are exact e4m3 codes with power-of-two block exponents, so the only difference from the one-thread-per-output reference is the
    f32 accumulation order (and one rounding of the output to T / e4m3)."""
    err, ref, _ = eng.selftest_gemm_ex(M, N, K, kind, fp8=True)
    assert ref > 0.5
    if kind in (F8_RES, F8_STORE_F32):
        tol = 1e-4 * ref           # f32 accumulation order inside / across the 64-k MFMAs vs the sequential reference
    elif kind == F8_GELU:
        tol = 1e-4 * ref          # err is already the excess over the 2^-4 relative quantisation step of an e4m3 output
    else:
        tol = (1.2e-3 if eng.dtype_name == "f16" else 9e-3) * ref
    assert err <= tol, f"max |diff| {err} vs max |ref| {ref}"


def test_ab_reference_main_loops_match_reference(toy_ml_path):
    """The two A/B switches that select the PREVIOUS main loops -- `SS_GEMM_K64=0` (f16 / bf16 GEMM with 32-deep stages) and `SS_F8_K128=0` (e4m3 GEMM
    with 64-byte LDS rows) -- are read once per process, so they get a process of their own: the kernels they select must still pass the self-test
    (DESIGN.md section 7a promises that every switch selects a tested path)."""
    import os
    import subprocess
    import sys
    code = f"""
import sys
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from speaksense_amd import binding
e = binding.Engine({toy_ml_path!r}, dtype=binding.DTYPE_F16, max_batch=1)
for M, N, K, kind, tol in ((12000, 5120, 1280, 1, 1.2e-3), (12000, 1280, 5120, 2, 2e-5), (1500, 1280, 1280, 0, 1.2e-3)):
    err, ref = e.selftest_gemm(M, N, K, kind)
    assert ref > 0.5 and err <= tol * ref, ("f16", M, N, K, kind, err, ref)
for M, N, K, kind, tol in ((12000, 5120, 1280, 5, 1e-4), (12000, 5120, 1280, 1, 1e-4), (12000, 1280, 5120, 2, 1e-4), (777, 256, 256, 5, 1e-4)):
    err, ref, _ = e.selftest_gemm_ex(M, N, K, kind, fp8=True)
    assert ref > 0.5 and err <= tol * ref, ("e4m3", M, N, K, kind, err, ref)
e.close()
print("ok")
"""
    env = dict(os.environ, SS_GEMM_K64="0", SS_F8_K128="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("dtype", ["f16", "fp8"])
def test_fused_qkv_projection_switch_gives_the_same_bits(wide2_path, dtype):
    """`SS_VT_GEMM=0` (round-5 experiment, measured 1.4 % slower, profiles/r05_ah_vt_gemm_ab.txt): Q | K | V in one GEMM as plain rows and a V -> V^T pass
    through LDS tiles instead of the transposing GEMM epilogue.  The switch is read at engine creation, so each form gets a process of its own; the encoder
    output of one window must be identical bit for bit (the projection's summation order per element does not depend on the tiling of N), also with a
    shortened context (the transpose writes the zero key columns of the last 64-key tile itself)."""
    import hashlib
    import os
    import subprocess
    import sys
    code = f"""
import sys, hashlib
import numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
from speaksense_amd import binding, synth
e = binding.Engine({wide2_path!r}, dtype=binding.DTYPE_{dtype.upper()}, max_batch=2)
pcm = synth.speech_like(3)
mel = e.log_mel(pcm)
h = hashlib.sha256(e.encode(mel, 0).tobytes()).hexdigest()
r = e.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, audio_ctx=500))
print(h, hashlib.sha256(np.asarray(r["trace"], np.int32).tobytes()).hexdigest(), len(r["trace"]))
e.close()
"""
    outs = []
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SS_VT_GEMM=v), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1], outs
