"""GPU: the BENCHMARKED configuration under test -- the real large-v3 preset (32 + 32 layers, d = 1280, 20 heads, 128 mels, multilingual
vocabulary; BASELINE.json configs[2]: batch = 8 x 30 s chunks on one MI355X), seeded random weights in ggml format.

(i)  one window against the CPU oracle at FULL depth: encoder output (ggml-f16 arithmetic) and 8 teacher-forced decoder steps with the
     KV cache -- the same tolerances as the toy-model stage tests, so depth-dependent error growth or a kernel path only large shapes
     take (persistent multi-tile GEMM, grouped rasterisation, split-K plans at d = 1280) cannot hide;
(ii) properties at B = 8 that need no oracle: a device batch of 8 gives exactly the 8 single-chunk results, two runs are bit-identical,
     Mode F yields exactly 96 tokens per chunk -- what bench.py times.
The oracle costs ~1 min per window here (64 threads); everything else is seconds."""
import os
import sys

import numpy as np
import pytest

from speaksense_amd import synth
from conftest import SLOW, report, shared_oracle_model

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def large_v3_path():
    sys.path.insert(0, ROOT)
    import bench
    from speaksense_amd import ggml_io
    path = bench.model_path_for("large-v3")        # shared with bench.py: written once per box
    if not os.path.exists(path):
        ggml_io.write_model(path + ".tmp", "large-v3", seed=0)
        os.replace(path + ".tmp", path)
    return path


@pytest.fixture(scope="module")
def eng8(large_v3_path):
    from speaksense_amd import binding
    e = binding.Engine(large_v3_path, dtype=binding.DTYPE_F16, max_batch=8)
    yield e
    e.close()


def test_large_v3_full_depth_stages_vs_oracle(large_v3_path, eng8):
    from oracle import binding as orc
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    orc.set_thread_cap(min(64, ncpu))
    try:
        om = shared_oracle_model(large_v3_path)
        assert (om.n_audio_layer, om.n_text_layer, om.n_audio_state, om.n_mels, om.n_vocab) == (32, 32, 1280, 128, 51866)
        pcm = synth.speech_like(0)
        mel = om.log_mel(pcm)
        assert np.abs(eng8.log_mel(pcm) - mel).max() < 1e-4                 # north_star: log-mel within 1e-4
        ref = om.encode(mel, 0, orc.MODE_GGML_F16)
        got = eng8.encode(mel, 0)
        rel = np.abs(got - ref).max() / np.abs(ref).max()
        report(f"large-v3 encoder (32 layers): max|gpu - oracle| / max|oracle| = {rel:.2e}")
        assert rel < 4e-3, rel                                              # same bound as the 2-layer models (test_encoder_matches_oracle)
        # decoder: both sides start from the SAME encoder output (the oracle's), so this isolates the 32 decoder layers + cross-KV
        ost = om.new_state(orc.MODE_GGML_F16)
        ost.set_encoder(ref)
        ses = eng8.new_session()
        ses.set_encoder(ref)
        toks = [om.sot, om.sot + 1, om.transcribe, om.beg + 2, 1234, 777, 42, 31000, om.beg + 40, om.beg + 40, 9]
        r = ost.decode(toks[:3], 0)
        g = ses.decode(toks[:3], 0)
        sd = float(r.std())
        worst = np.abs(g - r).max() / sd
        assert int(g.argmax()) == int(r.argmax())
        for i in range(3, len(toks)):                                       # 8 teacher-forced steps on the KV cache
            r = ost.decode(toks[i:i + 1], i)
            g = ses.decode(toks[i:i + 1], i)
            e = np.abs(g - r).max() / sd
            worst = max(worst, e)
            assert e < 6e-3, f"step {i}: max|logits - oracle| / std = {e}"
            top2 = np.sort(r)[-2:]
            if top2[1] - top2[0] > 6e-3 * sd * 2:                           # argmax must agree unless the oracle's own top-2 is inside the noise
                assert int(g.argmax()) == int(r.argmax()), f"step {i}"
        report(f"large-v3 decoder (32 layers, 8 cached steps): worst max|logits - oracle| / std = {worst:.2e}")
        om.close()
    finally:
        orc.set_thread_cap(16)


def test_large_v3_batch8_properties(eng8):
    """What bench.py runs: 8 chunks, Mode F (96 greedy steps, EOT suppressed)."""
    from speaksense_amd import binding
    P = binding.default_params(language="en", fixed_steps=96)
    pcms = [synth.speech_like(cid) for cid in range(8)]
    ses = [eng8.new_session() for _ in pcms]
    a = eng8.transcribe_batch(ses, pcms, P)
    for r in a:
        assert len(r["tokens"]) == 96 and r["n_encode"] == 1 and r["n_windows"] == 1 and r["n_fail"] == 0
    b = eng8.transcribe_batch([eng8.new_session() for _ in pcms], pcms, P)     # run-to-run: no atomics, fixed reduction orders
    for x, y in zip(a, b):
        assert list(x["tokens"]) == list(y["tokens"]) and np.array_equal(x["plog"], y["plog"])
    for i in (0, 3, 7):                                                      # a row of the batch == the chunk alone
        s = eng8.new_session().transcribe(pcms[i], P)
        assert list(s["tokens"]) == list(a[i]["tokens"]), f"chunk {i}: batch of 8 differs from the single run"
        # the 3-token prompt pass runs as 3 rows of the fused step alone and as 24 rows of the 17..64-row kernels in the batch: other
        # accumulation orders in the K/V of the prompt positions, so log-probabilities agree to f16-noise, ids exactly
        np.testing.assert_allclose(s["plog"], a[i]["plog"], atol=2e-2)
    # chunks differ from one another (the audio matters: the cross-attention path is live at full depth)
    assert len({tuple(r["tokens"]) for r in a}) > 1


def test_large_v3_decode_vs_oracle_forced_replay(large_v3_path, eng8):
    """The whole path on one chunk at full depth (log-mel -> 32 encoder layers -> cross-KV -> prompt + 32 greedy steps with every logits rule):
    the device's token stream is replayed on the oracle step by step (oracle/binding.py `full(forced=...)`): every pick must be the oracle's
    argmax or within the f16 noise of it.  One oracle encoder pass + 35 oracle decoder steps."""
    from oracle import binding as orc
    from speaksense_amd import binding
    from test_gpu_parity import GAP_TOL_F16, check_against_oracle
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    orc.set_thread_cap(min(64, ncpu))
    try:
        om = shared_oracle_model(large_v3_path)
        pcm = synth.speech_like(1)
        got = eng8.new_session().transcribe(pcm, binding.default_params(language="en", fixed_steps=32))
        assert len(got["tokens"]) == 32
        check_against_oracle(got, om, orc, orc.MODE_GGML_F16, pcm, orc.default_params(language="en", fixed_steps=32), "large-v3", GAP_TOL_F16,
                             replay_only=True)
        om.close()
    finally:
        orc.set_thread_cap(16)


def test_large_v3_bf16_forced_replay(large_v3_path):
    """The dtype BASELINE.json names (bf16 MFMA) at full depth: one chunk end to end, the device's token stream replayed on the oracle's bf16
    mode (bf16-rounded weights and activations at every mat-mul input): every pick the oracle's argmax or within GAP_TOL_BF16 of it."""
    from oracle import binding as orc
    from speaksense_amd import binding
    from test_gpu_parity import GAP_TOL_BF16, check_against_oracle
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    orc.set_thread_cap(min(64, ncpu))
    try:
        om = shared_oracle_model(large_v3_path)
        eng = binding.Engine(large_v3_path, dtype=binding.DTYPE_BF16, max_batch=2)
        pcm = synth.speech_like(2)
        got = eng.new_session().transcribe(pcm, binding.default_params(language="en", fixed_steps=32))
        assert len(got["tokens"]) == 32
        _, gap = check_against_oracle(got, om, orc, orc.MODE_BF16, pcm, orc.default_params(language="en", fixed_steps=32), "large-v3 bf16", GAP_TOL_BF16,
                                      replay_only=True)
        report(f"large-v3 bf16 (32 + 32 layers, 32 greedy steps): largest near-tie margin in the forced replay {gap:.4f}")
        eng.close(); om.close()
    finally:
        orc.set_thread_cap(16)


def test_large_v3_full_depth_matches_hf_golden(eng8):
    """The HIP engine at FULL depth (32 + 32 layers) held DIRECTLY to HF transformers' Whisper on the same seeded large-v3 weights
    (tests/golden/hf_large_v3_golden.npz, tanh GELU; made by tests/golden/make_golden.py large_v3 in the build container): encoder rows, the top-16 logits
    of a prompt pass and 8 KV-cached steps, and the detected language -- no oracle in between.  Tolerances: the f16-operand bounds of the 2-layer
    golden tests (tests/test_gpu_golden.py), i.e. no allowance for depth."""
    from speaksense_amd import binding
    from test_gpu_golden import ENC_TOL_F16, LOGIT_TOL_F16
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_large_v3_golden.npz"))
    pcm = synth.speech_like(int(g["seed_audio"]))
    full = eng8.log_mel(pcm)
    mel = np.zeros_like(full)
    mel[:, :3000] = full[:, :3000].astype(np.float16).astype(np.float32)
    enc = eng8.encode(mel, 0)
    enc_err = float(np.abs(enc[g["enc_rows"]] - g["enc"]).max()) / float(g["enc_absmax"])
    assert enc_err < ENC_TOL_F16, enc_err
    ses = eng8.new_session()
    ses.set_encoder(enc)
    toks = [int(t) for t in g["tokens"]]
    n_prompt, sd, worst = int(g["n_prompt"]), float(g["logit_std"]), 0.0
    steps = [(n_prompt - 1, ses.decode(toks[:n_prompt], 0))] + [(i, ses.decode(toks[i:i + 1], i)) for i in range(n_prompt, len(toks))]
    for i, lg in steps:
        e = float(np.abs(lg[g["topk"][i]] - g["topv"][i]).max()) / sd
        worst = max(worst, e)
        assert e < LOGIT_TOL_F16, (i, e)
        if g["topv"][i][0] - g["topv"][i][1] > 2 * LOGIT_TOL_F16 * sd:
            assert int(lg.argmax()) == int(g["topk"][i][0]), i
    ses.close()
    det = eng8.new_session().transcribe(pcm, binding.default_params(language="en", detect_language=1))
    assert det["lang_id"] == int(g["lang_id"]) and float(g["lang_margin"]) > 1.0
    report(f"HIP engine (f16) vs HF golden at FULL depth (large-v3, 32 + 32 layers): encoder rows {enc_err:.1e} of absmax (tol {ENC_TOL_F16}), top-16 logits over "
           f"{len(steps)} steps {worst:.1e} sigma (tol {LOGIT_TOL_F16}), detected language {det['lang_id']} == HF's")
