"""GPU: the model shapes of /root/reference/script/download-ggml-model.sh:36-48 that no other test reaches, each as a 2-layer-per-stack model of the
real width (the "wide2" idea: every kernel configuration of the shape at a cost the CPU oracle can pay):

  medium2    d = 1024, 16 heads, 80 mels, the 51865-token vocabulary (medium / large-v2 era: 99 languages, special ids shifted by one)
  small2.en  d = 768, 12 heads (small.en, small.en-tdrz): English-only vocabulary, run with tdrz_enable so [_SOLM_] stays sampleable
  turbo41    large-v3-turbo's asymmetry: more encoder than decoder layers (4 / 1 here, 32 / 4 in the checkpoint)

Per shape: encoder output against the oracle, then whole chunks -- identical ids or a forced replay proving every divergence a near tie, with
identical windows, segments and timestamps (check_against_oracle).  medium2 additionally through the fp8 engine (1024 % 256 == 0)."""
import numpy as np
import pytest

from speaksense_amd import synth
from conftest import _model, report
from test_gpu_parity import GAP_TOL_F16, check_against_oracle
from test_gpu_fp8 import GAP_TOL_FP8

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import binding as o
    return o


@pytest.mark.parametrize("which", ["medium2", "small2.en", "turbo41"])
def test_shape_stages_and_full_path_f16(model_dir, orc, which):
    from speaksense_amd import binding, ggml_io
    path = _model(model_dir, which)
    hp = ggml_io.PRESETS[which]
    om = orc.OracleModel(path)
    eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=4)
    assert (eng.n_audio_state, eng.n_audio_head, eng.n_audio_layer, eng.n_text_layer, eng.n_vocab) == (hp.n_audio_state, hp.n_audio_head, hp.n_audio_layer,
                                                                                                         hp.n_text_layer, hp.n_vocab)
    assert (eng.eot, eng.sot, eng.transcribe, eng.solm, eng.beg) == (om.eot, om.sot, om.transcribe, om.solm, om.beg)
    pcm = synth.speech_like(5)
    mel = om.log_mel(pcm)
    assert np.abs(eng.log_mel(pcm) - mel).max() < 1e-4
    ref = om.encode(mel, 0, orc.MODE_GGML_F16)
    got = eng.encode(mel, 0)
    rel = float(np.abs(got - ref).max() / np.abs(ref).max())
    assert rel < 4e-3, rel
    tdrz = which == "small2.en"
    kw = dict(language="en", temperature_inc=0.0, tdrz_enable=1 if tdrz else 0)
    same = 0
    cases = [(3, 8), (4, 30), (6, 17)]
    # a device batch of all three chunks (different lengths, multi-window) AND each alone must give the same thing
    P = binding.default_params(**kw)
    pcms = [synth.speech_like(seed, 16000 * sec) for seed, sec in cases]
    batch = eng.transcribe_batch([eng.new_session() for _ in pcms], pcms, P)
    for (seed, sec), x, b in zip(cases, pcms, batch):
        g = eng.new_session().transcribe(x, P)
        assert list(g["tokens"]) == list(b["tokens"]), f"{which} seed {seed}: batch differs from the single run"
        ok, _ = check_against_oracle(g, om, orc, orc.MODE_GGML_F16, x, orc.default_params(**kw), f"{which} seed {seed}", GAP_TOL_F16)
        same += ok
        assert len(g["tokens"]) > 0
    report(f"{which} (d={hp.n_audio_state}, {hp.n_audio_head} heads, {hp.n_audio_layer}/{hp.n_text_layer} layers, vocab {hp.n_vocab}): encoder {rel:.2e} of max; "
           f"{same}/{len(cases)} chunks token-identical to the free-running oracle, the rest proven near ties")
    assert same >= 1
    eng.close(); om.close()


def test_medium_width_fp8(model_dir, orc):
    """d = 1024 through the e4m3 engine (k-step groups of 256, 16-head attention epilogue quantisation): one chunk against the FP8-mode oracle."""
    from speaksense_amd import binding
    path = _model(model_dir, "medium2")
    om = orc.OracleModel(path)
    eng = binding.Engine(path, dtype=binding.DTYPE_FP8, max_batch=2)
    pcm = synth.speech_like(4, 16000 * 20)
    kw = dict(language="en", temperature_inc=0.0)
    g = eng.new_session().transcribe(pcm, binding.default_params(**kw))
    assert len(g["tokens"]) > 0
    ok, gap = check_against_oracle(g, om, orc, orc.MODE_FP8, pcm, orc.default_params(**kw), "fp8 medium2", GAP_TOL_FP8, tid_slack_beg=om.beg)
    report(f"fp8 medium2: token-identical {ok}, largest proven near-tie margin {gap:.4f}")
    eng.close(); om.close()
