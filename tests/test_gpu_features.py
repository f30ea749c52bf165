"""GPU: the whisper_full features a whisper-rs caller can reach beyond what the reference sets (VERDICT r1 "missing" #6): language
auto-detect (`language: None` -> whisper.rs:60-63 leaves the default, "auto" detects), initial_prompt / prompt_tokens, offset_ms / duration_ms,
n_max_text_ctx.  Each against the oracle's restatement of the same whisper.cpp branch: identical ids, segments, timestamps."""
import numpy as np
import pytest

from speaksense_amd import synth
from test_gpu_parity import _same_result

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import binding as o
    return o


@pytest.fixture(scope="module")
def pair(toy_ml_path, orc):
    from speaksense_amd import binding
    om = orc.OracleModel(toy_ml_path)
    eng = binding.Engine(toy_ml_path, dtype=binding.DTYPE_F16, max_batch=4)
    yield om, eng
    eng.close(); om.close()


def test_language_auto_detect(pair, orc):
    from speaksense_amd import binding
    om, eng = pair
    langs = set()
    for seed in (3, 4, 5, 6):
        pcm = synth.speech_like(seed, 16000 * 10)
        ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="auto", temperature_inc=0.0))
        got = eng.new_session().transcribe(pcm, binding.default_params(language="auto", temperature_inc=0.0))
        assert got["lang_id"] == ref["lang_id"] and 0 <= got["lang_id"] < 100
        _same_result(got, ref, f"auto seed {seed}")
        # the detected language drives the prompt: the explicit run with that language gives the same transcript
        code = binding.lang_code(got["lang_id"])
        exp = eng.new_session().transcribe(pcm, binding.default_params(language=code, temperature_inc=0.0))
        assert list(exp["tokens"]) == list(got["tokens"])
        langs.add(got["lang_id"])
        # detect only: no segments, language reported
        det = eng.new_session().transcribe(pcm, binding.default_params(language="en", detect_language=1))
        assert det["lang_id"] == ref["lang_id"] and det["segments"] == [] and len(det["tokens"]) == 0
    # "" behaves like "auto"; mixed batch: detection rows and plain rows share one device batch
    pcms = [synth.speech_like(20 + k, 16000 * 9) for k in range(3)]
    ses = [eng.new_session() for _ in pcms]
    ps = [binding.default_params(language=l, temperature_inc=0.0) for l in ("", "de", "auto")]
    tickets = [s.submit(x, p) for s, x, p in zip(ses, pcms, ps)]
    for s, t, x, l in zip(ses, tickets, pcms, ("", "de", "auto")):
        got = s.wait(t)
        ref = om.new_state(orc.MODE_GGML_F16).full(x, orc.default_params(language=l, temperature_inc=0.0))
        _same_result(got, ref, f"mixed batch language={l!r}")
        assert got["lang_id"] == ref["lang_id"]


def test_initial_prompt_and_prompt_tokens(pair, orc):
    from speaksense_amd import binding
    om, eng = pair
    text = b" " + om.token_str(1300) + om.token_str(2222) + b" " + om.token_str(901) + b"'s 12"
    toks = om.tokenize(text)
    assert eng.tokenize(text) == toks and len(toks) >= 4
    pcm = synth.speech_like(9, 16000 * 20)
    base = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0))
    a = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, initial_prompt=text))
    b = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, prompt_tokens=toks))
    ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="en", temperature_inc=0.0, initial_prompt=text))
    _same_result(a, ref, "initial_prompt")
    _same_result(b, ref, "prompt_tokens")
    assert list(a["tokens"]) != list(base["tokens"]), "the prompt did not condition the decoder"
    # n_max_text_ctx = 0 switches the conditioning off again
    c = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, initial_prompt=text, n_max_text_ctx=0))
    refc = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="en", temperature_inc=0.0, initial_prompt=text, n_max_text_ctx=0))
    _same_result(c, refc, "n_max_text_ctx=0")
    assert list(c["tokens"]) == list(base["tokens"])


@pytest.mark.parametrize("offset_ms,duration_ms", [(5000, 0), (0, 12000), (7000, 9000), (29500, 0)])
def test_offset_and_duration(pair, orc, offset_ms, duration_ms):
    from speaksense_amd import binding
    om, eng = pair
    pcm = synth.speech_like(13)
    kw = dict(language="en", temperature_inc=0.0, offset_ms=offset_ms, duration_ms=duration_ms)
    ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(**kw))
    got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
    _same_result(got, ref, f"offset {offset_ms} duration {duration_ms}")
    if offset_ms < 29000:
        assert got["segments"] and got["segments"][0]["t0"] >= offset_ms // 10
    else:
        assert got["segments"] == []      # less than 1 s left
