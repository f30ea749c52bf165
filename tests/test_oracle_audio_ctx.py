"""CPU: `whisper_full_params.audio_ctx` below n_audio_ctx (the reference passes 1500, /root/reference/src/asr/whisper.rs:144; a whisper-rs caller may
shorten it) in the oracle, held to HF transformers: tests/golden/hf_audio_ctx_golden.npz comes from HF models whose `max_source_positions` IS the
shortened context (tests/golden/make_golden.py audio_ctx_fixture) -- encoder rows including the last ones, top-16 logits of a teacher-forced
sequence over that many cross-attention keys, and HF `generate`'s first window."""
import os

import numpy as np
import pytest

from speaksense_amd import ggml_io, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_audio_ctx_golden.npz")


def audio_ctx_cases():
    g = np.load(GOLD)
    out = []
    for ci in range(int(g["n_cases"])):
        k = f"c{ci}"
        out.append(dict(preset=str(g[f"{k}_preset"]), seed=int(g[f"{k}_seed"]), audio=int(g[f"{k}_audio"]), audio_ctx=int(g[f"{k}_audio_ctx"]),
                        rows=g[f"{k}_rows"], enc=g[f"{k}_enc"], enc_absmax=float(g[f"{k}_enc_absmax"]), tokens=[int(t) for t in g[f"{k}_tokens"]],
                        n_prompt=int(g[f"{k}_n_prompt"]), topk=g[f"{k}_topk"], topv=g[f"{k}_topv"], logit_std=float(g[f"{k}_logit_std"]),
                        ids=[int(t) for t in g[f"{k}_ids"]], n0=int(g[f"{k}_n_window0"]),
                        seg=list(zip(g[f"{k}_seg_t0"].tolist(), g[f"{k}_seg_t1"].tolist()))))
    return out


def audio_ctx_case_model(c, model_dir):
    path = os.path.join(model_dir, f"{c['preset']}-natural-s{c['seed']}.bin")
    if not os.path.exists(path):
        ggml_io.write_model(path, c["preset"], seed=c["seed"], **ggml_io.NATURAL)
    return path


@pytest.mark.parametrize("ci", range(3))
def test_oracle_shortened_context_matches_hf(ci, model_dir):
    from oracle import binding as orc
    c = audio_ctx_cases()[ci]
    om = orc.OracleModel(audio_ctx_case_model(c, model_dir))
    pcm = synth.speech_like(c["audio"])
    A = c["audio_ctx"]
    enc = om.encode(om.log_mel(pcm), 0, orc.MODE_F32, audio_ctx=A)
    assert enc.shape == (A, om.n_audio_state)
    err = np.abs(enc[c["rows"]] - c["enc"]).max() / c["enc_absmax"]
    assert err < 2e-4, err                                   # f32 against f32: summation order only (measured 5 - 8e-5)
    if ci != 2:      # (the d = 1280 case skips the two full-context comparisons: CPU time)
        full = om.encode(om.log_mel(pcm), 0, orc.MODE_F32)
        assert np.abs(full[A - 1] - enc[A - 1]).max() / c["enc_absmax"] > 1e-2      # the cut is visible: the last rows are not a slice of the full pass
    st = om.new_state(orc.MODE_F32)
    st.set_encoder(enc)
    worst = 0.0
    n_p = c["n_prompt"]
    for pos in range(n_p - 1, len(c["tokens"])):
        lg = st.decode(c["tokens"][:n_p], 0) if pos == n_p - 1 else st.decode(c["tokens"][pos:pos + 1], pos)
        e = np.abs(lg[c["topk"][pos]] - c["topv"][pos]).max() / c["logit_std"]
        worst = max(worst, e)
        assert e < 5e-4, (pos, e)
        assert int(lg.argmax()) == int(c["topk"][pos][0])
    # sequence level: HF generate's first window on the shortened model
    ref = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language="en", temperature_inc=0.0, audio_ctx=A, duration_ms=30000))
    tr = [int(t) for t in ref["trace"]]
    assert tr[:c["n0"]] == c["ids"][:c["n0"]]
    assert [(s["t0"], s["t1"]) for s in ref["segments"]][:len(c["seg"])] == c["seg"]
    # and the context matters: the full-context run of the same audio picks differently somewhere in the window
    if ci == 0:
        other = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language="en", temperature_inc=0.0, duration_ms=30000))
        assert [int(t) for t in other["trace"]][:c["n0"]] != c["ids"][:c["n0"]]
    om.close()


def test_oracle_audio_ctx_edge_values(model_dir):
    """0 and n_audio_ctx are the full context; larger is whisper.cpp's -5; the state keeps the last call's context for the next call's language detection."""
    from oracle import binding as orc
    c = audio_ctx_cases()[1]
    om = orc.OracleModel(audio_ctx_case_model(c, model_dir))
    pcm = synth.speech_like(c["audio"])[:16000 * 8]
    a = om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", audio_ctx=0))
    b = om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", audio_ctx=1500))
    assert list(a["tokens"]) == list(b["tokens"])
    with pytest.raises(RuntimeError):
        om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", audio_ctx=1501))
    om.close()
