"""CPU: whisper_process_logits / whisper_full_with_state semantics of the oracle on synthetic inputs."""
import numpy as np
import pytest

from oracle import binding as orc
from speaksense_amd import synth


@pytest.fixture(scope="module")
def om(toy_ml_path):
    m = orc.OracleModel(toy_ml_path)
    yield m
    m.close()


def test_special_token_ids(om, toy_en_path):
    # large-v3 style vocabulary (51866): SURVEY.md §8 a-8
    assert (om.eot, om.sot, om.translate, om.transcribe, om.solm, om.prev, om.nosp, om.not_, om.beg) == \
        (50257, 50258, 50359, 50360, 50361, 50362, 50363, 50364, 50365)
    en = orc.OracleModel(toy_en_path)
    assert (en.eot, en.sot, en.beg) == (50256, 50257, 50363)
    assert en.token_str(en.eot) == b"[_EOT_]" and en.token_str(en.beg + 7) == b"[_TT_7]"
    en.close()


def test_logits_rules(om):
    rng = np.random.default_rng(0)
    st = om.new_state()
    P = orc.default_params()
    raw = (5 * rng.standard_normal(om.n_vocab)).astype(np.float32)
    # initial step: EOT and " " suppressed, timestamps > 1.0 s (tid 50) masked
    r = raw.copy(); r[om.eot] = 100.0
    tid, lp, _ = st.process_logits(r, [], False, 3000, P)
    assert tid != om.eot and lp[om.eot] == -np.inf
    assert np.all(lp[om.beg + 51:] == -np.inf) and lp[om.beg + 50] > -np.inf
    # special/task/language tokens are never sampled
    for t in (om.sot, om.nosp, om.not_, om.translate, om.transcribe, om.prev, om.solm, om.sot + 1, om.sot + 100):
        assert lp[t] == -np.inf
    # after a lone timestamp: text is masked (next must be timestamp or EOT)
    tid, lp, _ = st.process_logits(raw, [100, om.beg + 20], True, 40, P)
    assert np.all(lp[:om.eot] == -np.inf) and tid >= om.eot
    # ... and timestamps may not decrease
    assert np.all(lp[om.beg:om.beg + 20] == -np.inf)
    # after a timestamp pair: no third timestamp
    tid, lp, _ = st.process_logits(raw, [om.beg + 20, om.beg + 20], True, 40, P)
    assert np.all(lp[om.beg:] == -np.inf) and tid < om.beg
    # if timestamps carry more probability mass than the best text token, a timestamp is forced
    r = raw.copy(); r[om.beg:] = r[:om.beg].max() - 1.0
    tid, lp, o5 = st.process_logits(r, [5, 6, 7], False, 3000, P)
    assert tid >= om.beg and np.all(lp[:om.beg] == -np.inf)
    assert int(o5[2]) == tid
    st.close()


def test_full_structure(om):
    st = om.new_state(orc.MODE_GGML_F16)
    r = st.full(synth.speech_like(3), orc.default_params(language="en", temperature_inc=0.0))
    assert r["n_encode"] >= 1 and len(r["tokens"]) > 0
    t_prev = -10**9
    for s in r["segments"]:
        assert s["t1"] >= s["t0"] and s["t0"] >= t_prev - 1 and len(s["text"]) > 0
        t_prev = s["t0"]
    # < 1 s of audio: nothing is decoded (whisper.cpp issue #39 rule)
    r = st.full(synth.speech_like(3, 8000), orc.default_params(language="en"))
    assert r["segments"] == [] and r["n_encode"] == 0
    # unknown language on a multilingual model
    with pytest.raises(RuntimeError):
        st.full(synth.speech_like(3, 32000), orc.default_params(language="xx"))
    # Mode F: one window, exactly N steps
    r = st.full(synth.speech_like(4), orc.default_params(language="en", fixed_steps=12))
    assert len(r["tokens"]) == 12 and r["n_encode"] == 1 and om.eot not in list(r["tokens"])
    st.close()


@pytest.mark.parametrize("topo", ["per_decoder", "rng_state"])
def test_trace_replay_is_exact_on_the_oracle_itself(tmp_path, topo):
    """The test hook the sampled-attempt parity rests on (oracle `full(trace=...)`): a chunk that walks the temperature ladder, replayed on a fresh
    state from its own trace, must consume every entry, recompute each uniform from the generator state before the call, find it INSIDE the
    interval of the cumulative distribution that selects the traced id (gap exactly 0: the recomputed CDF is libstdc++'s) and reproduce the
    result; a trace with one sampled id replaced by its upper neighbour must show a positive gap at exactly that call."""
    import numpy as np
    from oracle import binding as orc
    from speaksense_amd import ggml_io, synth
    path = str(tmp_path / "toy.en.bin")
    ggml_io.write_model(path, "toy.en", seed=1)
    om = orc.OracleModel(path)
    compat = orc.COMPAT_RNG_STATE if topo == "rng_state" else 0
    pcm = synth.speech_like(5, 16000 * 30)
    P = orc.default_params(language="en")
    ref = om.new_state(orc.MODE_GGML_F16, compat=compat).full(pcm, P)
    assert ref["n_fail"] >= 1 and len(ref["trace"]) > len(ref["sampled"])          # failed attempts and losing decoders are in the trace
    rep = om.new_state(orc.MODE_GGML_F16, compat=compat).full(pcm, P, trace=ref["trace"])
    assert len(rep["trace_gap"]) == len(ref["trace"]) and int(rep["trace_kind"].sum()) > 100
    assert float(rep["trace_gap"].max()) == 0.0 and (rep["trace_best"] == ref["trace"]).all()
    assert list(rep["tokens"]) == list(ref["tokens"]) and rep["n_fail"] == ref["n_fail"]
    k = int(np.nonzero(rep["trace_kind"] == 1)[0][7])
    bad = ref["trace"].copy()
    bad[k] = bad[k] + 1 if bad[k] + 1 < om.n_vocab else bad[k] - 1
    rep2 = om.new_state(orc.MODE_GGML_F16, compat=compat).full(pcm, P, trace=bad)
    assert rep2["trace_gap"][k] > 0.0 and rep2["trace_best"][k] == ref["trace"][k] and float(rep2["trace_gap"][:k].max()) == 0.0
    om.close()


def test_rng_topology_semantics(tmp_path):
    """What the two generator topologies mean (ledger row 1).
    per_decoder (whisper.cpp >= 1.5.0, default): every best_of decoder owns std::mt19937(0); decoder 0's is seeded with the state and carried
    across calls, decoders 1.. are re-seeded by every call.  On a FRESH state all five generators are therefore in the same position, the
    decoders start from the same distribution (they copy decoder 0's after the prompt) and draw the same uniforms: their sampled ids coincide call for
    call, and from the second call on the same state only decoders 1..4 still coincide.  rng_state (<= 1.4.x): one generator, consecutive draws, the
    decoders differ.  Greedy (t = 0) attempts are identical under both."""
    import numpy as np
    from oracle import binding as orc
    from speaksense_amd import ggml_io, synth
    path = str(tmp_path / "toy.en.bin")
    ggml_io.write_model(path, "toy.en", seed=1)
    om = orc.OracleModel(path)
    pcm = synth.speech_like(5, 16000 * 30)
    P = orc.default_params(language="en")

    def sampled_groups(state, r):
        """sampled calls of the trace grouped per step: replaying a state's own trace yields the kind of every call"""
        rep = om.new_state(orc.MODE_GGML_F16, compat=state.compat)
        # the replay state must be where `state` was BEFORE the call: advance its carried generator by replaying the earlier calls
        for earlier in state._history:
            rep.full(pcm, P)
        out = rep.full(pcm, P, trace=r["trace"])
        assert float(out["trace_gap"].max()) == 0.0
        ids = np.asarray(r["trace"])[out["trace_kind"] == 1]
        return ids

    a = om.new_state(orc.MODE_GGML_F16); a._history = []
    r1 = a.full(pcm, P)
    assert r1["n_fail"] >= 1
    ids = sampled_groups(a, r1)
    assert len(ids) >= 50 and len(ids) % 5 == 0
    g = ids.reshape(-1, 5)
    assert (g == g[:, :1]).all(), "fresh state, per-decoder generators: the five decoders must sample the same ids"
    a._history.append(1)
    peek0 = a.rng_peek(0)
    r2 = a.full(pcm, P)                                   # same audio, same state: decoder 0's generator has moved on, 1..4 start from 0 again
    ids2 = sampled_groups(a, r2)
    k = min(len(ids), len(ids2)) // 5 * 5
    g2 = ids2[:k].reshape(-1, 5)
    assert (g2[:, 1:] == g2[:, 1:2]).all() and (g2[:, 0] != g2[:, 1]).any(), "second call: decoders 1..4 coincide, decoder 0 does not"
    assert (g2[0, 1:] == g[0, 1:]).all(), "decoders 1..4 are re-seeded by every call: their first draw repeats the first call's"
    fresh = om.new_state(orc.MODE_GGML_F16)
    assert fresh.rng_peek(0) != peek0 or len(ids) == 0     # the carried generator really advanced

    s = om.new_state(orc.MODE_GGML_F16, compat=orc.COMPAT_RNG_STATE); s._history = []
    q1 = s.full(pcm, P)
    assert q1["n_fail"] >= 1
    qs = sampled_groups(s, q1)
    gq = qs[: len(qs) // 5 * 5].reshape(-1, 5)
    assert (gq != gq[:, :1]).any(), "one shared generator: the decoders draw consecutive uniforms and differ"
    assert list(q1["trace"]) != list(r1["trace"]), "the two topologies must be distinguishable on a chunk that falls back"
    # greedy-only calls never touch a generator: identical under both
    Pg = orc.default_params(language="en", temperature_inc=0.0)
    assert list(om.new_state(orc.MODE_GGML_F16).full(pcm, Pg)["trace"]) == list(om.new_state(orc.MODE_GGML_F16, compat=orc.COMPAT_RNG_STATE).full(pcm, Pg)["trace"])
    om.close()


def test_natural_preset_behaves_like_a_transcriber_on_the_oracle(tmp_path):
    """`ggml_io.NATURAL` (the synthetic weights behind bench.py's `mode_n` and the full-depth distinct-streams test): under whisper.cpp's FULL rules -- the
    reference's parameters, ladder on -- the decoder must stay at temperature 0 (logprob and entropy checks pass), end with EOT or at the end of the
    audio after a number of tokens that differs between audios, emit increasing timestamp pairs, and give a different stream for every audio.
    Smallest shape the style supports (d = 256) here; tools/natural_preset_stats.py prints the same statistics for large-v3 on the GPU box."""
    import numpy as np
    from oracle import binding as orc
    from speaksense_amd import ggml_io, synth
    path = str(tmp_path / "toy256-natural.bin")
    ggml_io.write_model(path, "toy256", seed=0, **ggml_io.NATURAL)
    om = orc.OracleModel(path)
    res = [om.new_state(orc.MODE_GGML_F16).full(synth.speech_like(100 + i), orc.default_params(language="en")) for i in range(6)]
    n_win, n_fail = sum(r["n_encode"] for r in res), sum(r["n_fail"] for r in res)
    lens = [len(r["tokens"]) for r in res]
    assert n_fail <= 0.2 * n_win, (n_fail, n_win)
    assert len({tuple(int(t) for t in r["tokens"]) for r in res}) == 6
    assert max(lens) - min(lens) >= 10 and min(lens) >= 2, lens
    for r in res:
        ts = [int(t) - om.beg for t in r["tokens"] if t >= om.beg]
        assert ts and ts[0] == 0                                    # the first window opens at <|0.00|>
        assert all(s["t1"] >= s["t0"] for s in r["segments"])
    with pytest.raises(ValueError):
        ggml_io.write_model(str(tmp_path / "x.bin"), "toy", seed=0, **ggml_io.NATURAL)     # d = 128: no room for the control subspace
    om.close()
