"""GPU: `suppress_non_speech_tokens` and `max_len` + `split_on_word` on the HIP engine, through the C ABI and through the whisper.h shim, against the
oracle (tests/test_oracle_wrap.py holds the oracle itself to an independent restatement; the reference leaves all three at their defaults,
/root/reference/src/asr/whisper.rs:156,161,167)."""
import ctypes as C
import os

import numpy as np
import pytest

from speaksense_amd import ggml_io, synth
from conftest import report
from test_oracle_wrap import NON_SPEECH, planted_model, py_wrap, unwrapped_segments

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planted(model_dir):
    return planted_model(model_dir)


@pytest.fixture(scope="module")
def eng(planted):
    from speaksense_amd import binding
    e = binding.Engine(planted[0], dtype=binding.DTYPE_F16, max_batch=4)
    yield e
    e.close()


def _transcribe(eng, pcm, P):
    """Session.transcribe + the per-segment token data (ids, token-level t0 / t1) in the oracle binding's shape."""
    ses = eng.new_session()
    got = ses.transcribe(pcm, P)
    for s, tt in zip(got["segments"], ses.token_times()):
        s["token_times"] = tt
    return got


def test_rules_kernel_masks_the_non_speech_ids(planted, eng):
    """ss_process_logits_row under the flag: the processed row is -inf at exactly the ids the oracle masks -- the planted list on top of the usual rules --
    and the log-probabilities of everything else agree; without the flag nothing changed."""
    from speaksense_amd import binding
    from oracle import binding as orc
    om = orc.OracleModel(planted[0])
    st = om.new_state(orc.MODE_F32)
    raw = np.random.default_rng(5).standard_normal(om.n_vocab).astype(np.float32)
    n_bits = 0
    for hist, has_ts in (([], False), ([om.beg + 3, 700, 701], True), ([om.beg, 900, om.beg + 10, om.beg + 10], True), ([om.beg + 7], True)):
        for flag in (0, 1):
            _, want, _ = st.process_logits(raw, hist, has_ts, 0, orc.default_params(language="en", suppress_non_speech_tokens=flag))
            got = eng.process_logits(raw, hist, has_ts, 0, binding.default_params(language="en", suppress_non_speech_tokens=flag), want_row=True)["logprobs"]
            assert np.array_equal(np.isneginf(got), np.isneginf(want)), (hist, flag)
            live = ~np.isneginf(want)
            assert np.abs(got[live] - want[live]).max() < 2e-5
            n_bits += om.n_vocab
        assert all(np.isneginf(got[i]) for i in NON_SPEECH | set(planted[1]))
    report(f"non-speech suppression on the device rules kernel: {n_bits} mask bits equal to the oracle's with and without the flag")
    om.close()


@pytest.mark.parametrize("max_len,split_on_word,ns", [(20, 1, 0), (20, 0, 1), (7, 1, 1)])
def test_engine_wrapped_segments_match_oracle(planted, eng, max_len, split_on_word, ns):
    from speaksense_amd import binding
    from oracle import binding as orc
    from test_gpu_parity import GAP_TOL_F16, check_against_oracle
    om = orc.OracleModel(planted[0])
    _, _, strs, _ = ggml_io.read_model(planted[0])
    n_same, worst, n_pieces = 0, 0.0, 0
    for seed in (11, 12):
        pcm = synth.speech_like(seed)
        kw = dict(language="en", temperature_inc=0.0, max_len=max_len, split_on_word=split_on_word, suppress_non_speech_tokens=ns)
        got = _transcribe(eng, pcm, binding.default_params(**kw))
        same, w = check_against_oracle(got, om, orc, orc.MODE_GGML_F16, pcm, orc.default_params(**kw), f"wrap {max_len}/{split_on_word}/{ns} seed {seed}", GAP_TOL_F16)
        n_same += same; worst = max(worst, w)
        if ns:
            assert not (set(int(t) for t in got["trace"]) & (NON_SPEECH | set(planted[1])))
        # the engine's own unwrapped run + the Python restatement of whisper_wrap_segment = the engine's wrapped run, piece by piece
        plain = _transcribe(eng, pcm, binding.default_params(**dict(kw, max_len=0)))
        assert list(plain["tokens"]) == list(got["tokens"])
        want = [p for seg in unwrapped_segments(plain) for p in py_wrap(seg, strs, om.eot, max_len, split_on_word)]
        have = [(s["t0"], s["t1"], s["text"], len(s["token_times"]["ids"])) for s in got["segments"]]
        assert have == want
        assert len(have) > len(plain["segments"])
        n_pieces += len(have)
    report(f"max_len {max_len}, split_on_word {split_on_word}, suppress_non_speech_tokens {ns}: {n_same}/2 chunks identical to the oracle (the rest proven near ties, "
           f"largest margin {worst:.4f}); {n_pieces} wrapped segments equal to whisper_wrap_segment applied to the unwrapped run")
    om.close()


def test_whisper_h_shim_honours_max_len_and_non_speech(planted, eng, monkeypatch):
    """whisper_full_with_state through include/whisper_compat.h with max_len / split_on_word / suppress_non_speech_tokens set (refused with -9 until
    round 5): segments, times, texts and per-segment token counts equal the C ABI's for the same parameters."""
    from speaksense_amd import binding
    from test_gpu_variants import WCtxParams, WFullParams          # the by-value whisper.h v1.5.4 mirrors of that file
    monkeypatch.setenv("SS_DTYPE", "f16")
    monkeypatch.setenv("SS_MAX_BATCH", "2")
    L = C.CDLL(binding.LIB_PATH)
    L.whisper_context_default_params.restype = WCtxParams
    L.whisper_init_from_file_with_params_no_state.restype = C.c_void_p
    L.whisper_init_from_file_with_params_no_state.argtypes = [C.c_char_p, WCtxParams]
    L.whisper_init_state.restype = C.c_void_p
    L.whisper_init_state.argtypes = [C.c_void_p]
    L.whisper_full_default_params.restype = WFullParams
    L.whisper_full_default_params.argtypes = [C.c_int]
    L.whisper_full_with_state.argtypes = [C.c_void_p, C.c_void_p, WFullParams, C.c_void_p, C.c_int]
    L.whisper_full_n_segments_from_state.argtypes = [C.c_void_p]
    L.whisper_full_n_tokens_from_state.argtypes = [C.c_void_p, C.c_int]
    L.whisper_full_get_segment_text_from_state.restype = C.c_char_p
    L.whisper_full_get_segment_text_from_state.argtypes = [C.c_void_p, C.c_int]
    for f in ("t0", "t1"):
        fn = getattr(L, f"whisper_full_get_segment_{f}_from_state")
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p, C.c_int]
    L.whisper_free_state.argtypes = [C.c_void_p]
    L.whisper_free.argtypes = [C.c_void_p]
    pcm = synth.speech_like(11)
    ctx = L.whisper_init_from_file_with_params_no_state(planted[0].encode(), L.whisper_context_default_params())
    assert ctx
    st = L.whisper_init_state(ctx)
    assert st
    p = L.whisper_full_default_params(0)
    p.language = b"en"; p.token_timestamps = True; p.max_len = 20; p.split_on_word = True; p.suppress_non_speech_tokens = True; p.temperature_inc = 0.0
    p.no_context = True
    rc = L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(C.c_void_p), len(pcm))
    assert rc == 0, rc
    want = _transcribe(eng, pcm, binding.default_params(language="en", temperature_inc=0.0, max_len=20, split_on_word=1, suppress_non_speech_tokens=1))
    n = L.whisper_full_n_segments_from_state(st)
    assert n == len(want["segments"]) and n > 3
    for i, s in enumerate(want["segments"]):
        assert L.whisper_full_get_segment_t0_from_state(st, i) == s["t0"] and L.whisper_full_get_segment_t1_from_state(st, i) == s["t1"]
        assert L.whisper_full_get_segment_text_from_state(st, i) == s["text"]
        assert L.whisper_full_n_tokens_from_state(st, i) == len(s["token_times"]["ids"])
    L.whisper_free_state(st)
    L.whisper_free(ctx)


def test_rules_kernel_single_character_symbols_match_hf_list(model_dir):
    """The device's non-speech bitmask against transformers' NON_SPEECH_TOKENS on the head of a GPT-2 byte-level vocabulary (ids 0 .. 93 = '!' .. '~'):
    the ids the flag masks below 94 are exactly HF's (tests/test_oracle_wrap.py holds the oracle to the same list)."""
    import os
    from speaksense_amd import binding
    from test_oracle_wrap import ascii_vocab_model
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_non_speech_ids.npz"))
    want = sorted(int(i) for i in g["en"] if i < 94)
    e = binding.Engine(ascii_vocab_model(model_dir), dtype=binding.DTYPE_F16, max_batch=2)
    raw = np.zeros(e.n_vocab, np.float32)
    beg = e.beg
    raw[beg:] = -30.0
    off = e.process_logits(raw, [beg + 3, 700], True, 0, binding.default_params(language="en"), want_row=True)["logprobs"]
    on = e.process_logits(raw, [beg + 3, 700], True, 0, binding.default_params(language="en", suppress_non_speech_tokens=1), want_row=True)["logprobs"]
    got = sorted(int(i) for i in np.nonzero(np.isneginf(on[:94]) & ~np.isneginf(off[:94]))[0])
    assert got == want
    report(f"non-speech bitmask on the device: the {len(got)} single ASCII characters it masks are transformers' NON_SPEECH_TOKENS below id 94")
    e.close()
