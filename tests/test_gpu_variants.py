"""GPU parity, part 2: whisper_full parameter variants, multi-call context, long audio, threads, and the whisper.h-compatible shim."""
import ctypes as C
import threading

import numpy as np
import pytest

from speaksense_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import binding as o
    return o


@pytest.fixture(scope="module")
def eng(toy_ml_path):
    from speaksense_amd import binding
    e = binding.Engine(toy_ml_path, dtype=binding.DTYPE_F16, max_batch=4)
    yield e
    e.close()


@pytest.fixture(scope="module")
def om(toy_ml_path, orc):
    m = orc.OracleModel(toy_ml_path)
    yield m
    m.close()


def _same(got, ref, ctx):
    assert list(got["tokens"]) == list(ref["tokens"]), f"{ctx}: token ids differ"
    assert [(s["t0"], s["t1"], s["text"], s["speaker_turn_next"]) for s in got["segments"]] == \
        [(s["t0"], s["t1"], s["text"], s["speaker_turn_next"]) for s in ref["segments"]], ctx


VARIANTS = {
    "zh": dict(language="zh"),
    "ja_translate": dict(language="ja", translate=1),
    "no_timestamps": dict(language="en", no_timestamps=1),
    "single_segment": dict(language="en", single_segment=1),
    "max_tokens": dict(language="en", max_tokens=10),
    "tdrz": dict(language="en", tdrz_enable=1),
    "print_special": dict(language="en", print_special=1),
    "no_suppress_blank": dict(language="en", suppress_blank=0),
    "max_initial_ts_off": dict(language="en", max_initial_ts=0.0),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_param_variants_match_oracle(eng, om, orc, name):
    from speaksense_amd import binding
    kw = dict(VARIANTS[name], temperature_inc=0.0)
    for seed in (3, 5):
        pcm = synth.speech_like(seed)
        ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(**kw))
        got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
        _same(got, ref, f"{name} seed {seed}")


def test_context_carries_across_calls_when_no_context_is_false(eng, om, orc):
    """build_params sets no_context(false) (whisper.rs:159); without stream mode the previous call's text conditions the next."""
    from speaksense_amd import binding
    kw = dict(language="en", temperature_inc=0.0, no_context=0)
    ost = om.new_state(orc.MODE_GGML_F16)
    ses = eng.new_session()
    outs = []
    for seed in (3, 4, 5):
        pcm = synth.speech_like(seed, 16000 * 12)
        ref = ost.full(pcm, orc.default_params(**kw))
        got = ses.transcribe(pcm, binding.default_params(**kw))
        _same(got, ref, f"call with seed {seed}")
        outs.append(list(got["tokens"]))
    # and it really is context dependent: a fresh session gives a different continuation for the last chunk
    fresh = eng.new_session().transcribe(synth.speech_like(5, 16000 * 12), binding.default_params(**kw))
    assert list(fresh["tokens"]) != outs[-1]


@pytest.mark.parametrize("seconds", [1.0, 7.3, 29.99, 30.08, 47.0, 95.0])
def test_audio_lengths(eng, om, orc, seconds):
    from speaksense_amd import binding
    n = int(seconds * 16000)
    pcm = synth.speech_like(17, n)
    kw = dict(language="en", temperature_inc=0.0)
    ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(**kw))
    got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
    _same(got, ref, f"{seconds}s")
    assert got["n_encode"] == ref["n_encode"]
    if seconds > 60:
        assert got["n_encode"] >= 3


def test_edge_signals(eng, om, orc):
    from speaksense_amd import binding
    kw = dict(language="en", temperature_inc=0.0)
    cases = {"silence": synth.silence(), "noise": synth.noise(5), "clipped": np.clip(4 * synth.speech_like(9), -1, 1).astype(np.float32),
             "dc": np.full(480000, 0.25, np.float32), "tiny": (1e-6 * synth.speech_like(10)).astype(np.float32)}
    for name, pcm in cases.items():
        ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(**kw))
        got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
        _same(got, ref, name)


def test_threads_share_one_engine(eng):
    """One engine, many sessions, concurrent callers (the reference runs one state per gRPC stream on one context)."""
    from speaksense_amd import binding
    P = binding.default_params(language="en", temperature_inc=0.0)
    pcms = [synth.speech_like(60 + i, 16000 * 10) for i in range(8)]
    ref = [eng.new_session().transcribe(p, P) for p in pcms]
    out = [None] * len(pcms)

    def work(i):
        s = eng.new_session()
        out[i] = s.wait(s.submit(pcms[i], P))

    th = [threading.Thread(target=work, args=(i,)) for i in range(len(pcms))]
    [t.start() for t in th]
    [t.join() for t in th]
    for i, (a, b) in enumerate(zip(out, ref)):
        _same(a, b, f"thread {i}")


def test_unsupported_and_bad_arguments(eng):
    from speaksense_amd import binding
    with pytest.raises(binding.SpeakSenseError) as e:
        eng.new_session().transcribe(synth.speech_like(1, 32000), binding.default_params(language="en", audio_ctx=2000))
    assert e.value.code == -5      # whisper_full: audio_ctx > n_audio_ctx -> -5
    with pytest.raises(binding.SpeakSenseError) as e:
        eng.new_session().transcribe(synth.speech_like(1, 32000), binding.default_params(language="en", best_of=9))
    assert e.value.code == -1


# ------------------------------------------------------------------------------------------------
# whisper.h-compatible shim (include/whisper_compat.h), driven exactly as whisper-rs-sys would
# ------------------------------------------------------------------------------------------------
class WFullParams(C.Structure):
    _fields_ = [
        ("strategy", C.c_int), ("n_threads", C.c_int), ("n_max_text_ctx", C.c_int), ("offset_ms", C.c_int), ("duration_ms", C.c_int),
        ("translate", C.c_bool), ("no_context", C.c_bool), ("no_timestamps", C.c_bool), ("single_segment", C.c_bool), ("print_special", C.c_bool),
        ("print_progress", C.c_bool), ("print_realtime", C.c_bool), ("print_timestamps", C.c_bool), ("token_timestamps", C.c_bool),
        ("thold_pt", C.c_float), ("thold_ptsum", C.c_float), ("max_len", C.c_int), ("split_on_word", C.c_bool), ("max_tokens", C.c_int),
        ("speed_up", C.c_bool), ("debug_mode", C.c_bool), ("audio_ctx", C.c_int), ("tdrz_enable", C.c_bool), ("initial_prompt", C.c_char_p),
        ("prompt_tokens", C.c_void_p), ("prompt_n_tokens", C.c_int), ("language", C.c_char_p), ("detect_language", C.c_bool),
        ("suppress_blank", C.c_bool), ("suppress_non_speech_tokens", C.c_bool), ("temperature", C.c_float), ("max_initial_ts", C.c_float),
        ("length_penalty", C.c_float), ("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float),
        ("no_speech_thold", C.c_float), ("greedy_best_of", C.c_int), ("beam_size", C.c_int), ("beam_patience", C.c_float),
        ("new_segment_callback", C.c_void_p), ("new_segment_callback_user_data", C.c_void_p), ("progress_callback", C.c_void_p),
        ("progress_callback_user_data", C.c_void_p), ("encoder_begin_callback", C.c_void_p), ("encoder_begin_callback_user_data", C.c_void_p),
        ("abort_callback", C.c_void_p), ("abort_callback_user_data", C.c_void_p), ("logits_filter_callback", C.c_void_p),
        ("logits_filter_callback_user_data", C.c_void_p), ("grammar_rules", C.c_void_p), ("n_grammar_rules", C.c_size_t),
        ("i_start_rule", C.c_size_t), ("grammar_penalty", C.c_float),
    ]


class WCtxParams(C.Structure):
    _fields_ = [("use_gpu", C.c_bool)]


def test_whisper_h_shim_matches_native_api(toy_ml_path, eng, monkeypatch, capfd):
    """The call sequence of /root/reference/src/asr/whisper.rs (new -> create_state -> build_params -> full -> segment getters)."""
    from speaksense_amd import binding
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
    L.whisper_full_get_segment_text_from_state.restype = C.c_char_p
    L.whisper_full_get_segment_text_from_state.argtypes = [C.c_void_p, C.c_int]
    for f in ("t0", "t1"):
        fn = getattr(L, f"whisper_full_get_segment_{f}_from_state")
        fn.restype = C.c_int64
        fn.argtypes = [C.c_void_p, C.c_int]
    L.whisper_full_get_segment_speaker_turn_next_from_state.restype = C.c_bool
    L.whisper_full_get_segment_speaker_turn_next_from_state.argtypes = [C.c_void_p, C.c_int]
    L.whisper_free_state.argtypes = [C.c_void_p]
    L.whisper_free.argtypes = [C.c_void_p]

    ctx = L.whisper_init_from_file_with_params_no_state(toy_ml_path.encode(), L.whisper_context_default_params())
    assert ctx
    st = L.whisper_init_state(ctx)
    assert st
    p = L.whisper_full_default_params(0)   # WHISPER_SAMPLING_GREEDY
    assert p.greedy_best_of == 5 and p.no_context and abs(p.temperature_inc - 0.2) < 1e-7 and p.language == b"en"
    # build_params (whisper.rs:131-173) + stream mode (65-69) + language
    p.n_threads = 16; p.audio_ctx = 1500; p.print_realtime = True; p.print_timestamps = True; p.single_segment = False
    p.print_progress = True; p.print_special = False; p.suppress_non_speech_tokens = False; p.max_initial_ts = 1.0
    p.no_context = False; p.token_timestamps = True; p.split_on_word = True; p.temperature = 0.0; p.entropy_thold = 2.4
    p.logprob_thold = -1.0; p.no_speech_thold = 0.6; p.max_len = 0; p.max_tokens = 0; p.speed_up = False
    p.thold_pt = 0.01; p.thold_ptsum = 0.01; p.length_penalty = -1.0
    p.language = b"zh"
    p.single_segment = False; p.no_context = True; p.audio_ctx = 0
    p.temperature_inc = 0.0   # keep this comparison deterministic (no sampled fallback)
    pcm = synth.speech_like(21)
    capfd.readouterr()
    rc = L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(C.c_void_p), len(pcm))
    assert rc == 0
    printed = capfd.readouterr().out
    ref = eng.new_session().transcribe(pcm, binding.default_params(language="zh", temperature_inc=0.0))
    n = L.whisper_full_n_segments_from_state(st)
    assert n == len(ref["segments"]) and n > 0
    # print_realtime + print_timestamps (the reference sets both, whisper.rs:145-150): one line per segment in whisper.cpp's format
    def ts(t):
        ms = t * 10
        return "%02d:%02d:%02d.%03d" % (ms // 3600000, ms // 60000 % 60, ms // 1000 % 60, ms % 1000)
    want = "".join("[%s --> %s]  %s\n" % (ts(s["t0"]), ts(s["t1"]), s["text"].decode("utf-8", "replace")) for s in ref["segments"])
    assert printed.encode("utf-8", "replace").decode("utf-8", "replace") == want, (printed[:200], want[:200])
    p.print_timestamps = False
    assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(C.c_void_p), len(pcm)) == 0
    assert capfd.readouterr().out == "".join(s["text"].decode("utf-8", "replace") for s in ref["segments"])
    p.print_realtime = False
    assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(C.c_void_p), len(pcm)) == 0
    assert capfd.readouterr().out == ""
    for i, s in enumerate(ref["segments"]):
        assert L.whisper_full_get_segment_text_from_state(st, i) == s["text"]
        assert L.whisper_full_get_segment_t0_from_state(st, i) == s["t0"]
        assert L.whisper_full_get_segment_t1_from_state(st, i) == s["t1"]
        assert L.whisper_full_get_segment_speaker_turn_next_from_state(st, i) == s["speaker_turn_next"]
    # features this path does not implement are refused, not ignored
    q = L.whisper_full_default_params(1)   # beam search
    q.language = b"en"
    assert L.whisper_full_with_state(ctx, st, q, pcm.ctypes.data_as(C.c_void_p), len(pcm)) == -9
    L.whisper_free_state(st)
    L.whisper_free(ctx)


# ------------------------------------------------------------------------------------------------
# the host-side mirror of the reference's AsrEngine (speaksense_amd/asr.py) end to end
# ------------------------------------------------------------------------------------------------
def test_whisper_asr_mirror_end_to_end(toy_ml_path, om, orc):
    """Mirrors the reference's only test that crosses this boundary (test_transcribe_processor,
    /root/reference/src/schedule/processors/transcribe.rs:248-304: asserts non-empty text and segments), plus what that test
    could not check: the exact segments, the stream-mode last-segment rule and the punctuation pass."""
    from speaksense_amd import asr
    eng = asr.WhisperAsr(toy_ml_path, max_batch=4)
    pcm = synth.speech_like(3)
    p = asr.AsrParams(language="zh", stream_mode=True)     # what both reference callers set (transcribe.rs:66-70, asr.rs:154-157)
    st = eng.create_state()
    res = eng.transcribe_with_state(st, pcm, p)
    assert len(res.full_text) > 0 and len(res.segments) == 1           # stream mode keeps only the last segment (whisper.rs:102-111)
    # same chunk without stream mode: every (non-promotional) segment, texts post-processed by add_punctuation
    p2 = asr.AsrParams(language="zh", stream_mode=False)
    res2 = eng.transcribe(pcm, p2)
    ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="zh", no_context=0))
    assert ref["n_fail"] == 0, "fixture drifted: the oracle fell back to sampling; re-pick the seed with tools/find_nofallback_seeds.py"
    kept = [s for s in ref["segments"] if not asr.is_promotional_text(s["text"].decode())]
    assert [s.text for s in res2.segments] == [asr.add_punctuation(s["text"].decode()) for s in kept]
    assert [(s.start, s.end) for s in res2.segments] == [(float(s["t0"]), float(s["t1"])) for s in kept]
    assert res2.full_text == "".join(s.text for s in res2.segments)
    assert res.segments[0].text == res.full_text
    # batched form: identical per-chunk results
    pcms = [synth.speech_like(70 + i, 16000 * 8) for i in range(3)]
    single = [eng.transcribe(x, p) for x in pcms]
    many = eng.transcribe_many([eng.create_state() for _ in pcms], pcms, p)
    assert [r.full_text for r in many] == [r.full_text for r in single]
    eng.engine.close()


def test_two_engines_in_one_process(toy_ml_path, toy_en_path):
    """A service process may hold several engines (one per GPU; here two models on one GPU): kernel attributes and
    caches must not be process-global in a way that breaks the second engine."""
    from speaksense_amd import binding
    e1 = binding.Engine(toy_ml_path, max_batch=2)
    e2 = binding.Engine(toy_en_path, dtype=binding.DTYPE_BF16, max_batch=2)
    pcm = synth.speech_like(33, 16000 * 9)
    P = binding.default_params(language="en", temperature_inc=0.0)
    a1 = e1.new_session().transcribe(pcm, P)
    b1 = e2.new_session().transcribe(pcm, P)
    a2 = e1.new_session().transcribe(pcm, P)
    b2 = e2.new_session().transcribe(pcm, P)
    assert list(a1["tokens"]) == list(a2["tokens"]) and list(b1["tokens"]) == list(b2["tokens"])
    assert len(a1["tokens"]) > 0 and len(b1["tokens"]) > 0
    e1.close(); e2.close()


def test_c_harness_matches_python_binding(toy_ml_path, eng, tmp_path):
    """The C consumer of both public headers (tests/c_harness/harness.c) gives the segments the ctypes binding gives, through the native API and
    through the whisper.h-compatible subset."""
    import subprocess
    from test_host_cpu import _build_c_harness
    from speaksense_amd import binding
    exe = _build_c_harness(tmp_path)
    r = subprocess.run([exe, toy_ml_path, "8"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.strip().splitlines()
    nat = [l[2:] for l in lines if l.startswith("N ")]
    cmp_ = [l[2:] for l in lines if l.startswith("W ")]
    assert nat and nat == cmp_
    # the same audio through the Python binding
    n = 16000 * 8
    s = np.uint32(12345)
    x = np.empty(n, np.float32)
    t = np.arange(n, dtype=np.float32) / np.float32(16000.0)
    noise = np.empty(n, np.float32)
    state = 12345
    for i in range(n):
        state = (state * 1664525 + 1013904223) & 0xFFFFFFFF
        noise[i] = (np.float32(state >> 8) / np.float32(8388608.0) - np.float32(1.0)) * np.float32(0.02)
    two_pi = np.float32(6.2831853)
    x = (np.float32(0.3) * np.sin(two_pi * np.float32(180.0) * t, dtype=np.float32) * (np.float32(0.6) + np.float32(0.4) * np.sin(two_pi * np.float32(4.0) * t, dtype=np.float32))
         + np.float32(0.15) * np.sin(two_pi * np.float32(360.0) * t, dtype=np.float32) + noise).astype(np.float32)
    got = eng.new_session().transcribe(x, binding.default_params(language="en", temperature_inc=0.0))
    py = [f"{sg['t0']} {sg['t1']} {sg['text'].decode()}" for sg in got["segments"]]
    assert len(py) == len(nat)      # sinf in C vs numpy may differ in the last bit of a few samples: compare structure, then text when equal
    if py != nat:
        assert [p.split(" ")[:2] for p in py] == [q.split(" ")[:2] for q in nat]


@pytest.mark.timeout(600)
def test_pool_two_engines_on_one_gpu(toy_ml_path):
    """ss_pool_*: the multi-GPU router with a device list that names GPU 0 twice (the box has one GPU; on a node it would be [0..7]).  Chunks of one
    session alternate between the engines (its state is host-side), every result equals the single-engine result, and a burst spreads evenly."""
    from speaksense_amd import binding
    single = binding.Engine(toy_ml_path, max_batch=4)
    pool = binding.Pool(toy_ml_path, [0, 0], max_batch=4)
    assert pool.n_engines == 2
    P = binding.default_params(language="en", temperature_inc=0.0, no_context=0)       # context carried from chunk to chunk ACROSS engines
    s_ref, s_pool = single.new_session(), pool.new_session()
    used = []
    for seed in (61, 62, 63, 64):
        pcm = synth.speech_like(seed, 16000 * 9)
        a = s_ref.transcribe(pcm, P)
        b = s_pool.transcribe(pcm, P)
        assert list(a["tokens"]) == list(b["tokens"]) and [s["text"] for s in a["segments"]] == [s["text"] for s in b["segments"]]
        used.append(s_pool.last_engine())
    assert used == [0, 1, 0, 1]
    # burst: 8 sessions at once -> 4 chunks per engine, results as from one engine
    pcms = [synth.speech_like(70 + k, 16000 * 8) for k in range(8)]
    P1 = binding.default_params(language="en", temperature_inc=0.0)
    want = [single.new_session().transcribe(x, P1) for x in pcms]
    ses = [pool.new_session() for _ in pcms]
    tickets = [s.submit(x, P1) for s, x in zip(ses, pcms)]
    got = [s.wait(t) for s, t in zip(ses, tickets)]
    for a, b in zip(want, got):
        assert list(a["tokens"]) == list(b["tokens"])
    assert sorted(s.last_engine() for s in ses) == [0] * 4 + [1] * 4
    # several tickets outstanding on ONE pool session (a single engine runs such chunks one after another in submission order): the router
    # must keep them on the engine that holds the first -- on two engines they would run concurrently on one Session, and ss_wait of the
    # first ticket would wait on the wrong engine's condition variable
    s_ref2, s_pool2 = single.new_session(), pool.new_session()
    pcms3 = [synth.speech_like(90 + k, 16000 * 9) for k in range(3)]
    for x in pcms3:
        last_ref = s_ref2.transcribe(x, P)                       # serial, context carried (no_context = 0)
    tickets = [s_pool2.submit(x, P) for x in pcms3]
    engines_used = s_pool2.last_engine()
    for t in tickets:
        last_pool = s_pool2.wait(t)
    assert s_pool2.last_engine() == engines_used
    assert list(last_pool["tokens"]) == list(last_ref["tokens"])
    b = s_pool2.transcribe(pcms3[0], P)                          # nothing in flight any more: the router may move the session again
    assert len(b["tokens"]) > 0
    pool.close(); single.close()


class WTokenData(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float),
                ("t0", C.c_int64), ("t1", C.c_int64), ("vlen", C.c_float)]


def test_whisper_h_full_surface(toy_ml_path, eng, monkeypatch):
    """The rest of whisper.h v1.5.4 (what whisper-rs can reach beyond the reference's call sites): model / token getters, whisper_tokenize,
    language helpers and detection, per-token results, whisper_full_parallel, and the low-level whisper_encode / whisper_decode /
    whisper_get_logits (round 5)."""
    from speaksense_amd import binding
    monkeypatch.setenv("SS_DTYPE", "f16")
    monkeypatch.setenv("SS_MAX_BATCH", "4")
    L = C.CDLL(binding.LIB_PATH)
    vp = C.c_void_p
    L.whisper_init_from_file.restype = vp
    L.whisper_init_from_file.argtypes = [C.c_char_p]
    L.whisper_init_state.restype = vp
    L.whisper_init_state.argtypes = [vp]
    L.whisper_full_default_params.restype = WFullParams
    L.whisper_full_default_params.argtypes = [C.c_int]
    L.whisper_full_with_state.argtypes = [vp, vp, WFullParams, vp, C.c_int]
    L.whisper_full_parallel.argtypes = [vp, WFullParams, vp, C.c_int, C.c_int]
    L.whisper_full.argtypes = [vp, WFullParams, vp, C.c_int]
    for f in ("whisper_model_n_vocab", "whisper_model_n_mels", "whisper_model_n_audio_layer", "whisper_model_n_text_state", "whisper_model_type", "whisper_n_len",
              "whisper_token_transcribe", "whisper_token_solm", "whisper_token_not", "whisper_full_n_segments", "whisper_full_lang_id"):
        getattr(L, f).argtypes = [vp]
    L.whisper_token_lang.argtypes = [vp, C.c_int]
    L.whisper_model_type_readable.restype = C.c_char_p
    L.whisper_model_type_readable.argtypes = [vp]
    L.whisper_lang_id.argtypes = [C.c_char_p]
    L.whisper_lang_str.restype = C.c_char_p
    L.whisper_lang_str_full.restype = C.c_char_p
    L.whisper_tokenize.argtypes = [vp, C.c_char_p, vp, C.c_int]
    L.whisper_full_n_segments_from_state.argtypes = [vp]
    L.whisper_full_n_tokens_from_state.argtypes = [vp, C.c_int]
    L.whisper_full_get_token_id_from_state.argtypes = [vp, C.c_int, C.c_int]
    L.whisper_full_get_token_data_from_state.restype = WTokenData
    L.whisper_full_get_token_data_from_state.argtypes = [vp, C.c_int, C.c_int]
    L.whisper_full_get_token_text_from_state.restype = C.c_char_p
    L.whisper_full_get_token_text_from_state.argtypes = [vp, vp, C.c_int, C.c_int]
    L.whisper_full_lang_id_from_state.argtypes = [vp]
    L.whisper_full_get_segment_t0.restype = C.c_int64
    L.whisper_full_get_segment_t0.argtypes = [vp, C.c_int]
    L.whisper_full_get_segment_t1.restype = C.c_int64
    L.whisper_full_get_segment_t1.argtypes = [vp, C.c_int]
    L.whisper_full_get_segment_text.restype = C.c_char_p
    L.whisper_full_get_segment_text.argtypes = [vp, C.c_int]
    L.whisper_pcm_to_mel_with_state.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.whisper_n_len_from_state.argtypes = [vp]
    L.whisper_lang_auto_detect_with_state.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.whisper_encode_with_state.argtypes = [vp, vp, C.c_int, C.c_int]
    L.whisper_get_logits_from_state.restype = vp
    L.whisper_get_logits_from_state.argtypes = [vp]
    L.whisper_free_state.argtypes = [vp]
    L.whisper_free.argtypes = [vp]

    ctx = L.whisper_init_from_file(toy_ml_path.encode())          # with a default state
    assert ctx
    assert (L.whisper_model_n_vocab(ctx), L.whisper_model_n_mels(ctx), L.whisper_model_n_audio_layer(ctx), L.whisper_model_n_text_state(ctx)) == (eng.n_vocab, 128, 2, 128)
    assert L.whisper_token_transcribe(ctx) == eng.transcribe and L.whisper_token_solm(ctx) == eng.solm and L.whisper_token_not(ctx) == eng.not_
    assert L.whisper_token_lang(ctx, 1) == eng.sot + 2
    assert L.whisper_lang_id(b"zh") == 1 and L.whisper_lang_id(b"chinese") == 1 and L.whisper_lang_id(b"klingon") == -1
    assert L.whisper_lang_str(7) == b"ja" and L.whisper_lang_str_full(7) == b"japanese" and L.whisper_lang_max_id() == 99
    text = b" " + eng.token_str(1300) + eng.token_str(2222)
    buf = (C.c_int32 * 64)()
    n = L.whisper_tokenize(ctx, text, buf, 64)
    assert list(buf[:n]) == eng.tokenize(text) and L.whisper_tokenize(ctx, text, buf, 1) == -n
    assert L.whisper_tokenize(ctx, text, None, 0) == -n      # whisper.h: -(needed) whenever n_max_tokens < needed, NULL buffer included (whisper_token_count relies on the sign)

    st = L.whisper_init_state(ctx)
    pcm = synth.speech_like(23, 16000 * 12)
    p = L.whisper_full_default_params(0)
    p.language = b"auto"; p.temperature_inc = 0.0
    assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(vp), len(pcm)) == 0
    ref = eng.new_session().transcribe(pcm, binding.default_params(language="auto", temperature_inc=0.0))
    assert L.whisper_full_lang_id_from_state(st) == ref["lang_id"]
    n_seg = L.whisper_full_n_segments_from_state(st)
    assert n_seg == len(ref["segments"]) and n_seg > 0
    ids = []
    for i in range(n_seg):
        for k in range(L.whisper_full_n_tokens_from_state(st, i)):
            d = L.whisper_full_get_token_data_from_state(st, i, k)
            assert d.id == L.whisper_full_get_token_id_from_state(st, i, k) and 0.0 < d.p <= 1.0 and d.t0 == -1
            assert L.whisper_full_get_token_text_from_state(ctx, st, i, k) == eng.token_str(d.id)
            ids.append(d.id)
    assert ids == [int(t) for t in ref["tokens"]][:len(ids)] and len(ids) > 0      # segment tokens = the accepted stream (minus a trailing text-less tail)

    # callbacks at chunk granularity: abort / encoder_begin before the chunk is submitted (-6 when they say stop), progress(100) and new_segment(all segments) after
    seen = []
    NEWSEG = C.CFUNCTYPE(None, vp, vp, C.c_int, vp)
    PROG = C.CFUNCTYPE(None, vp, vp, C.c_int, vp)
    ENCB = C.CFUNCTYPE(C.c_bool, vp, vp, vp)
    ABORT = C.CFUNCTYPE(C.c_bool, vp)
    cb_new = NEWSEG(lambda c, s_, n_new, ud: seen.append(("new", n_new, L.whisper_full_n_segments_from_state(s_))))
    cb_prog = PROG(lambda c, s_, pr, ud: seen.append(("progress", pr)))
    go = {"enc": True, "abort": False}
    cb_enc = ENCB(lambda c, s_, ud: go["enc"])
    cb_abort = ABORT(lambda ud: go["abort"])
    pc = L.whisper_full_default_params(0)
    pc.language = b"auto"; pc.temperature_inc = 0.0
    pc.new_segment_callback = C.cast(cb_new, vp); pc.progress_callback = C.cast(cb_prog, vp)
    pc.encoder_begin_callback = C.cast(cb_enc, vp); pc.abort_callback = C.cast(cb_abort, vp)
    assert L.whisper_full_with_state(ctx, st, pc, pcm.ctypes.data_as(vp), len(pcm)) == 0
    assert seen == [("progress", 100), ("new", n_seg, n_seg)]
    go["enc"] = False
    assert L.whisper_full_with_state(ctx, st, pc, pcm.ctypes.data_as(vp), len(pcm)) == -6
    go["enc"], go["abort"] = True, True
    assert L.whisper_full_with_state(ctx, st, pc, pcm.ctypes.data_as(vp), len(pcm)) == -6 and len(seen) == 2

    # language detection alone, on the samples given to pcm_to_mel
    assert L.whisper_pcm_to_mel_with_state(ctx, st, pcm.ctypes.data_as(vp), len(pcm), 4) == 0
    assert L.whisper_n_len_from_state(st) == (len(pcm) + 480000) // 160
    probs = (C.c_float * 100)()
    assert L.whisper_lang_auto_detect_with_state(ctx, st, 0, 4, probs) == ref["lang_id"] and probs[ref["lang_id"]] == 1.0
    # whisper.h's low-level API (whisper-rs: state.encode / state.decode / state.get_logits) on the engine's stage hooks: the logits of the last token
    # are the bits ss_session_decode gives a native session fed the same spectrogram window (same kernels, and a row's bits do not depend on the pass)
    L.whisper_decode_with_state.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.whisper_get_logits_from_state.restype = C.POINTER(C.c_float)
    assert not L.whisper_get_logits_from_state(st)                                    # nothing decoded yet
    toks = np.array([eng.sot, eng.sot + 1, eng.transcribe, 1234, 4321], np.int32)
    assert L.whisper_decode_with_state(ctx, st, toks.ctypes.data_as(vp), 3, 0, 4) == -1      # no whisper_encode yet (whisper_full reset the state's window)
    assert L.whisper_encode_with_state(ctx, st, -1, 4) == -1
    assert L.whisper_encode_with_state(ctx, st, 100000, 4) == 0                       # offset beyond the spectrogram: a zero-padded window, as whisper.cpp
    assert L.whisper_encode_with_state(ctx, st, 200, 4) == 0
    assert L.whisper_decode_with_state(ctx, st, toks.ctypes.data_as(vp), 3, 0, 4) == 0
    got1 = np.ctypeslib.as_array(L.whisper_get_logits_from_state(st), (eng.n_vocab,)).copy()
    assert L.whisper_decode_with_state(ctx, st, toks[3:].ctypes.data_as(vp), 1, 3, 4) == 0
    got2 = np.ctypeslib.as_array(L.whisper_get_logits_from_state(st), (eng.n_vocab,)).copy()
    assert L.whisper_decode_with_state(ctx, st, toks[4:].ctypes.data_as(vp), 1, 9, 4) == -1      # history beyond what was decoded since whisper_encode
    ses = eng.new_session()
    ses.set_encoder(eng.encode(eng.log_mel(pcm), 200))                                # a native session of ANOTHER engine (`eng` is not the context's): same bits
    assert np.array_equal(got1, ses.decode(toks[:3], 0)) and np.array_equal(got2, ses.decode(toks[3:4], 3))
    ses2 = eng.new_session()
    ses2.set_encoder(eng.encode(eng.log_mel(pcm), 0))                                 # ... and on one engine a second session takes the stage-hook slot:
    with pytest.raises(binding.SpeakSenseError):                                      # the first is refused, not answered from the other's audio
        ses.decode(toks[3:4], 3)
    ses.close(); ses2.close()
    ses3 = eng.new_session()                                                          # the owner was freed: whoever comes next (possibly at its address) must encode first
    with pytest.raises(binding.SpeakSenseError):
        ses3.decode(toks[:3], 0)
    ses3.close()
    # two states on ONE context (whisper-rs create_state() twice), interleaved.  The decoder context behind whisper_encode / whisper_decode is
    # one per engine: a state that lost it to the other's whisper_encode (or to a whisper_full on lane 0) gets -1 until it encodes again -- never the
    # other state's audio with return code 0.
    st2 = L.whisper_init_state(ctx)
    pcm2 = synth.speech_like(77, 16000 * 8)
    assert L.whisper_pcm_to_mel_with_state(ctx, st2, pcm2.ctypes.data_as(vp), len(pcm2), 4) == 0
    assert L.whisper_encode_with_state(ctx, st, 200, 4) == 0                          # A encodes
    assert L.whisper_encode_with_state(ctx, st2, 0, 4) == 0                           # B encodes: the slot is B's
    assert L.whisper_decode_with_state(ctx, st, toks.ctypes.data_as(vp), 3, 0, 4) == -1      # A must not see B's audio
    assert L.whisper_decode_with_state(ctx, st2, toks.ctypes.data_as(vp), 3, 0, 4) == 0
    b1 = np.ctypeslib.as_array(L.whisper_get_logits_from_state(st2), (eng.n_vocab,)).copy()
    assert not np.array_equal(b1, got1)                                                # other audio, other logits
    assert L.whisper_encode_with_state(ctx, st, 200, 4) == 0                          # A again
    assert L.whisper_decode_with_state(ctx, st2, toks[3:].ctypes.data_as(vp), 1, 3, 4) == -1     # now B lost it (and its self-KV history)
    assert L.whisper_decode_with_state(ctx, st, toks.ctypes.data_as(vp), 3, 0, 4) == 0
    assert np.array_equal(np.ctypeslib.as_array(L.whisper_get_logits_from_state(st), (eng.n_vocab,)), got1)
    assert L.whisper_full_with_state(ctx, st2, p, pcm2.ctypes.data_as(vp), len(pcm2)) == 0        # a transcription on the other state (any lane)
    rc = L.whisper_decode_with_state(ctx, st, toks[3:].ctypes.data_as(vp), 1, 3, 4)
    assert rc == -1 or (rc == 0 and np.array_equal(np.ctypeslib.as_array(L.whisper_get_logits_from_state(st), (eng.n_vocab,)), got2))
    L.whisper_free_state(st2)

    # whisper_full_parallel: two halves as one device batch, merged on the context's default state with whisper.cpp's offset rule
    long_pcm = synth.speech_like(24, 16000 * 24)
    q = L.whisper_full_default_params(0)
    q.language = b"en"; q.temperature_inc = 0.0
    assert L.whisper_full_parallel(ctx, q, long_pcm.ctypes.data_as(vp), len(long_pcm), 2) == 0
    half = len(long_pcm) // 2
    P = binding.default_params(language="en", temperature_inc=0.0)
    a = eng.new_session().transcribe(long_pcm[:half], P)
    b = eng.new_session().transcribe(long_pcm[half:], P)
    want = [(s["t0"], s["t1"], s["text"]) for s in a["segments"]]
    off = 100 * half // 16000
    for s in b["segments"]:
        t0 = max(s["t0"] + off, want[-1][1]) if want else s["t0"] + off
        want.append((t0, s["t1"] + off, s["text"]))
    got = [(L.whisper_full_get_segment_t0(ctx, i), L.whisper_full_get_segment_t1(ctx, i), L.whisper_full_get_segment_text(ctx, i)) for i in range(L.whisper_full_n_segments(ctx))]
    assert got == want and len(got) > len(a["segments"])
    L.whisper_free_state(st)
    L.whisper_free(ctx)


class WContextParams155(C.Structure):      # whisper.h v1.5.5 (include/whisper_compat.h, SS_WHISPER_H_POST_1_5_4; tests/golden/abi_layout_post_1_5_4.txt)
    _fields_ = [("use_gpu", C.c_bool), ("gpu_device", C.c_int), ("dtw_token_timestamps", C.c_bool), ("dtw_aheads_preset", C.c_int), ("dtw_n_top", C.c_int),
                ("dtw_n_heads", C.c_size_t), ("dtw_heads", C.c_void_p), ("dtw_mem_size", C.c_size_t)]


class WTokenData155(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float), ("pt", C.c_float), ("ptsum", C.c_float),
                ("t0", C.c_int64), ("t1", C.c_int64), ("t_dtw", C.c_int64), ("vlen", C.c_float)]


def test_whisper_h_post_1_5_4_library(toy_ml_path, eng, monkeypatch):
    """libspeaksense_whisper_post154.so: the shim laid out as whisper.h v1.5.5 -- a 48-byte whisper_context_params passed BY VALUE (in memory, not in
    a register as the one-byte v1.5.4 struct), gpu_device honoured, whisper_token_data with t_dtw = -1 -- gives the results of the native API."""
    from speaksense_amd import binding, build
    monkeypatch.setenv("SS_DTYPE", "f16")
    monkeypatch.setenv("SS_MAX_BATCH", "4")
    assert C.sizeof(WContextParams155) == 48 and C.sizeof(WTokenData155) == 56
    L = C.CDLL(build.LIB_W155)
    vp = C.c_void_p
    L.whisper_context_default_params.restype = WContextParams155
    L.whisper_init_from_file_with_params_no_state.restype = vp
    L.whisper_init_from_file_with_params_no_state.argtypes = [C.c_char_p, WContextParams155]
    L.whisper_init_state.restype = vp
    L.whisper_init_state.argtypes = [vp]
    L.whisper_full_default_params.restype = WFullParams
    L.whisper_full_default_params.argtypes = [C.c_int]
    L.whisper_full_with_state.argtypes = [vp, vp, WFullParams, vp, C.c_int]
    L.whisper_full_n_segments_from_state.argtypes = [vp]
    L.whisper_full_n_tokens_from_state.argtypes = [vp, C.c_int]
    L.whisper_full_get_token_data_from_state.restype = WTokenData155
    L.whisper_full_get_token_data_from_state.argtypes = [vp, C.c_int, C.c_int]
    L.whisper_full_get_segment_text_from_state.restype = C.c_char_p
    L.whisper_full_get_segment_text_from_state.argtypes = [vp, C.c_int]
    for f in ("whisper_full_get_segment_t0_from_state", "whisper_full_get_segment_t1_from_state"):
        getattr(L, f).restype = C.c_int64
        getattr(L, f).argtypes = [vp, C.c_int]
    L.whisper_free_state.argtypes = [vp]
    L.whisper_free.argtypes = [vp]
    cp = L.whisper_context_default_params()
    assert (cp.use_gpu, cp.gpu_device, cp.dtw_token_timestamps, cp.dtw_aheads_preset, cp.dtw_n_top, cp.dtw_n_heads, cp.dtw_mem_size) == (True, 0, False, 0, -1, 0, 128 << 20)
    bad = L.whisper_context_default_params()
    bad.gpu_device = 63                                            # no such device: the field is read where the v1.5.5 header puts it
    assert not L.whisper_init_from_file_with_params_no_state(toy_ml_path.encode(), bad)
    ctx = L.whisper_init_from_file_with_params_no_state(toy_ml_path.encode(), cp)
    assert ctx
    st = L.whisper_init_state(ctx)
    pcm = synth.speech_like(29, 16000 * 14)
    p = L.whisper_full_default_params(0)
    p.language = b"en"; p.temperature_inc = 0.0; p.token_timestamps = True
    assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(vp), len(pcm)) == 0
    ref = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0))
    n_seg = L.whisper_full_n_segments_from_state(st)
    assert n_seg == len(ref["segments"]) and n_seg > 0
    got = [(L.whisper_full_get_segment_t0_from_state(st, i), L.whisper_full_get_segment_t1_from_state(st, i), L.whisper_full_get_segment_text_from_state(st, i))
           for i in range(n_seg)]
    assert got == [(s["t0"], s["t1"], s["text"]) for s in ref["segments"]]
    ids = []
    for i in range(n_seg):
        for k in range(L.whisper_full_n_tokens_from_state(st, i)):
            d = L.whisper_full_get_token_data_from_state(st, i, k)
            assert d.t_dtw == -1 and 0.0 < d.p <= 1.0 and d.t0 >= 0 and d.t1 >= d.t0 and d.vlen >= 0.0
            ids.append(d.id)
    assert ids == [int(t) for t in ref["tokens"]][:len(ids)] and len(ids) > 0
    L.whisper_free_state(st)
    L.whisper_free(ctx)


@pytest.mark.parametrize("ft", ["q5_0", "q5_1", "q8_0", "q4_0"])
def test_quantised_ggml_models(tmp_path, ft, capfd):
    """Block-quantised ggml files (script/download-ggml-model.sh:28-51 lists the -q5_0 / -q5_1 variants): dequantised at load into the engine's
    f16 operands; ids, segments and timestamps identical to the oracle reading the same file, stages within the f16 tolerances."""
    from oracle import binding as orc
    from speaksense_amd import binding, ggml_io
    path = str(tmp_path / f"toy-{ft}.bin")
    ggml_io.write_model(path, "toy", seed=1, ftype=ft)
    om = orc.OracleModel(path)
    capfd.readouterr()
    e = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=2)
    # the library says at load that such a file runs de-quantised 16-bit arithmetic, not ggml's quantised x q8_0 one
    assert "block-quantised tensors were de-quantised" in capfd.readouterr().err
    assert e.ftype == 2000 + ggml_io.FTYPE_BY_NAME[ft]
    pcm = synth.speech_like(5, 16000 * 20)
    mel = om.log_mel(pcm)
    ref_enc = om.encode(mel, 0, orc.MODE_GGML_F16)
    assert np.abs(e.encode(mel, 0) - ref_enc).max() / np.abs(ref_enc).max() < 4e-3
    P = dict(language="en", temperature_inc=0.0)
    got = e.new_session().transcribe(pcm, binding.default_params(**P))
    ref = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(**P))
    assert list(got["tokens"]) == list(ref["tokens"]) and len(got["tokens"]) > 0
    assert [(s["t0"], s["t1"], s["text"]) for s in got["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in ref["segments"]]
    e.close(); om.close()


def test_continuous_batching_early_completion_and_admission(toy_ml_path):
    """A device group reports each chunk as soon as ITS last window is done, and while multi-window chunks keep the group alive it admits queued
    chunks into its free slots at the next window boundary (one lane, max_batch 4).  Results must equal the one-chunk-at-a-time results."""
    import threading
    import time
    from speaksense_amd import binding
    ref_eng = binding.Engine(toy_ml_path, max_batch=4, n_lanes=1)
    P = binding.default_params(language="en", temperature_inc=0.0)
    long_pcms = [synth.speech_like(80 + k) for k in range(2)]                  # 30 s: several windows each on the toy model
    short_pcms = [synth.speech_like(90 + k, 16000 * 3) for k in range(5)]       # 3 s: one window
    want_long = [ref_eng.new_session().transcribe(x, P) for x in long_pcms]
    want_short = [ref_eng.new_session().transcribe(x, P) for x in short_pcms]
    assert all(r["n_encode"] >= 2 for r in want_long)
    ref_eng.close()

    eng = binding.Engine(toy_ml_path, max_batch=4, n_lanes=1, batch_wait_us=50000)
    s_long = [eng.new_session() for _ in long_pcms]
    s_short = [eng.new_session() for _ in short_pcms]
    # 7 chunks at once, 4 slots: the batch former takes 2 long + 2 short; the other 3 short ones wait in the queue for a free slot
    tickets = [(s, s.submit(x, P), "long") for s, x in zip(s_long, long_pcms)] + [(s, s.submit(x, P), "short") for s, x in zip(s_short, short_pcms)]
    done_at, results = {}, {}

    def waiter(i, s, t):
        results[i] = s.wait(t)
        done_at[i] = time.perf_counter()

    th = [threading.Thread(target=waiter, args=(i, s, t)) for i, (s, t, _) in enumerate(tickets)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for i, b in enumerate(want_long + want_short):
        a = results[i]
        assert list(a["tokens"]) == list(b["tokens"]) and [s["text"] for s in a["segments"]] == [s["text"] for s in b["segments"]], i
    tot = eng.totals()
    first_short, last_long = min(done_at[2], done_at[3]), max(done_at[0], done_at[1])
    print(f"one-window chunks of the first batch back {1e3 * (last_long - first_short):.1f} ms before the multi-window ones; admitted into the running group: {tot['admitted']}, "
          f"windows started while others were decoding: {tot['started_midway']}")
    assert first_short < last_long, "the one-window chunks were held until the multi-window chunks of their group had finished"
    assert tot["admitted"] >= 1, "no queued chunk joined the running group at a window boundary"
    assert tot["started_midway"] >= 1, "no window was started while other windows of the group were still decoding (rows freed by an early end were not refilled)"
    eng.close()


def test_lanes_under_concurrent_callers(toy_ml_path):
    """Two lanes, three kinds of callers at once: threads in blocking ss_transcribe_batch calls (groups spread over the lanes), async tickets
    through the batch former, and a session reused for consecutive chunks.  Every result equals the serial single-lane result; nothing hangs."""
    import threading
    from speaksense_amd import binding
    serial = binding.Engine(toy_ml_path, max_batch=4, n_lanes=1)
    P = binding.default_params(language="en", temperature_inc=0.0)
    pcms = [synth.speech_like(200 + k, 16000 * (5 + k % 4)) for k in range(18)]
    want = [serial.new_session().transcribe(x, P) for x in pcms]
    serial.close()
    eng = binding.Engine(toy_ml_path, max_batch=4, n_lanes=2, batch_wait_us=1000)
    got = [None] * len(pcms)
    errors = []

    def blocking(idx):
        try:
            ses = [eng.new_session() for _ in idx]
            res = eng.transcribe_batch(ses, [pcms[i] for i in idx], P)
            for i, r in zip(idx, res):
                got[i] = r
        except Exception as e:   # pragma: no cover
            errors.append(e)

    def reused(idx):
        try:
            s = eng.new_session()
            for i in idx:
                got[i] = s.transcribe(pcms[i], P)
        except Exception as e:   # pragma: no cover
            errors.append(e)

    th = [threading.Thread(target=blocking, args=(list(range(0, 6)),)), threading.Thread(target=blocking, args=(list(range(6, 9)),)),
          threading.Thread(target=reused, args=(list(range(9, 12)),))]
    for t in th:
        t.start()
    ses = [eng.new_session() for _ in range(12, 18)]
    tickets = [s.submit(pcms[i], P) for s, i in zip(ses, range(12, 18))]
    for s, t, i in zip(ses, tickets, range(12, 18)):
        got[i] = s.wait(t)
    for t in th:
        t.join(timeout=120)
        assert not t.is_alive(), "a caller is stuck"
    assert not errors, errors
    for i, (a, b) in enumerate(zip(got, want)):
        assert a is not None and list(a["tokens"]) == list(b["tokens"]), i
    tot = eng.totals()
    assert tot["n_lanes"] == 2 and tot["encoder_windows"] >= len(pcms)
    eng.close()


@pytest.mark.parametrize("env", [{"SS_DECODE_GRAPH": "0"}, {"SS_DECODE_CHAIN": "0"}, {"SS_DECODE_GRAPH": "0", "SS_DECODE_CHAIN": "0"}, {"SS_LN_FUSE": "0"}])
def test_decode_issue_modes(toy_ml_path, om, orc, monkeypatch, env):
    """The two switches left in the engine change how a decoder pass is ISSUED, not which kernels run: SS_DECODE_GRAPH=0 launches the pass kernel by
    kernel instead of replaying its captured hipGraph, SS_DECODE_CHAIN=0 waits for every step's samples before enqueuing the next step (no
    device-side advance of the control blocks).  Every mode must give the oracle's tokens and segments -- greedy, a batch of 4 with an early EOT
    and a multi-window chunk, and the sampled fallback ladder."""
    from speaksense_amd import binding
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = binding.Engine(toy_ml_path, dtype=binding.DTYPE_F16, max_batch=4)
    P, OP = binding.default_params(language="en", temperature_inc=0.0), orc.default_params(language="en", temperature_inc=0.0)
    pcms = [synth.speech_like(3), synth.speech_like(4, 16000 * 47), synth.speech_like(6, 16000 * 9), synth.silence()]
    got = eng.transcribe_batch([eng.new_session() for _ in pcms], pcms, P)
    for x, g in zip(pcms, got):
        ref = om.new_state(orc.MODE_GGML_F16).full(x, OP)
        assert list(g["tokens"]) == list(ref["tokens"]) and [(s["t0"], s["t1"], s["text"]) for s in g["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in ref["segments"]]
    x = synth.speech_like(5)      # walks the temperature ladder (tests/test_gpu_parity.py::test_full_path_default_ladder_f16)
    g = eng.new_session().transcribe(x, binding.default_params(language="en"))
    from test_gpu_parity import GAP_TOL_F16, check_trace_against_oracle
    assert g["n_fail"] >= 1
    check_trace_against_oracle(g, om, orc, orc.MODE_GGML_F16, x, orc.default_params(language="en"), f"issue mode {env}", GAP_TOL_F16)
    eng.close()
