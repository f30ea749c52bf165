"""GPU: `audio_ctx` below the model's n_audio_ctx on the HIP engine (round 5; refused with -9 before).  The encoder pass covers the first audio_ctx
positions in buffers laid out for that context, the cross-KV cache keeps the model's geometry and a shortened window fills the first audio_ctx key rows
of its slot, every decoder row carries its window's key count (RowCtl.n_keys).  Held to HF `generate` on models whose max_source_positions is the
shortened context (tests/golden/hf_audio_ctx_golden.npz), to the oracle under the reference's real parameters, and to itself: chunks of different
contexts sharing one engine and one decoder pass give their single-chunk results."""
import ctypes as C
import os

import numpy as np
import pytest

from speaksense_amd import synth
from conftest import report
from test_oracle_audio_ctx import audio_ctx_case_model, audio_ctx_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("which", ["f16", "bf16"])
@pytest.mark.parametrize("ci", range(3))
def test_engine_shortened_context_matches_hf_generate(ci, which, model_dir):
    from speaksense_amd import binding
    from oracle import binding as orc
    from test_gpu_parity import GAP_TOL_BF16, GAP_TOL_F16, check_against_oracle
    c = audio_ctx_cases()[ci]
    path = audio_ctx_case_model(c, model_dir)
    dtype, omode, gap_tol = (binding.DTYPE_F16, orc.MODE_GGML_F16, GAP_TOL_F16) if which == "f16" else (binding.DTYPE_BF16, orc.MODE_BF16, GAP_TOL_BF16)
    eng = binding.Engine(path, dtype=dtype, max_batch=2, compat=binding.COMPAT_OPENAI_TS_RULES)
    pcm = synth.speech_like(c["audio"])
    kw = dict(language="en", temperature_inc=0.0, audio_ctx=c["audio_ctx"], duration_ms=30000)
    got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
    tr = [int(t) for t in got["trace"]]
    n0 = c["n0"]
    same = tr[:n0] == c["ids"][:n0]
    worst = 0.0
    if same:
        assert [(s["t0"], s["t1"]) for s in got["segments"]][:len(c["seg"])] == c["seg"]
    else:
        om = orc.OracleModel(path)
        _, worst = check_against_oracle(got, om, orc, omode, pcm, orc.default_params(**kw), f"audio_ctx {c['audio_ctx']} case {ci} ({which})", gap_tol, replay_only=True,
                                        compat=orc.COMPAT_OPENAI_TS_RULES)
        om.close()
    if which == "f16":
        assert same or ci != 0
    report(f"HIP engine ({which}) vs HF generate with max_source_positions = audio_ctx = {c['audio_ctx']} ({c['preset']}, {n0} ids): "
           + ("ids and segment times identical" if same else f"near-tie flip(s) proven by forced replay, largest margin {worst:.4f}"))
    eng.close()


def test_engine_audio_ctx_matches_oracle_with_the_ladder(model_dir):
    """The reference's real parameters (best_of 5, temperature ladder) with shortened contexts, 70 s of audio (several windows: the seek still advances by
    the timestamps, whatever the encoder saw): every sampled id of every attempt replayed on the oracle."""
    from speaksense_amd import binding
    from oracle import binding as orc
    from test_gpu_parity import GAP_TOL_F16, check_trace_against_oracle
    c = audio_ctx_cases()[0]
    path = audio_ctx_case_model(c, model_dir)
    eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=2)
    om = orc.OracleModel(path)
    pcm = np.concatenate([synth.speech_like(31), synth.speech_like(32), synth.speech_like(33)])[:16000 * 70]
    worst = 0.0
    for A in (752, 256):
        got = eng.new_session().transcribe(pcm, binding.default_params(language="en", audio_ctx=A))
        assert got["n_windows"] >= 2
        fg, fs, wg, ws = check_trace_against_oracle(got, om, orc, orc.MODE_GGML_F16, pcm, orc.default_params(language="en", audio_ctx=A), f"audio_ctx {A}", GAP_TOL_F16)
        worst = max(worst, wg)
    report(f"audio_ctx 752 / 256 with the ladder (f16, {c['preset']}): every sampled id replayed on the oracle, largest greedy margin {worst:.4f}")
    om.close(); eng.close()


def test_fp8_engine_audio_ctx_forced_replay(model_dir):
    """The e4m3 engine (wide2: d = 1280) with shortened contexts, greedy: every pick is the FP8-mode oracle's argmax or within the margin this model shows
    at EVERY context, the full one included (profiles/r05_z_fp8_audio_ctx_margins.txt: 3 % of the picks differ, margins up to 0.45 at 1500, 0.56 at
    1024, 0.60 at 512 -- the natural-EOT wide2 weights are harder on e4m3 than the default ones; f16 differs nowhere)."""
    from speaksense_amd import binding
    from oracle import binding as orc
    from test_gpu_parity import check_against_oracle
    c = audio_ctx_cases()[2]
    path = audio_ctx_case_model(c, model_dir)
    eng = binding.Engine(path, dtype=binding.DTYPE_FP8, max_batch=2)
    om = orc.OracleModel(path)
    worst, n_same = 0.0, 0
    for A, seed in ((1000, 31), (752, 32), (500, 33)):
        pcm = synth.speech_like(seed)
        kw = dict(language="en", temperature_inc=0.0, audio_ctx=A)
        got = eng.new_session().transcribe(pcm, binding.default_params(**kw))
        same, w = check_against_oracle(got, om, orc, orc.MODE_FP8, pcm, orc.default_params(**kw), f"fp8 audio_ctx {A}", 0.8)
        n_same += same; worst = max(worst, w)
    report(f"fp8 engine, audio_ctx 1000 / 752 / 500 (wide2): {n_same}/3 chunks identical to the FP8-mode oracle, the rest proven by forced replay, largest margin {worst:.4f}")
    om.close(); eng.close()


def test_mixed_contexts_share_an_engine(model_dir):
    """Chunks with audio_ctx 0 / 752 / 256 / 1000 submitted together (their windows ride in the same decoder passes, their encoder passes alternate
    contexts): each equals its single-chunk run bit for bit, and the full-context chunk equals a run on an engine that never saw a shortened one."""
    from speaksense_amd import binding
    c = audio_ctx_cases()[0]
    path = audio_ctx_case_model(c, model_dir)
    ctxs = [0, 752, 256, 1000, 0, 256, 752, 0]
    pcms = [synth.speech_like(40 + i) for i in range(len(ctxs))]
    fresh = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=2)
    want_full = [list(fresh.new_session().transcribe(pcms[i], binding.default_params(language="en"))["trace"]) for i in (0, 4, 7)]
    fresh.close()
    eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=8, batch_wait_us=300000)
    ses = [eng.new_session() for _ in ctxs]
    tickets = [s.submit(p, binding.default_params(language="en", audio_ctx=a)) for s, p, a in zip(ses, pcms, ctxs)]
    res = [s.wait(t) for s, t in zip(ses, tickets)]
    singles = [eng.new_session().transcribe(p, binding.default_params(language="en", audio_ctx=a)) for p, a in zip(pcms, ctxs)]
    for i, (r, s1) in enumerate(zip(res, singles)):
        assert list(r["trace"]) == list(s1["trace"]), f"chunk {i} (audio_ctx {ctxs[i]}) differs from its single run"
        assert [(s["t0"], s["t1"], s["text"]) for s in r["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in s1["segments"]]
    assert [list(res[i]["trace"]) for i in (0, 4, 7)] == want_full
    # the same audio under two contexts is two different transcriptions (the context is not ignored)
    a = eng.new_session().transcribe(pcms[0], binding.default_params(language="en", audio_ctx=256))
    assert list(a["trace"]) != list(res[0]["trace"])
    report(f"mixed audio_ctx {ctxs} in one engine: 8/8 chunks equal their single-chunk runs; full-context chunks equal an engine that never shortened")
    eng.close()


def test_audio_ctx_refusals_and_shim(model_dir, monkeypatch):
    from speaksense_amd import binding
    from test_gpu_variants import WCtxParams, WFullParams
    c = audio_ctx_cases()[1]          # toy256: multilingual
    path = audio_ctx_case_model(c, model_dir)
    eng = binding.Engine(path, dtype=binding.DTYPE_F16, max_batch=2)
    pcm = synth.speech_like(5)[:16000 * 10]
    for kw, code in ((dict(audio_ctx=750), -9), (dict(audio_ctx=1504), -5), (dict(audio_ctx=-4), -9), (dict(audio_ctx=500, language="auto"), -9),
                     (dict(audio_ctx=500, detect_language=1), -9)):
        with pytest.raises(binding.SpeakSenseError) as e:
            eng.new_session().transcribe(pcm, binding.default_params(**dict(dict(language="en"), **kw)))
        assert e.value.code == code, (kw, e.value.code)
    want = eng.new_session().transcribe(pcm, binding.default_params(language="en", audio_ctx=500, temperature_inc=0.0))
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
    L.whisper_free_state.argtypes = [C.c_void_p]
    L.whisper_free.argtypes = [C.c_void_p]
    ctx = L.whisper_init_from_file_with_params_no_state(path.encode(), L.whisper_context_default_params())
    st = L.whisper_init_state(ctx)
    p = L.whisper_full_default_params(0)
    p.language = b"en"; p.audio_ctx = 500; p.temperature_inc = 0.0; p.no_context = True; p.token_timestamps = True
    assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(C.c_void_p), len(pcm)) == 0
    assert L.whisper_full_n_segments_from_state(st) == len(want["segments"])
    for i, s in enumerate(want["segments"]):
        assert L.whisper_full_get_segment_text_from_state(st, i) == s["text"]
    p.audio_ctx = 1504
    assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(C.c_void_p), len(pcm)) == -5
    L.whisper_free_state(st); L.whisper_free(ctx)
    eng.close()


@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_shared_encoder_pass_does_not_change_a_short_context_window(which, wide2_path, toy_ml_path):
    """Found by a 600 s soak in round 5 (1 of 102 688 results): the f16 / bf16 GEMM launcher picked its kernel by M = windows x positions, so ONE 752-position
    window (M < 1024) ran the 128 x 128 kernel and two together the 256 x 256 one -- different summation grouping, different low bits, and a decode with
    forced temperature fallbacks (sampled tokens) told them apart.  A window's result may not depend on what shares its encoder pass: chunks decoded with
    forced fallbacks, alone and as 2 / 3 copies in one pass, at contexts on both sides of every tile boundary."""
    from speaksense_amd import binding
    dtype = {"f16": binding.DTYPE_F16, "bf16": binding.DTYPE_BF16, "fp8": binding.DTYPE_FP8}[which]
    n_checked = 0
    for path, ctxs in ((toy_ml_path, (256, 700, 752, 768, 1000, 0)), (wide2_path, (500, 752))):
        if which == "fp8" and path == toy_ml_path:
            continue                                   # the e4m3 engine needs n_audio_state % 256 == 0
        eng = binding.Engine(path, dtype=dtype, max_batch=8, n_lanes=1, batch_wait_us=200000)
        for A in ctxs:
            for seed in (1, 3):
                X = synth.speech_like(seed, 16000 * 3)
                P = binding.default_params(language="en", audio_ctx=A, temperature_inc=0.2, logprob_thold=0.0)
                alone = [int(t) for t in eng.new_session().transcribe(X, P)["trace"]]
                for n in (2, 3):
                    ss = [eng.new_session() for _ in range(n)]
                    ts = [s.submit(X, P) for s in ss]
                    for s, t in zip(ss, ts):
                        r = s.wait(t)
                        assert r["n_fail"] >= 1
                        assert [int(x) for x in r["trace"]] == alone, f"{os.path.basename(path)} audio_ctx {A} seed {seed}: {n} copies in one encoder pass differ from the single run"
                        n_checked += 1
        eng.close()
    report(f"shared encoder passes at shortened contexts ({which}): {n_checked} chunks with forced fallbacks equal their single runs")


@pytest.mark.parametrize("which", ["f16", "bf16"])
@pytest.mark.parametrize("ci", range(3))
def test_engine_shortened_encoder_rows_match_hf(ci, which, model_dir):
    """Stage level, no oracle in between: `ss_encode_ctx` against the encoder rows of the HF model whose max_source_positions is the shortened context --
    the LAST rows included (the ones a wrong cut or a stale padding row would change) -- at the tolerances of the full-context stage tests
    (tests/test_gpu_golden.py); and the last row is far from the full-context pass's row at the same position."""
    from speaksense_amd import binding
    from test_gpu_golden import ENC_TOL_BF16, ENC_TOL_F16
    c = audio_ctx_cases()[ci]
    path = audio_ctx_case_model(c, model_dir)
    eng = binding.Engine(path, dtype=binding.DTYPE_F16 if which == "f16" else binding.DTYPE_BF16, max_batch=2)
    pcm = synth.speech_like(c["audio"])
    mel = eng.log_mel(pcm)
    A = c["audio_ctx"]
    full = eng.encode(mel, 0)
    enc = eng.encode(mel, 0, audio_ctx=A)
    again = eng.encode(mel, 0)
    assert enc.shape == (A, eng.n_audio_state) and np.array_equal(full, again)          # the context switch leaves the full-context pass bit-identical
    err = np.abs(enc[c["rows"]] - c["enc"]).max() / c["enc_absmax"]
    tol = ENC_TOL_F16 if which == "f16" else ENC_TOL_BF16
    assert err < tol, err
    assert np.abs(full[A - 1] - enc[A - 1]).max() / c["enc_absmax"] > 0.1          # (measured 0.27 - 0.29: another computation, not noise)
    report(f"ss_encode_ctx ({which}) vs HF at max_source_positions = {A} ({c['preset']}): encoder rows {err:.2e} of absmax (tol {tol})")
    eng.close()


@pytest.mark.parametrize("ci", range(3))
def test_engine_shortened_decoder_logits_match_hf(ci, model_dir):
    """Stage level for the decoder: the HF model's encoder rows are not stored whole, so the engine's own short-context encoder output (held to HF's rows
    above) goes into `ss_session_set_encoder_ctx`, and the top-16 logits of a prompt pass + 8 KV-cached steps over audio_ctx keys are compared with HF's
    at the full-context tolerance (tests/test_gpu_golden.py LOGIT_TOL_F16, in units of the logits' standard deviation)."""
    from speaksense_amd import binding
    from test_gpu_golden import LOGIT_TOL_F16
    c = audio_ctx_cases()[ci]
    eng = binding.Engine(audio_ctx_case_model(c, model_dir), dtype=binding.DTYPE_F16, max_batch=2)
    pcm = synth.speech_like(c["audio"])
    enc = eng.encode(eng.log_mel(pcm), 0, audio_ctx=c["audio_ctx"])
    ses = eng.new_session()
    ses.set_encoder(enc)
    n_p, worst = c["n_prompt"], 0.0
    for pos in range(n_p - 1, len(c["tokens"])):
        lg = ses.decode(c["tokens"][:n_p], 0) if pos == n_p - 1 else ses.decode(c["tokens"][pos:pos + 1], pos)
        e = np.abs(lg[c["topk"][pos]] - c["topv"][pos]).max() / c["logit_std"]
        worst = max(worst, e)
        assert e < LOGIT_TOL_F16, (pos, e)
    # the same session back on a full-context encoder output attends over all 1500 keys again
    full = eng.encode(eng.log_mel(pcm), 0)
    ses.set_encoder(full)
    a = ses.decode(c["tokens"][:n_p], 0)
    ses2 = eng.new_session(); ses2.set_encoder(full)
    assert np.array_equal(a, ses2.decode(c["tokens"][:n_p], 0))
    report(f"ss_session_set_encoder_ctx + decode (f16) vs HF over {c['audio_ctx']} keys ({c['preset']}): top-16 logits of {len(c['tokens']) - n_p + 1} steps within {worst:.2e} sigma (tol {LOGIT_TOL_F16})")
    eng.close()



@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_tiny_contexts_through_encoder_attention_and_v_transpose(which, wide2_path, toy_ml_path):
    """audio_ctx 4, 8 and 64 -- less than one 64-key chunk of enc_attn_lds_kernel, less than one 256-row GEMM tile, a V^T row of mostly padding,
    cross-attention key ranges of one or two keys -- stage by stage against the oracle run over the same shortened context, at the tolerances of the
    full-context stage tests: the encoder rows, then the decoder's logits (prompt pass + 3 KV-cached steps) over those few keys; and at 64 keys a whole
    chunk with ids equal to the oracle's or proven near ties."""
    from speaksense_amd import binding
    from oracle import binding as orc
    from test_gpu_parity import GAP_TOL_BF16, GAP_TOL_F16, check_against_oracle
    dtype, omode, tol, ltol, gap = {"f16": (binding.DTYPE_F16, orc.MODE_GGML_F16, 6e-3, 6e-3, GAP_TOL_F16), "bf16": (binding.DTYPE_BF16, orc.MODE_BF16, 5e-2, 5e-2, GAP_TOL_BF16),
                                    "fp8": (binding.DTYPE_FP8, orc.MODE_FP8, 8e-2, 8e-2, 0.8)}[which]
    worst, worst_l = {}, {}
    for path in ([wide2_path] if which == "fp8" else [toy_ml_path, wide2_path]):        # the e4m3 engine needs n_audio_state % 256 == 0
        eng = binding.Engine(path, dtype=dtype, max_batch=2)
        om = orc.OracleModel(path)
        pcm = synth.speech_like(61, 16000 * 4)
        mel = om.log_mel(pcm)
        toks = [om.sot, om.sot + 1, om.transcribe, om.beg + 3, 1234, 777]
        for A in (4, 8, 64):
            ref = om.encode(mel, 0, omode, audio_ctx=A)
            got = eng.encode(mel, 0, audio_ctx=A)
            assert got.shape == ref.shape == (A, eng.n_audio_state) and np.isfinite(got).all()
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            worst[A] = max(worst.get(A, 0.0), err)
            assert err < tol, f"{os.path.basename(path)} {which} audio_ctx {A}: encoder rows {err:.2e} from the oracle"
            ost = om.new_state(omode); ost.set_encoder(ref)                 # both sides from the SAME encoder rows: the decoder over A keys alone
            ses = eng.new_session(); ses.set_encoder(ref)
            r, g = ost.decode(toks[:3], 0), ses.decode(toks[:3], 0)
            sd = float(r.std())
            e = float(np.abs(g - r).max()) / sd
            for i in range(3, len(toks)):
                r, g = ost.decode(toks[i:i + 1], i), ses.decode(toks[i:i + 1], i)
                e = max(e, float(np.abs(g - r).max()) / sd)
            worst_l[A] = max(worst_l.get(A, 0.0), e)
            assert e < ltol, f"{os.path.basename(path)} {which} audio_ctx {A}: decoder logits {e:.2e} sigma from the oracle"
            ses.close()
        kw = dict(language="en", temperature_inc=0.0, audio_ctx=64)
        res = eng.new_session().transcribe(pcm, binding.default_params(**kw))
        check_against_oracle(res, om, orc, omode, pcm, orc.default_params(**kw), f"{os.path.basename(path)} {which} audio_ctx 64", gap)
        om.close(); eng.close()
    report(f"tiny contexts ({which}): encoder rows vs oracle max|diff|/max " + ", ".join(f"audio_ctx {a}: {e:.1e}" for a, e in sorted(worst.items()))
           + "; decoder logits over those keys (sigma) " + ", ".join(f"{a}: {e:.1e}" for a, e in sorted(worst_l.items())) + "; a whole chunk at 64 keys equals the oracle or is a proven near tie")


@pytest.mark.parametrize("which", ["f16", "fp8"])
def test_mixed_context_batch_wide2(which, wide2_path):
    """a mixed-context group on the d = 1280 shape, f16 and e4m3: chunks asking for 0 / 64 / 752 / 8 / 1000 keys submitted together (one encoder pass per
    waiting context in the same round, the windows then share decoder passes) equal their single-chunk runs, forced fallbacks included."""
    from speaksense_amd import binding
    dtype = binding.DTYPE_F16 if which == "f16" else binding.DTYPE_FP8
    eng = binding.Engine(wide2_path, dtype=dtype, max_batch=8, n_lanes=1, batch_wait_us=300000)
    ctxs = [0, 64, 752, 8, 1000, 64, 0, 752]
    pcms = [synth.speech_like(70 + i, 16000 * 4) for i in range(len(ctxs))]
    Ps = [binding.default_params(language="en", audio_ctx=a, temperature_inc=0.2, logprob_thold=0.0) for a in ctxs]
    singles = [eng.new_session().transcribe(p, P) for p, P in zip(pcms, Ps)]
    t0 = eng.totals()
    ses = [eng.new_session() for _ in ctxs]
    tickets = [s.submit(p, P) for s, p, P in zip(ses, pcms, Ps)]
    res = [s.wait(t) for s, t in zip(ses, tickets)]
    t1 = eng.totals()
    for i, (r, s1) in enumerate(zip(res, singles)):
        assert [int(x) for x in r["trace"]] == [int(x) for x in s1["trace"]], f"{which}: chunk {i} (audio_ctx {ctxs[i] or 1500}) differs from its single run"
        assert np.array_equal(np.asarray(r["plog"]), np.asarray(s1["plog"]))
    rows = (t1["decoder_rows"] - t0["decoder_rows"]) / max(1, t1["decoder_passes"] - t0["decoder_passes"])
    assert rows > 2.0, "the mixed-context chunks never shared a decoder pass"
    report(f"mixed contexts on wide2 ({which}): 8 chunks (audio_ctx 0 / 64 / 752 / 8 / 1000) in one group equal their single runs; {rows:.1f} rows per decoder pass")
    eng.close()
