"""Token-level timestamps on the MI355X path (ss_params.token_timestamps, /root/reference/src/asr/whisper.rs:160,170-171): the signal-energy kernel
bit for bit against the oracle, the engine's host pass against the oracle's on the engine's own token data (exact), and the whole call against the
oracle's whole call (exact wherever both sides sampled the same timestamp evidence)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import report
from speaksense_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def natural256_path(model_dir):
    from speaksense_amd import ggml_io
    path = os.path.join(model_dir, "toy256-natural-s0.bin")
    if not os.path.exists(path):
        ggml_io.write_model(path, "toy256", seed=0, **ggml_io.NATURAL)
    return path


def test_signal_energy_kernel_is_bit_exact(toy_ml_path):
    """get_signal_energy(pcm, n, 32): 65 sequential f32 adds and one division per sample -- the device result equals the CPU's bit for bit, at the
    edges (windows clipped to the signal), for lengths around the block size and for a full 30 s chunk."""
    from oracle import binding as orc
    from speaksense_amd import binding
    eng = binding.Engine(toy_ml_path, max_batch=1)
    rng = np.random.default_rng(3)
    for n in (1, 2, 31, 33, 64, 65, 255, 256, 257, 1000, 16000 * 30, 16000 * 30 + 77):
        x = (rng.standard_normal(n) * rng.uniform(0.01, 0.9)).astype(np.float32)
        if n > 1000:
            x[n // 3: n // 2] = 0.0                                  # a silent stretch: exact zeros stay exact zeros
        got, want = eng.signal_energy(x), orc.signal_energy(x)
        assert np.array_equal(got, want), (n, float(np.abs(got - want).max()))
    eng.close()


def _key_times(tt):
    return [(int(a), int(b)) for a, b in zip(tt["t0"], tt["t1"])]


@pytest.mark.parametrize("model", ["natural256", "toy_ml"])
def test_engine_token_times_equal_the_oracle_pass_on_the_same_tokens(model, natural256_path, toy_ml_path):
    """For every segment of every chunk -- short, full-window and multi-window audio; the reference's parameters and single_segment / no_timestamps /
    max_tokens variants -- the engine's t0 / t1 / vlen equal what the oracle's pass computes from the SAME (id, tid, pt, ptsum) and the same PCM,
    with the state (t_beg / t_last / tid_last) carried through the chunk's segments.  Then the whole call against the oracle's whole call."""
    from oracle import binding as orc
    from speaksense_amd import binding
    path = natural256_path if model == "natural256" else toy_ml_path
    om = orc.OracleModel(path)
    eng = binding.Engine(path, max_batch=4)
    cases = [(101, 16000 * 30, {}), (102, 16000 * 3, {}), (103, 16000 * 65, {}), (104, 16000 * 30, dict(single_segment=1)),
             (105, 16000 * 30, dict(no_timestamps=1)), (106, 16000 * 12, dict(max_tokens=9)), (107, 16000 * 30, dict(thold_pt=0.5, thold_ptsum=0.5))]
    n_seg = n_tok = n_tok_full = n_same_full = 0
    for seed, n, kw in cases:
        pcm = synth.speech_like(seed, n)
        P = binding.default_params(language="en", temperature_inc=0.0, **kw)
        ses = eng.new_session()
        got = ses.transcribe(pcm, P)
        tt = ses.token_times()
        assert len(tt) == len(got["segments"])
        want = om.token_times_chunk(pcm, [(s["t0"], s["t1"]) for s in got["segments"]], tt, thold_pt=P.thold_pt, thold_ptsum=P.thold_ptsum)
        for i, (g, w) in enumerate(zip(tt, want)):
            assert _key_times(g) == _key_times(w), f"{model} seed {seed} {kw}: segment {i}: engine {_key_times(g)} oracle pass {_key_times(w)}"
            assert np.array_equal(g["vlen"], w["vlen"])
            assert (g["t0"] >= 0).all() and (g["t1"] >= g["t0"]).all()
            n_tok += len(g["ids"])
        n_seg += len(tt)
        # whole call: the oracle decodes on its own (forced onto the engine's ids where a near tie flipped); its timestamp evidence (pt, ptsum, tid)
        # is computed in f32 on the CPU, so a token whose evidence sits at a threshold may be anchored on one side only
        OP = orc.default_params(language="en", temperature_inc=0.0, **kw)
        rep = om.new_state(orc.MODE_GGML_F16).full(pcm, OP, forced=got["sampled"])
        if [(s["t0"], s["t1"]) for s in rep["segments"]] == [(s["t0"], s["t1"]) for s in got["segments"]]:
            for g, s in zip(tt, rep["segments"]):
                w = s["token_times"]
                if list(w["ids"]) == list(g["ids"]):
                    n_tok_full += len(g["ids"])
                    n_same_full += sum(a == b for a, b in zip(_key_times(g), _key_times(w)))
        ses.close()
    report(f"token-level timestamps [{model}]: {n_seg} segments, {n_tok} tokens: engine pass == oracle pass on the engine's token data (exact); "
           f"whole call vs the oracle's whole call: {n_same_full}/{n_tok_full} tokens with identical (t0, t1)")
    assert n_tok >= 60 and n_tok_full >= 0.5 * n_tok
    assert n_same_full >= 0.9 * n_tok_full, (n_same_full, n_tok_full)
    eng.close(); om.close()


def test_flag_off_leaves_minus_one_and_changes_nothing_else(natural256_path):
    from speaksense_amd import binding
    eng = binding.Engine(natural256_path, max_batch=2)
    pcm = synth.speech_like(110)
    a, b = eng.new_session(), eng.new_session()
    on = a.transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0))
    off = b.transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, token_timestamps=0))
    assert list(on["tokens"]) == list(off["tokens"])
    assert [(s["text"], s["t0"], s["t1"]) for s in on["segments"]] == [(s["text"], s["t0"], s["t1"]) for s in off["segments"]]
    for g in b.token_times():
        assert (g["t0"] == -1).all() and (g["t1"] == -1).all() and (g["vlen"] == 0).all()
    assert sum(len(g["ids"]) for g in a.token_times()) > 10 and all((g["t0"] >= 0).all() for g in a.token_times())
    a.close(); b.close(); eng.close()


def test_batched_chunks_get_the_serial_token_times(natural256_path):
    """Eight chunks of different lengths in one device batch (PCM resident on the device for half of them): per-slot energy buffers, per-session state --
    every chunk's token times equal those of the same chunk run alone."""
    from speaksense_amd import binding
    hip = C.CDLL("libamdhip64.so")     # the runtime the library itself is linked against (torch in this process would bring a second copy)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    eng = binding.Engine(natural256_path, max_batch=8)
    P = binding.default_params(language="en", temperature_inc=0.0)
    pcms = [synth.speech_like(120 + i, 16000 * (4 + 7 * i)) for i in range(8)]
    alone = []
    for p in pcms:
        s = eng.new_session(); s.transcribe(p, P); alone.append([_key_times(g) for g in s.token_times()]); s.close()
    ses = [eng.new_session() for _ in pcms]
    dev = []
    for i, p in enumerate(pcms):
        d = C.c_void_p()
        if i % 2:
            assert hip.hipMalloc(C.byref(d), p.nbytes) == 0 and hip.hipMemcpy(d, p.ctypes.data_as(C.c_void_p), p.nbytes, 1) == 0   # hipMemcpyHostToDevice
        dev.append(d)
    tickets = [s.submit_device(d.value, len(p), P) if d.value else s.submit(p, P) for s, p, d in zip(ses, pcms, dev)]
    for s, t, want in zip(ses, tickets, alone):
        s.wait(t)
        assert [_key_times(g) for g in s.token_times()] == want
        s.close()
    eng.close()
    for d in dev:
        if d.value:
            hip.hipFree(d)


def test_whisper_h_token_data_carries_the_times(natural256_path):
    """The whisper.h shim: whisper_full_get_token_data(...).t0 / t1 / vlen are the session's, switched by whisper_full_params.token_timestamps
    (false in whisper_full_default_params, set by the reference: whisper.rs:160)."""
    from speaksense_amd import binding
    from test_gpu_variants import WCtxParams, WFullParams, WTokenData
    L = C.CDLL(binding.lib()._name)
    vp = C.c_void_p
    L.whisper_context_default_params.restype = WCtxParams
    L.whisper_init_from_file_with_params_no_state.restype = vp
    L.whisper_init_from_file_with_params_no_state.argtypes = [C.c_char_p, WCtxParams]
    L.whisper_init_state.restype = vp
    L.whisper_init_state.argtypes = [vp]
    L.whisper_full_default_params.restype = WFullParams
    L.whisper_full_default_params.argtypes = [C.c_int]
    L.whisper_full_with_state.argtypes = [vp, vp, WFullParams, vp, C.c_int]
    L.whisper_full_n_segments_from_state.argtypes = [vp]
    L.whisper_full_n_tokens_from_state.argtypes = [vp, C.c_int]
    L.whisper_full_get_token_data_from_state.restype = WTokenData
    L.whisper_full_get_token_data_from_state.argtypes = [vp, C.c_int, C.c_int]
    L.whisper_free_state.argtypes = [vp]
    L.whisper_free.argtypes = [vp]
    pcm = synth.speech_like(130, 16000 * 10)
    ctx = L.whisper_init_from_file_with_params_no_state(natural256_path.encode(), L.whisper_context_default_params())
    assert ctx
    st = L.whisper_init_state(ctx)
    eng = binding.Engine(natural256_path, max_batch=1)
    try:
        for flag in (True, False):
            p = L.whisper_full_default_params(0)
            p.temperature_inc = 0.0; p.print_progress = False; p.print_timestamps = False; p.token_timestamps = flag
            assert L.whisper_full_with_state(ctx, st, p, pcm.ctypes.data_as(vp), len(pcm)) == 0
            ses = eng.new_session()
            ses.transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, token_timestamps=int(flag)))
            want = ses.token_times()
            ses.close()
            n = L.whisper_full_n_segments_from_state(st)
            assert n == len(want) and n > 0
            n_tok = 0
            for i in range(n):
                assert L.whisper_full_n_tokens_from_state(st, i) == len(want[i]["ids"])
                for k in range(len(want[i]["ids"])):
                    d = L.whisper_full_get_token_data_from_state(st, i, k)
                    assert (d.id, d.t0, d.t1, d.vlen) == (int(want[i]["ids"][k]), int(want[i]["t0"][k]), int(want[i]["t1"][k]), float(want[i]["vlen"][k]))
                    assert (d.t0 >= 0) == flag
                    n_tok += 1
            assert n_tok > 5
    finally:
        eng.close()
        L.whisper_free_state(st)
        L.whisper_free(ctx)
