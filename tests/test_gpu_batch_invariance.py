"""GPU: BATCH INVARIANCE -- a decoder row's logits must not depend on what else rides in its pass.

`state.full(params, &audio)` (/root/reference/src/asr/whisper.rs:75) gives one answer per (state, audio): many states run concurrently on one
`Arc<WhisperContext>` (`whisper.rs:17,26`) and never see each other.  The batched engine carries rows of many sessions in one decoder pass and
picks kernels by the row count (1 / 2 / 4 / 8 column tiles per GEMV, LayerNorm-prologue GEMVs at <= 4 rows, one-workgroup cross-attention from
rows x heads >= 320): since round 5 every one of those forms performs the same floating-point operations on a row's operands in the same
order, so the results are BIT-identical.  Two kinds of test:

(i)  stage (`ss_engine_decode_rows`, raw logits, `np.array_equal`): one six-token sequence alone -- token by token (1-row passes), as 3 + 3,
     2 + 4 and 6 rows -- and inside passes of 8 .. 128 rows at different places, for f16 / bf16 / fp8 on a 20-head shape
     (wide2: the one-workgroup cross-attention starts at 16 rows) and a 4-head shape (toy256: starts at 80 rows);
(ii) whole path: chunks transcribed alone and among 15 others give identical token ids, log-probs and segments (greedy AND the sampled
     temperature ladder, whose draws depend on every probability bit);
(iii) `test_launcher_thresholds` (round 6): EVERY host-side selection that looks at a batch-shaped quantity, both sides of its threshold -- the
     round-5 soak bug (a GEMM kernel picked by M, commit 61ddfac) is the template.  Decoder: LayerNorm-prologue GEMVs (rows <= 4), one / two column
     tiles (16 | 17), row groups in grid.z (32 | 33, 64 | 65, 96 | 97), the one-workgroup cross-attention (rows x heads >= 320), the pass limit (127 | 128).
     Encoder: M = windows x positions only changes how many 256-row tiles a launch has (no kernel is selected by it any more): one window alone against the
     same window among 1 .. 7 others, full and shortened context, decoded with forced fallbacks so that every low bit reaches the trace."""
import numpy as np
import pytest

from speaksense_amd import synth
from conftest import report

pytestmark = pytest.mark.gpu


def _dtype(binding, which):
    return {"f16": binding.DTYPE_F16, "bf16": binding.DTYPE_BF16, "fp8": binding.DTYPE_FP8}[which]


@pytest.mark.parametrize("shape", ["wide2", "toy256"])
@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_decoder_row_bits_do_not_depend_on_the_pass(which, shape, wide2_path, toy256_path):
    from speaksense_amd import binding
    path = {"wide2": wide2_path, "toy256": toy256_path}[shape]
    eng = binding.Engine(path, dtype=_dtype(binding, which), max_batch=16, max_decoders=5)      # 80 self-KV slots, 16 cross-KV windows
    sot, transcribe = eng.sot, eng.transcribe
    n_win = 4
    for w in range(n_win):
        eng.set_encoder_window(w, eng.encode(eng.log_mel(synth.speech_like(300 + w)), 0))
    rng = np.random.default_rng(11)
    target = [sot, sot + 1, transcribe, 1234, 20000, 777]           # the sequence under test: self-KV slot 3, cross-KV window 1
    T_SLOT, T_WIN = 3, 1

    def filler(n_rows, slots):
        """n_rows rows of other sequences (3 positions each, the last one possibly shorter), each in a self-KV slot of its own"""
        tok, pos, slot, cross = [], [], [], []
        while len(tok) < n_rows:
            s = next(slots)
            for i in range(min(3, n_rows - len(tok))):
                tok.append(int(rng.integers(300, 40000))); pos.append(i); slot.append(s); cross.append(s % n_win)
        return tok, pos, slot, cross

    def run(n_before, n_after, t_from, t_to):
        """rows [t_from, t_to) of the target inside one pass with n_before / n_after filler rows around them -> logits of the target rows"""
        slots = iter([s for s in range(80) if s != T_SLOT])
        b = filler(n_before, slots)
        a = filler(n_after, slots)
        tok = b[0] + target[t_from:t_to] + a[0]
        pos = b[1] + list(range(t_from, t_to)) + a[1]
        slot = b[2] + [T_SLOT] * (t_to - t_from) + a[2]
        cross = b[3] + [T_WIN] * (t_to - t_from) + a[3]
        samp = list(range(n_before, n_before + t_to - t_from))
        return eng.decode_rows(tok, pos, slot, cross, samp)

    n = len(target)
    # reference: the target alone, one row per pass (LayerNorm-prologue GEMVs, key-split cross-attention + combine kernel)
    ref = np.concatenate([run(0, 0, i, i + 1) for i in range(n)])
    assert np.isfinite(ref).all() and float(ref.std()) > 1e-3
    cases = {
        "3 + 3 rows alone": [(0, 0, 0, 3), (0, 0, 3, 6)],
        "2 + 4 rows alone": [(0, 0, 0, 2), (0, 0, 2, 6)],
        "6 rows alone": [(0, 0, 0, 6)],
        "last of 8 rows": [(2, 0, 0, 6)],
        "16 rows": [(5, 5, 0, 6)],
        "17 rows (2 column tiles)": [(11, 0, 0, 6)],
        "32 rows": [(7, 19, 0, 6)],
        "33 rows (4 column tiles)": [(27, 0, 0, 6)],
        "64 rows": [(40, 18, 0, 6)],
        "80 rows (8 column tiles)": [(1, 73, 0, 6)],
        "128 rows": [(100, 22, 0, 6)],
        "steps inside 32-row passes": [(10, 21, i, i + 1) for i in range(n)],
        "steps inside 100-row passes": [(50, 49, i, i + 1) for i in range(n)],
    }
    for name, passes in cases.items():
        got = np.concatenate([run(*p) for p in passes])
        same = np.array_equal(got, ref)
        if not same:
            bad = np.argwhere(got != ref)
            d = float(np.abs(got - ref).max()) / float(ref.std())
            raise AssertionError(f"{shape} {which}, {name}: {len(bad)} logits differ from the one-row passes (first at row {bad[0][0]}, max |diff| / std {d:.2e})")
    report(f"batch invariance, {shape} {which}: the target sequence's {n} x {ref.shape[1]} raw logits are bit-identical across {len(cases)} pass compositions "
           f"(1 .. 128 rows: 1 / 2 / 4 / 8 column tiles, LayerNorm-prologue GEMVs, both cross-attention forms)")
    eng.close()


@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_chunk_alone_equals_chunk_among_others(which, wide2_path):
    """Whole path, real parameters (best_of 5, temperature ladder: the default synthetic weights walk the ladder on nearly every window, so the
    sampled attempts -- five decoders per window, draws that depend on every bit of the probability rows -- are covered too)."""
    from speaksense_amd import binding
    eng = binding.Engine(wide2_path, dtype=_dtype(binding, which), max_batch=16, n_lanes=1, batch_wait_us=300000)
    pcms = [synth.speech_like(900 + i) for i in range(16)]
    for name, P in (("greedy, 24 fixed steps", binding.default_params(language="en", fixed_steps=24)),
                    ("reference parameters (ladder)", binding.default_params(language="en"))):
        ses = [eng.new_session() for _ in pcms]
        tickets = [s.submit(p, P) for s, p in zip(ses, pcms)]
        res = [s.wait(t) for s, t in zip(ses, tickets)]
        t0 = eng.totals()
        n_diff = 0
        for i in (0, 5, 11, 15):
            single = eng.new_session().transcribe(pcms[i], P)
            ok = (list(single["tokens"]) == list(res[i]["tokens"]) and np.array_equal(np.asarray(single["plog"]), np.asarray(res[i]["plog"])) and
                  [(s["t0"], s["t1"], s["text"]) for s in single["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in res[i]["segments"]])
            n_diff += 0 if ok else 1
        assert n_diff == 0, f"{which}, {name}: {n_diff} of 4 chunks differ between their single-chunk run and the 16-chunk batch"
        assert eng.totals()["decoder_passes"] > t0["decoder_passes"]
    report(f"batch invariance, wide2 {which}: 4 chunks of 16 give identical ids, log-probs and segments alone and in the batch (greedy and ladder)")
    eng.close()



def _thresholds(n_head):
    direct = -(-320 // n_head)          # engine.cpp direct_pairs = 320 (rows x heads)
    t = {"LayerNorm-prologue GEMVs (kLnFuseRows)": 4, "one | two column tiles": 16, "row groups (grid.z) 1 | 2": 32, "row groups 2 | 3": 64,
         "row groups 3 | 4": 96, "cross-attention: key splits + combine | one workgroup per (row, head)": direct - 1, "pass limit (kPartRows)": 127}
    return {k: v for k, v in t.items() if 1 <= v < 128}


@pytest.mark.parametrize("shape", ["wide2", "toy256"])
@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_launcher_thresholds(which, shape, wide2_path, toy256_path):
    """Decoder side: a two-step target sequence (prompt row, then one KV-cached row) decoded in passes of exactly t and t + 1 rows for every threshold t,
    once as the first and once as the last rows of the pass: raw logits bit-identical to the one-row passes."""
    from speaksense_amd import binding
    path = {"wide2": wide2_path, "toy256": toy256_path}[shape]
    eng = binding.Engine(path, dtype=_dtype(binding, which), max_batch=16, max_decoders=8)      # 128 self-KV slots
    n_win = 4
    for w in range(n_win):
        eng.set_encoder_window(w, eng.encode(eng.log_mel(synth.speech_like(320 + w)), 0))
    rng = np.random.default_rng(5)
    target = [eng.sot, 4242]
    T_SLOT, T_WIN = 7, 2

    def run(n_rows, first, i):
        """row i of the target inside a pass of n_rows rows (every other row a one-token sequence in a slot of its own), first or last in the pass"""
        others = [s for s in range(128) if s != T_SLOT][:n_rows - 1]
        tok = [int(rng.integers(300, 40000)) for _ in others]; pos = [0] * len(others); slot = list(others); cross = [s % n_win for s in others]
        at = 0 if first else len(others)
        tok.insert(at, target[i]); pos.insert(at, i); slot.insert(at, T_SLOT); cross.insert(at, T_WIN)
        return eng.decode_rows(tok, pos, slot, cross, [at])

    ref = [run(1, True, 0), run(1, True, 1)]
    assert np.isfinite(ref[0]).all() and np.isfinite(ref[1]).all() and not np.array_equal(ref[0], ref[1])
    n_checked = 0
    for name, t in _thresholds(eng.n_text_head).items():
        for n_rows in (t, t + 1):
            for first in (True, False):
                for i in (0, 1):        # position 0 first: the cached row 1 attends to it
                    got = run(n_rows, first, i)
                    assert np.array_equal(got, ref[i]), (f"{shape} {which}: {name}: target row {i} as the {'first' if first else 'last'} of {n_rows} rows differs from "
                                                         f"its one-row pass in {int((got != ref[i]).sum())} logits")
                    n_checked += 1
    report(f"launcher thresholds, {shape} {which}: {n_checked} passes on both sides of {len(_thresholds(eng.n_text_head))} row-count thresholds, target logits bit-identical to the one-row passes")
    eng.close()


@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_launcher_thresholds_encoder_windows_per_pass(which, wide2_path):
    """Encoder side: the same window alone and among 1 / 3 / 7 DIFFERENT windows in one encoder pass (M = 1 .. 8 x positions: 6 .. 47 row tiles, the last
    one partial), at the full context and at a shortened one, decoded with forced temperature fallbacks (five sampling decoders walk every bit)."""
    from speaksense_amd import binding
    eng = binding.Engine(wide2_path, dtype=_dtype(binding, which), max_batch=8, n_lanes=1, batch_wait_us=300000)
    n_checked = 0
    for actx in (0, 752):
        P = binding.default_params(language="en", audio_ctx=actx, temperature_inc=0.2, logprob_thold=0.0)
        X = synth.speech_like(41, 16000 * 4)
        alone = eng.new_session().transcribe(X, P)
        assert alone["n_fail"] >= 1
        for n_other in (1, 3, 7):
            ss = [eng.new_session() for _ in range(n_other + 1)]
            pcms = [X] + [synth.speech_like(600 + k, 16000 * 4) for k in range(n_other)]
            t0 = eng.totals()
            ts = [s_.submit(p_, P) for s_, p_ in zip(ss, pcms)]
            rs = [s_.wait(t_) for s_, t_ in zip(ss, ts)]
            assert [int(x) for x in rs[0]["trace"]] == [int(x) for x in alone["trace"]], f"{which} audio_ctx {actx or 1500}: window among {n_other} others differs from its single run"
            assert np.array_equal(np.asarray(rs[0]["plog"]), np.asarray(alone["plog"]))
            n_checked += 1
    report(f"launcher thresholds, encoder ({which}): {n_checked} windows among 1 / 3 / 7 others (full and 752-position context, forced fallbacks) equal their single runs")
    eng.close()
