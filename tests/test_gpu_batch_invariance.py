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
     temperature ladder, whose draws depend on every probability bit)."""
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
