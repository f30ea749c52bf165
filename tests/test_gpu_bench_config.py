"""GPU: the engine configuration bench.py MEASURES, under the oracle.

bench.py's default line runs `Engine(max_batch=32, n_lanes=3)` on the real large-v3 shape (BASELINE.json configs[2], SURVEY.md section 8d
config #3; reference call site /root/reference/src/asr/whisper.rs:75).  At 27-32 rows per decoder pass the engine takes kernel paths no
8-row test reaches: rows x heads >= 320 switches the cross-attention to its one-workgroup form (`dec_cross_attn_q_kernel<T,4>`, the fp8 engine's
`dec_cross_attn_q8_kernel<T,4>`: the same four key ranges and combine as the split form, merged in-kernel), and every projection runs through the multi-tile GEMVs (`dec_gemv_kernel<T,EPI,CT,NFR>` with CT = 2 column tiles for 17..32 rows,
CT = 4 for 33..64) -- together ~45 % of the benchmark's GPU time.  Two kinds of test, each for f16 / bf16 / fp8:

(i)  stage: `ss_engine_decode_rows` -- ONE decoder pass over 32 and over 64 rows (sequences of 8 prompt positions each, attending to different
     cross-KV windows) against the oracle's logits for each sequence, at full depth (32 layers);
(ii) whole path: 32 chunks submitted asynchronously in Mode F, one lane carries them as 32-row passes (asserted from the engine's own
     counters), every chunk's ids must equal its single-chunk run (8-row kernels) or the difference must be a near tie proven on the oracle,
     and chunks of the batch are force-replayed on the oracle."""
import os
import sys

import numpy as np
import pytest

from speaksense_amd import synth
from conftest import SLOW, report, shared_oracle_model
from test_gpu_parity import GAP_TOL_BF16, GAP_TOL_F16, check_against_oracle
from test_gpu_fp8 import GAP_TOL_FP8

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_REPLAY = {"f16": 2, "bf16": 2, "fp8": 1} if SLOW else {"f16": 1, "bf16": 1, "fp8": 1}          # oracle windows per dtype (each ~15 s on 64 host threads); SS_RUN_SLOW=1: the round-4 counts
# fp8 at FULL depth: 32 e4m3 encoder layers put the encoder output 8.5e-3 rms / up to 8e-2 of full scale from the FP8-mode oracle (test_gpu_fp8.py),
# and a pick's margin inherits that tail: r04_g measured 0.479 on one chunk of 32 (its single-chunk run, whose few-row kernels round differently,
# happened to agree with the oracle there, which is what sent it to the replay).  There is no reference arithmetic for this mode:
# the bound is 4 x the measured fp8 logit noise (1.6e-2 sigma x 9).
GAP_TOL_FP8_FULL_DEPTH = 0.6


@pytest.fixture(scope="module")
def large_v3_path():
    sys.path.insert(0, ROOT)
    import bench
    from speaksense_amd import ggml_io
    path = bench.model_path_for("large-v3")
    if not os.path.exists(path):
        ggml_io.write_model(path + ".tmp", "large-v3", seed=0)
        os.replace(path + ".tmp", path)
    return path


@pytest.fixture(scope="module")
def oracle_threads():
    from oracle import binding as orc
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    orc.set_thread_cap(min(64, ncpu))
    yield orc
    orc.set_thread_cap(16)


def _modes(orc, which):
    from speaksense_amd import binding
    return {"f16": (binding.DTYPE_F16, orc.MODE_GGML_F16, GAP_TOL_F16, 6e-3),
            "bf16": (binding.DTYPE_BF16, orc.MODE_BF16, GAP_TOL_BF16, 5e-2),
            "fp8": (binding.DTYPE_FP8, orc.MODE_FP8, GAP_TOL_FP8_FULL_DEPTH, 5e-2)}[which]


@pytest.fixture(scope="module", params=["f16", "bf16", "fp8"])
def bench_engine(request, large_v3_path):
    """What bench.py creates (bench.py: `--device-batch 32 --lanes 3`).  One option differs: the batch former may wait 0.5 s for a full batch
    (bench.py hands over device pointers in microseconds; this test copies host PCM at submit, and a former that only waits the default
    2 ms would start with the handful of chunks queued by then and spread the rest over the other lanes)."""
    from speaksense_amd import binding
    from oracle import binding as orc
    dtype = _modes(orc, request.param)[0]
    eng = binding.Engine(large_v3_path, dtype=dtype, max_batch=32, n_lanes=3, batch_wait_us=500000)
    yield request.param, eng
    eng.close()


def test_decoder_pass_32_64_and_128_rows_vs_oracle(bench_engine, large_v3_path, oracle_threads):
    """(i) One decoder pass of the benchmarked shape against the oracle.  4 (then 8) sequences x 8 prompt positions = 32 (64) rows in ONE launch;
    sequence s sits in self-KV slot 5 s and attends to cross-KV window s % 4, whose cache was filled from an encoder-output matrix that
    the oracle state of the same sequence gets too.  The logits of every sequence's last row are held to the per-step tolerance of the
    8-row stage tests (f16 6e-3 sigma; bf16 / fp8 5e-2), and the argmax must agree wherever the oracle's own top-2 is outside that noise."""
    orc = oracle_threads
    which, eng = bench_engine
    _, omode, _, tol = _modes(orc, which)
    om = shared_oracle_model(large_v3_path)
    rng = np.random.default_rng(7)
    n_win = 4
    # encoder outputs of four different chunks (from the device's own encoder: any matrix would do, both sides get the same one, but the
    # conditioning must be that of real ln_post outputs -- with white noise of unit scale the oracle's own f16 and f32 modes are 1e-1 sigma apart)
    encs = [eng.encode(om.log_mel(synth.speech_like(200 + w)), 0) for w in range(n_win)]
    for w in range(n_win):
        eng.set_encoder_window(w, encs[w])
    text = rng.integers(300, 40000, (8, 8))
    worst = {}
    for n_seq in (4, 8, 16):       # 32, 64 and (round 4: CT = 8 column tiles) 128 rows in one pass
        token, pos, slot, cross, samp = [], [], [], [], []
        seqs = []
        for s in range(n_seq):
            toks = [om.sot, om.sot + 1 + s, om.transcribe] + [int(t) for t in text[s % 8][:5]]
            seqs.append(toks)
            for i, t in enumerate(toks):
                token.append(t); pos.append(i); slot.append(5 * s); cross.append(s % n_win)
            samp.append(len(token) - 1)
        assert len(token) == 8 * n_seq
        got = eng.decode_rows(token, pos, slot, cross, samp)
        w_n = 0.0
        for s in (range(n_seq) if n_seq <= 4 else (0, 3, 5, 7) if n_seq <= 8 else (0, 7, 9, 15)):     # the oracle costs seconds per sequence: a sample of the wide passes
            ost = om.new_state(omode)
            ost.set_encoder(encs[s % n_win])
            ref = ost.decode(seqs[s], 0)
            ost.close()
            sd = float(ref.std())
            e = float(np.abs(got[s] - ref).max()) / sd
            w_n = max(w_n, e)
            assert e < tol, f"{which}, {8 * n_seq}-row pass, sequence {s}: max|logits - oracle| / std = {e}"
            top2 = np.sort(ref)[-2:]
            if top2[1] - top2[0] > 2 * tol * sd:
                assert int(got[s].argmax()) == int(ref.argmax()), f"{which}, {8 * n_seq}-row pass, sequence {s}"
        worst[8 * n_seq] = w_n
    # the same sequence alone (8 rows: one column tile per GEMV, key-split cross-attention + combine) against its row of the wide pass: the row count
    # changes the kernels and with them the accumulation order, not the arithmetic type -- the two passes differ by the f16 noise floor (measured
    # 1.85e-3 sigma, the distance each has from the oracle) and must stay within half the oracle tolerance
    s = 1
    toks = [om.sot, om.sot + 1 + s, om.transcribe] + [int(t) for t in text[s % 8][:5]]
    alone = eng.decode_rows(toks, list(range(8)), [5 * s] * 8, [s % n_win] * 8, [7])
    d = float(np.abs(alone[0] - got[s]).max()) / float(got[s].std())
    assert d < tol / 2, d
    report(f"large-v3 {which} decoder pass vs oracle at the benchmarked row counts (one-workgroup cross-attention, multi-tile GEMVs): worst max|logits - oracle| / std "
           f"= {worst[32]:.2e} at 32 rows, {worst[64]:.2e} at 64 rows, {worst[128]:.2e} at 128 rows; 8-row pass vs 128-row pass {d:.2e}")
    om.close()


def test_bench_engine_32_row_passes_vs_oracle(bench_engine, large_v3_path, oracle_threads):
    """(ii) 32 chunks through ss_submit / ss_wait, Mode F (32 greedy steps): the configuration whose throughput BENCH_r*.json reports."""
    from speaksense_amd import binding
    orc = oracle_threads
    which, eng = bench_engine
    _, omode, gap_tol, _ = _modes(orc, which)
    P = binding.default_params(language="en", fixed_steps=32)
    pcms = [synth.speech_like(100 + i) for i in range(32)]
    t0 = eng.totals()
    ses = [eng.new_session() for _ in pcms]
    tickets = [s.submit(p, P) for s, p in zip(ses, pcms)]     # host PCM (1.9 MB copied per submit): the batch former's wait below covers the ~30 ms
    res = [s.wait(t) for s, t in zip(ses, tickets)]
    t1 = eng.totals()
    passes, rows = t1["decoder_passes"] - t0["decoder_passes"], t1["decoder_rows"] - t0["decoder_rows"]
    assert passes > 0 and rows / passes >= 24, f"the batch former split the chunks: {rows} rows over {passes} passes"
    assert all(len(r["tokens"]) == 32 and r["n_encode"] == 1 and r["n_fail"] == 0 for r in res)
    n_distinct = len({tuple(r["tokens"]) for r in res})
    assert n_distinct > 1        # the audio matters (random weights: greedy streams fall into a handful of attractors, so not 32 distinct ones)
    om = shared_oracle_model(large_v3_path)
    OP = orc.default_params(language="en", fixed_steps=32)
    tid_slack = om.beg if which == "fp8" else None
    # every chunk against its own single-chunk run (1-3 rows per pass: the <= 16-row kernels and the key-split cross-attention)
    n_same, differing = 0, []
    for i, p in enumerate(pcms):
        single = eng.new_session().transcribe(p, P)
        if list(single["tokens"]) == list(res[i]["tokens"]):
            n_same += 1
        else:
            differing.append(i)
    # batch invariance (round 5, tests/test_gpu_batch_invariance.py): a row's bits do not depend on the pass it rides in, so this is an equality
    assert n_same == 32, f"{which}: only {n_same}/32 chunks equal their single-chunk run: {differing}"
    replay = [3, 17][:N_REPLAY[which]]
    worst = 0.0
    for i in replay:
        _, gap = check_against_oracle(res[i], om, orc, omode, pcms[i], OP, f"large-v3 {which} bench config, chunk {i}", gap_tol, replay_only=True,
                                      tid_slack_beg=tid_slack)
        worst = max(worst, gap)
    report(f"large-v3 {which}, Engine(max_batch=32, n_lanes=3), 32 chunks async: {rows / passes:.1f} rows per decoder pass; {n_distinct} distinct token streams; {n_same}/32 chunks identical to "
           f"their single-chunk runs; chunks {replay} force-replayed on the oracle, largest near-tie margin {worst:.4f} (tolerance {gap_tol})")
    om.close()


@pytest.fixture(scope="module")
def large_v3_natural_path():
    sys.path.insert(0, ROOT)
    import bench
    from speaksense_amd import ggml_io
    path = bench.model_path_for("large-v3-natural")
    if not os.path.exists(path):
        ggml_io.write_model(path + ".tmp", "large-v3", seed=0, **ggml_io.NATURAL)
        os.replace(path + ".tmp", path)
    return path


def test_natural_preset_32_chunks_distinct_streams_vs_oracle(large_v3_natural_path, oracle_threads):
    """The whole path at full depth with the reference's REAL parameters (whisper.rs:131-173: best_of 5, temperature ladder, decode to EOT) on the
    natural-EOT preset (ggml_io.NATURAL, what bench.py's `mode_n` runs): 32 different chunks through Engine(max_batch=32, n_lanes=3) must give
    >= 24 DIFFERENT token streams (with the default synthetic weights 32 chunks fell into 5 attractors, so a cross-KV
    slot mix-up between rows could hide behind coinciding streams), >= 90 % of the windows must stay at temperature 0, the transcript lengths
    must spread, every chunk must equal its single-chunk run, and two chunks are replayed call by call on the oracle."""
    from speaksense_amd import binding
    from test_gpu_parity import check_trace_against_oracle
    orc = oracle_threads
    eng = binding.Engine(large_v3_natural_path, dtype=binding.DTYPE_F16, max_batch=32, n_lanes=3, batch_wait_us=500000)
    P = binding.default_params(language="en")
    pcms = [synth.speech_like(5000 + i) for i in range(32)]
    ses = [eng.new_session() for _ in pcms]
    tickets = [s.submit(p, P) for s, p in zip(ses, pcms)]
    res = [s.wait(t) for s, t in zip(ses, tickets)]
    lens = [len(r["tokens"]) for r in res]
    n_win, n_fail = sum(r["n_windows"] for r in res), sum(r["n_fail"] for r in res)
    streams = {tuple(int(t) for t in r["tokens"]) for r in res}
    assert len(streams) >= 24, f"only {len(streams)} distinct token streams among 32 chunks"
    assert n_fail <= 0.1 * n_win, f"{n_fail} fallbacks over {n_win} windows"
    assert max(lens) - min(lens) >= 30 and min(lens) >= 2, lens
    check = list(range(32)) if SLOW else list(range(0, 32, 3)) + [31]       # every chunk with SS_RUN_SLOW=1; 12 of the 32 (~0.4 s each, one at a time) under the driver
    singles = {i: eng.new_session().transcribe(pcms[i], P) for i in check}
    differing = [i for i in check if list(singles[i]["tokens"]) != list(res[i]["tokens"])]
    n_same = len(check) - len(differing)
    assert not differing, f"only {n_same}/{len(check)} chunks equal their single-chunk run: {differing}"        # batch invariance (round 5)
    om = shared_oracle_model(large_v3_natural_path)
    order = sorted(range(32), key=lambda i: (res[i]["n_windows"], lens[i]))
    # the cheapest non-trivial chunks for the CPU oracle (~25 s per window on 64 threads): one by default, three with SS_RUN_SLOW=1 (the round-4 count).
    # The same model is also held to HF `generate` at full depth below and the f16 decoder arithmetic to the oracle in test_gpu_large_v3.py.
    picked = [i for i in order if lens[i] >= 8][:3 if SLOW else 1]
    worst = 0.0
    for i in picked:
        fg, fs, wg, ws = check_trace_against_oracle(res[i], om, orc, orc.MODE_GGML_F16, pcms[i], orc.default_params(language="en"),
                                                    f"large-v3 natural preset, chunk {i}", GAP_TOL_F16)
        worst = max(worst, wg)
    report(f"large-v3 natural preset, 32 chunks async on Engine(32, 3 lanes): {len(streams)} distinct streams, {n_win} windows, {n_fail} fallbacks, tokens per chunk "
           f"{min(lens)}..{max(lens)} (median {int(np.median(lens))}), {n_same}/{len(check)} equal their single-chunk runs; chunks {picked} replayed call by call on the oracle, "
           f"largest greedy margin {worst:.4f}")
    om.close(); eng.close()


@pytest.mark.parametrize("which", ["f16", "bf16", "fp8"])
def test_natural_preset_first_windows_match_hf_generate(which, large_v3_natural_path, oracle_threads):
    """Sequence-level pin at FULL depth: HF transformers' `generate(return_timestamps=True, do_sample=False)` on this very model (32 + 32 layers, the
    natural-EOT preset `mode_n` times), first window of two recordings (tests/golden/hf_generate_large_v3_golden.npz, written by
    tests/golden/make_golden.py generate_large_v3, which also checked that the oracle in exact f32 equals HF id for id on both).  The engine's sampled
    ids must equal HF's, with HF's segment times; a difference must be a near tie proven by forced replay on the oracle in the engine's arithmetic."""
    from speaksense_amd import binding
    orc = oracle_threads
    gold = np.load(os.path.join(ROOT, "tests", "golden", "hf_generate_large_v3_golden.npz"))
    dtype, omode, gap_tol = {"f16": (binding.DTYPE_F16, orc.MODE_GGML_F16, GAP_TOL_F16), "bf16": (binding.DTYPE_BF16, orc.MODE_BF16, GAP_TOL_BF16),
                             "fp8": (binding.DTYPE_FP8, orc.MODE_FP8, GAP_TOL_FP8_FULL_DEPTH)}[which]
    eng = binding.Engine(large_v3_natural_path, dtype=dtype, max_batch=2, compat=binding.COMPAT_OPENAI_TS_RULES)
    n_same, worst, lens = 0, 0.0, []
    for ci in range(int(gold["n_cases"])):
        if which != "f16" and ci == 0:
            continue            # bf16 / fp8 pick differently somewhere in a 113-id window, and every difference costs a full-depth replay on the CPU: the 45-id case
        k = f"c{ci}"
        pcm = synth.speech_like(int(gold[f"{k}_audio"]))
        ids, n0 = [int(t) for t in gold[f"{k}_ids"]], int(gold[f"{k}_n_window0"])
        seg = list(zip(gold[f"{k}_seg_t0"].tolist(), gold[f"{k}_seg_t1"].tolist()))
        got = eng.new_session().transcribe(pcm, binding.default_params(language="en", temperature_inc=0.0, duration_ms=30000))
        tr = [int(t) for t in got["trace"]]
        lens.append(n0)
        if tr[:n0] == ids[:n0]:
            n_same += 1
            assert [(s["t0"], s["t1"]) for s in got["segments"]][:len(seg)] == seg
            assert got["n_fail"] == 0
        else:
            om = shared_oracle_model(large_v3_natural_path)
            _, w = check_against_oracle(got, om, orc, omode, pcm, orc.default_params(language="en", temperature_inc=0.0, duration_ms=30000),
                                        f"HF generate at full depth, case {ci} ({which})", gap_tol, replay_only=True, compat=orc.COMPAT_OPENAI_TS_RULES)
            worst = max(worst, w)
            om.close()
    if which == "f16":
        assert n_same >= 1, "f16 differs from HF on both full-depth windows"
    report(f"HIP engine ({which}) vs HF generate at FULL depth (large-v3 natural preset, first windows of {lens} ids): {n_same}/{len(lens)} identical id for id with HF's segment times"
           + ("" if n_same == len(lens) else f"; the rest near-tie flips proven by forced replay, largest margin {worst:.4f}"))
    eng.close()
