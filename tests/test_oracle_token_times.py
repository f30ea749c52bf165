"""Token-level timestamps (whisper_full_params.token_timestamps; the reference sets it: /root/reference/src/asr/whisper.rs:160, thresholds :170-171).

whisper.cpp's whisper_exp_compute_token_level_timestamps is restated three times in this repository: oracle/whisper_oracle.cpp (C++, the checker),
speaksense_amd/csrc/engine.cpp (the product's host pass, compared with the oracle under -m gpu) and the plain-Python version below, which exists
only to catch a slip in the oracle on small random cases.  None of them is pinned to whisper.cpp itself (ledger row 8)."""
import numpy as np
import pytest

SR = 16000


def py_voice_length(text: bytes) -> np.float32:
    r = np.float32(0)
    for c in text:
        ch = chr(c) if c < 128 else "x"
        if ch == " ":
            r = np.float32(r + np.float32(0.01))
        elif ch == ",":
            r = np.float32(r + np.float32(2))
        elif ch in ".!?":
            r = np.float32(r + np.float32(3))
        elif "0" <= ch <= "9":
            r = np.float32(r + np.float32(3))
        else:
            r = np.float32(r + np.float32(1))
    return r


def py_energy(x: np.ndarray, hw: int = 32) -> np.ndarray:
    n = len(x)
    e = np.zeros(n, np.float32)
    ax = np.abs(x.astype(np.float32))
    for i in range(n):
        s = np.float32(0)
        for j in range(max(0, i - hw), min(n, i + hw + 1)):
            s = np.float32(s + ax[j])
        e[i] = np.float32(s / np.float32(2 * hw + 1))
    return e


def py_token_times(state, beg, eot, seg_t0, seg_t1, toks, strs, en, thold_pt, thold_ptsum):
    """toks: list of dict(id, tid, pt, ptsum); returns (t0[], t1[], vlen[]); state = dict(t_beg, t_last, tid_last), updated in place."""
    n, ns = len(toks), len(en)
    t0 = [-1] * n
    t1 = [-1] * n
    vl = [np.float32(0)] * n
    if ns == 0 or n == 0:
        return t0, t1, vl
    if n == 1:
        return [seg_t0], [seg_t1], vl
    to_sample = lambda t: max(0, min(ns - 1, int((t * SR) // 100)))
    to_time = lambda i: (100 * i) // SR
    for j in range(n):
        tk = toks[j]
        if j == 0:
            if tk["id"] == beg:
                t0[0] = seg_t0; t1[0] = seg_t0; t0[1] = seg_t0
                state["t_beg"] = seg_t0; state["t_last"] = seg_t0; state["tid_last"] = beg
            else:
                t0[0] = state["t_last"]
        tt = state["t_beg"] + 2 * (tk["tid"] - beg)
        vl[j] = py_voice_length(strs[tk["id"]])
        if tk["pt"] > thold_pt and tk["ptsum"] > thold_ptsum and tk["tid"] > state["tid_last"] and tt <= seg_t1:
            if j > 0:
                t1[j - 1] = tt
            t0[j] = tt
            state["tid_last"] = tk["tid"]
    t1[n - 2] = seg_t1; t0[n - 1] = seg_t1; t1[n - 1] = seg_t1
    state["t_last"] = seg_t1
    p0 = p1 = 0
    while True:
        while p1 < n and t1[p1] < 0:
            p1 += 1
        if p1 >= n:
            p1 -= 1
        if p1 > p0:
            psum = 0.0
            for j in range(p0, p1 + 1):
                psum += float(vl[j])
            dt = float(t1[p1] - t0[p0])
            for j in range(p0 + 1, p1 + 1):
                ct = t0[j - 1] + dt * float(vl[j - 1]) / psum
                t1[j - 1] = int(ct); t0[j] = int(ct)
        p1 += 1; p0 = p1
        if p1 >= n:
            break
    for j in range(n - 1):
        if t1[j] < 0:
            t0[j + 1] = t1[j]
        if j > 0 and t1[j - 1] > t0[j]:
            t0[j] = t1[j - 1]; t1[j] = max(t0[j], t1[j])
    hw = SR // 8
    for j in range(n):
        if toks[j]["id"] >= eot:
            continue
        s0, s1 = to_sample(t0[j]), to_sample(t1[j])
        ss0, ss1 = max(s0 - hw, 0), min(s1 + hw, ns)
        acc = np.float32(0)
        for k in range(ss0, ss1):
            acc = np.float32(acc + en[k])
        thold = np.float32(0.5 * float(acc) / (ss1 - ss0))
        k = s0
        if en[k] > thold and j > 0:
            while k > 0 and en[k] > thold:
                k -= 1
            t0[j] = to_time(k)
            if t0[j] < t1[j - 1]:
                t0[j] = t1[j - 1]
            else:
                s0 = k
        else:
            while en[k] < thold and k < s1:
                k += 1
            s0 = k; t0[j] = to_time(k)
        k = s1
        if en[k] > thold:
            while k < ns - 1 and en[k] > thold:
                k += 1
            t1[j] = to_time(k)
            if j + 1 < n and t1[j] > t0[j + 1]:
                t1[j] = t0[j + 1]
        else:
            while en[k] < thold and k > s0:
                k -= 1
            t1[j] = to_time(k)
    return t0, t1, vl


def test_voice_length_and_signal_energy_building_blocks():
    from oracle import binding as orc
    for text in (b"", b" ", b" Hello, 12.", b"?!.", b"[_TT_150]", " café".encode()):
        assert orc.voice_length(text) == pytest.approx(float(py_voice_length(text)), abs=0) or abs(orc.voice_length(text) - float(py_voice_length(text))) < 1e-6
    rng = np.random.default_rng(5)
    for n in (1, 5, 64, 65, 66, 700):
        x = (rng.standard_normal(n) * 0.3).astype(np.float32)
        got, want = orc.signal_energy(x), py_energy(x)
        assert np.array_equal(got, want), (n, np.abs(got - want).max())
    assert len(orc.signal_energy(np.zeros(0, np.float32))) == 0


def _strs(om):
    return [om.token_str(i) for i in range(om.n_vocab)]


def test_hand_computed_cases(toy_en_path):
    """Cases small enough to follow by hand: one token; no evidence and silence (pure voice-length split); one anchored token."""
    from oracle import binding as orc
    om = orc.OracleModel(toy_en_path)
    beg, eot = om.beg, om.eot
    strs = _strs(om)
    silence = np.zeros(3 * SR, np.float32)
    text_ids = [i for i in range(300, 400) if float(py_voice_length(strs[i])) > 0][:3]
    a, b, c = text_ids
    va, vb, vc = (float(py_voice_length(strs[i])) for i in text_ids)
    tok = lambda i, tid=beg, pt=0.0, ps=0.0: dict(ids=i, tid=tid, pt=pt, ptsum=ps)
    pack = lambda toks: dict(ids=[t["ids"] for t in toks], tid=[t["tid"] for t in toks], pt=[t["pt"] for t in toks], ptsum=[t["ptsum"] for t in toks])
    # (1) a single token takes the segment's times
    r = om.token_times_chunk(silence, [(40, 90)], [pack([tok(a)])])
    assert (int(r[0]["t0"][0]), int(r[0]["t1"][0])) == (40, 90)
    # (2) <|0.00|> A B C <|2.00|>, nothing passes the thresholds, silent audio: [0, 200] is split over A B C by voice length; the energy walk is a
    #     no-op on an all-zero signal (nothing is above or below the threshold 0)
    r = om.token_times_chunk(silence, [(0, 200)], [pack([tok(beg), tok(a), tok(b), tok(c), tok(beg + 100, beg + 100)])])[0]
    tot = va + vb + vc
    cut1 = int(0 + 200.0 * va / tot)
    cut2 = int(cut1 + 200.0 * vb / tot)
    assert [int(x) for x in r["t0"]] == [0, 0, cut1, cut2, 200]
    assert [int(x) for x in r["t1"]] == [0, cut1, cut2, 200, 200]
    assert [float(x) for x in r["vlen"]][1:4] == pytest.approx([va, vb, vc])
    # (3) the same with B carrying timestamp evidence for 1.00 s: A ends and B starts at 100; B and C share [100, 200]
    r = om.token_times_chunk(silence, [(0, 200)], [pack([tok(beg), tok(a), tok(b, beg + 50, 0.9, 0.9), tok(c), tok(beg + 100, beg + 100)])])[0]
    cut = int(100 + 100.0 * vb / (vb + vc))
    assert [int(x) for x in r["t0"]] == [0, 0, 100, cut, 200]
    assert [int(x) for x in r["t1"]] == [0, 100, cut, 200, 200]
    # (4) evidence that does not advance (tid <= the last accepted one) or lies beyond the segment end is ignored
    r2 = om.token_times_chunk(silence, [(0, 200)], [pack([tok(beg), tok(a), tok(b, beg, 0.9, 0.9), tok(c, beg + 150, 0.9, 0.9), tok(beg + 100, beg + 100)])])[0]
    assert [int(x) for x in r2["t0"]] == [0, 0, cut1, cut2, 200]
    om.close()


def test_oracle_matches_the_python_restatement_on_random_segments(toy_en_path):
    """Random token data (evidence on / off, timestamps inside the text, segments that do not start with <|0.00|>, state carried from segment to
    segment) over a signal with bursts and silences: the C++ restatement and the Python one above agree exactly."""
    from oracle import binding as orc
    om = orc.OracleModel(toy_en_path)
    beg, eot = om.beg, om.eot
    strs = _strs(om)
    rng = np.random.default_rng(11)
    n_checked = 0
    for case in range(12):
        n_s = int(rng.integers(2, 9)) * SR
        pcm = np.zeros(n_s, np.float32)
        for _ in range(int(rng.integers(1, 6))):                       # voiced bursts
            a = int(rng.integers(0, n_s - 1000)); b = min(n_s, a + int(rng.integers(500, SR)))
            pcm[a:b] = (rng.standard_normal(b - a) * rng.uniform(0.05, 0.5)).astype(np.float32)
        en = py_energy(pcm) if n_s <= 3 * SR else orc.signal_energy(pcm)   # the building block has its own test; keep the Python loop for short cases
        segs, lists = [], []
        t = 0
        for _ in range(int(rng.integers(1, 5))):
            t1 = t + int(rng.integers(20, 300))
            k = int(rng.integers(1, 9))
            toks = []
            if rng.random() < 0.6:
                toks.append(dict(ids=beg + (t // 2 if rng.random() < 0.5 else 0), tid=beg + t // 2, pt=0.5, ptsum=0.9))
            for _ in range(k):
                toks.append(dict(ids=int(rng.integers(0, eot)), tid=beg + int(rng.integers(0, 160)), pt=float(rng.choice([0.0, 0.005, 0.02, 0.7])),
                                 ptsum=float(rng.choice([0.0, 0.005, 0.3]))))
            if rng.random() < 0.7:
                toks.append(dict(ids=beg + t1 // 2, tid=beg + t1 // 2, pt=0.9, ptsum=0.9))
            elif rng.random() < 0.5:
                toks.append(dict(ids=eot, tid=beg, pt=0.0, ptsum=0.0))
            segs.append((t, t1)); lists.append(toks)
            t = t1
        packed = [dict(ids=[q["ids"] for q in l], tid=[q["tid"] for q in l], pt=[q["pt"] for q in l], ptsum=[q["ptsum"] for q in l]) for l in lists]
        got = om.token_times_chunk(pcm, segs, packed)
        state = dict(t_beg=0, t_last=0, tid_last=0)
        for (s0, s1), l, g in zip(segs, lists, got):
            toks = [dict(id=q["ids"], tid=q["tid"], pt=np.float32(q["pt"]), ptsum=np.float32(q["ptsum"])) for q in l]
            w0, w1, wv = py_token_times(state, beg, eot, s0, s1, toks, strs, en, np.float32(0.01), np.float32(0.01))
            assert [int(x) for x in g["t0"]] == w0 and [int(x) for x in g["t1"]] == w1, (case, s0, s1, l)
            assert np.allclose(g["vlen"], np.array(wv, np.float32), atol=1e-6)
            n_checked += len(l)
    assert n_checked > 100
    om.close()


def test_full_path_flag_changes_nothing_but_the_token_times(tmp_path):
    """token_timestamps on / off through whisper_full on the oracle: identical tokens, text and segment times (max_len = 0, whisper.rs:167); with the
    flag off every token keeps t0 = t1 = -1, with it on every token of every segment has times inside [0, audio length] and t0 <= t1."""
    from oracle import binding as orc
    from speaksense_amd import ggml_io, synth
    path = str(tmp_path / "toy256-natural.bin")
    ggml_io.write_model(path, "toy256", seed=0, **ggml_io.NATURAL)
    om = orc.OracleModel(path)
    pcm = synth.speech_like(103)
    on = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="en"))
    off = om.new_state(orc.MODE_GGML_F16).full(pcm, orc.default_params(language="en", token_timestamps=0))
    assert list(on["tokens"]) == list(off["tokens"])
    assert [(s["text"], s["t0"], s["t1"]) for s in on["segments"]] == [(s["text"], s["t0"], s["t1"]) for s in off["segments"]]
    n_tok = 0
    for s_on, s_off in zip(on["segments"], off["segments"]):
        assert (s_off["token_times"]["t0"] == -1).all() and (s_off["token_times"]["t1"] == -1).all() and (s_off["token_times"]["vlen"] == 0).all()
        tt = s_on["token_times"]
        assert (tt["t0"] >= 0).all() and (tt["t1"] >= tt["t0"]).all(), (tt["t0"], tt["t1"])
        n_tok += len(tt["ids"])
    assert n_tok >= 20
    om.close()
