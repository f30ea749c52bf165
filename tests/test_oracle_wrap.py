"""CPU: two `whisper_full_params` features a whisper-rs caller can reach and the reference pins to their defaults -- `suppress_non_speech_tokens`
(/root/reference/src/asr/whisper.rs:156, false) and `max_len` + `split_on_word` (whisper.rs:167,161: 0 = no wrapping) -- as restated in the oracle.

* `whisper_wrap_segment`: the oracle's C++ against an independent Python restatement of the same whisper.cpp function applied to the UNWRAPPED run's
  segments and token times, plus the properties the function guarantees whatever the implementation (pieces partition the tokens and the text, piece
  times chain through the cut tokens' t0, a piece is longer than max_len only when it is one token or no word boundary was available).
* the non-speech list: on a vocabulary with some of the symbols planted, exactly those ids (and " -" / " '", but not "-") are masked additionally.

whisper.cpp itself is absent (whisper-rs-sys 0.9.0, /root/reference/Cargo.lock:3888-3907): both are restated from memory, parity unpinned."""
import os

import numpy as np
import pytest

from speaksense_amd import ggml_io, synth

# id -> text planted into the synthetic vocabulary (speaksense_amd/ggml_io.py synth_vocab has none of whisper.cpp's non-speech symbols)
PLANTED = {1000: b"(", 1001: b" (", 1002: "♪".encode(), 1003: b" -", 1004: b" '", 1005: b"-", 1006: " ♪♪".encode(), 1007: b"[[", 1008: b" \\",
           1009: b"'", 1010: "「".encode(), 1011: b" )))", 1012: b"((((", 1013: b" #"}
NON_SPEECH = {1000, 1001, 1002, 1003, 1004, 1006, 1007, 1008, 1010, 1011, 1013}     # "-", "'" and "((((" are not on the list


def planted_model(model_dir):
    """tiny.en, natural-EOT style, with non-speech symbols planted into the vocabulary: the fixed PLANTED ids, and -- so that the flag provably changes a
    transcription -- "*", " [" and "♫" at three text ids the PLAIN model emits for synth.speech_like(11) (the weights do not depend on the vocabulary).
    Returns (path, the three ids)."""
    from oracle import binding as orc
    path = os.path.join(model_dir, "tiny.en-natural-planted.bin")
    ids_path = path + ".ids.npy"
    if not (os.path.exists(path) and os.path.exists(ids_path)):
        ggml_io.write_model(path + ".plain", "tiny.en", seed=3, **ggml_io.NATURAL)
        om = orc.OracleModel(path + ".plain")
        res = om.new_state(orc.MODE_F32).full(synth.speech_like(11), orc.default_params(language="en"))
        emitted = [int(t) for t in dict.fromkeys(int(t) for t in res["tokens"]) if t < om.eot and t not in PLANTED]
        om.close()
        hit = [emitted[len(emitted) // 4], emitted[len(emitted) // 2], emitted[3 * len(emitted) // 4]]
        ov = dict(PLANTED)
        ov.update({hit[0]: b"*", hit[1]: b" [", hit[2]: "♫".encode()})
        ggml_io.write_model(path, "tiny.en", seed=3, vocab_overrides=ov, **ggml_io.NATURAL)
        os.remove(path + ".plain")
        np.save(ids_path, np.array(hit, np.int32))
    return path, [int(i) for i in np.load(ids_path)]


@pytest.fixture(scope="module")
def wrap_model(model_dir):
    return planted_model(model_dir)[0]


def py_wrap(seg, strs, eot, max_len, split_on_word):
    """whisper_wrap_segment in its own words: `seg` = dict(t0, t1, ids, tok_t0); returns [(t0, t1, text, n_tokens)].  The function re-walks the
    remainder from its first token after every cut, and never cuts in front of the first token of a piece."""
    out = []
    ids, tt0 = list(seg["ids"]), list(seg["tok_t0"])
    t0 = seg["t0"]
    while True:
        acc, text, cut = 0, b"", None
        for i, tid in enumerate(ids):
            if tid >= eot:
                continue
            txt = strs[tid]
            cur = len(txt.split(b"\0")[0])
            if acc + cur > max_len and i > 0 and (not split_on_word or txt[:1] == b" "):
                cut = i
                break
            acc += cur
            text += txt
        if cut is None:
            out.append((t0, seg["t1"], text, len(ids)))
            return out
        out.append((t0, tt0[cut], text, cut))
        t0 = tt0[cut]
        ids, tt0 = ids[cut:], tt0[cut:]


_PLAIN = {}      # the unwrapped run of the one (model, audio) the wrap tests share


def unwrapped_segments(res):
    return [dict(t0=s["t0"], t1=s["t1"], ids=[int(x) for x in s["token_times"]["ids"]], tok_t0=[int(x) for x in s["token_times"]["t0"]]) for s in res["segments"]]


@pytest.mark.parametrize("max_len,split_on_word", [(20, 1), (20, 0), (7, 1), (1, 0), (60, 1)])
def test_oracle_wrap_segment(wrap_model, max_len, split_on_word):
    from oracle import binding as orc
    om = orc.OracleModel(wrap_model)
    _, _, strs, _ = ggml_io.read_model(wrap_model)
    pcm = synth.speech_like(11)
    if "plain" not in _PLAIN:
        _PLAIN["plain"] = om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en"))
    plain = _PLAIN["plain"]
    got = om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", max_len=max_len, split_on_word=split_on_word))
    assert list(got["tokens"]) == list(plain["tokens"]) and got["n_encode"] == plain["n_encode"]        # wrapping never feeds back into decoding
    want = [p for seg in unwrapped_segments(plain) for p in py_wrap(seg, strs, om.eot, max_len, split_on_word)]
    have = [(s["t0"], s["t1"], s["text"], len(s["token_times"]["ids"])) for s in got["segments"]]
    assert have == want
    assert len(have) > len(plain["segments"]) or max_len >= 60
    # properties, per original segment
    k = 0
    for seg in plain["segments"]:
        pieces = []
        n_tok = 0
        while n_tok < len(seg["token_times"]["ids"]):
            pieces.append(got["segments"][k]); n_tok += len(got["segments"][k]["token_times"]["ids"]); k += 1
        assert n_tok == len(seg["token_times"]["ids"])
        assert np.concatenate([p["token_times"]["ids"] for p in pieces]).tolist() == seg["token_times"]["ids"].tolist()
        assert np.concatenate([p["token_times"]["t0"] for p in pieces]).tolist() == seg["token_times"]["t0"].tolist()
        assert pieces[0]["t0"] == seg["t0"] and pieces[-1]["t1"] == seg["t1"]
        assert pieces[-1]["speaker_turn_next"] == seg["speaker_turn_next"] and not any(p["speaker_turn_next"] for p in pieces[:-1])
        for a, b in zip(pieces, pieces[1:]):
            assert a["t1"] == b["t0"] == int(b["token_times"]["t0"][0])
        assert b"".join(p["text"] for p in pieces) == b"".join(strs[i] for i in seg["token_times"]["ids"] if i < om.eot)
        for j, p in enumerate(pieces):
            texts = [strs[i] for i in p["token_times"]["ids"] if i < om.eot]
            if split_on_word and j > 0 and texts:
                assert texts[0][:1] == b" "
            if len(p["text"]) > max_len:         # only when no cut was allowed earlier: a single text token, or (split_on_word) no later word start
                later = texts[1:]
                assert len(texts) == 1 or (split_on_word and all(len(b"".join(texts[:q + 1])) <= max_len or t[:1] != b" " for q, t in enumerate(later)))
    assert k == len(got["segments"])
    om.close()


def test_oracle_wrap_needs_token_timestamps(wrap_model):
    """whisper.cpp wraps inside its token_timestamps branch: max_len without token_timestamps leaves the segments alone."""
    from oracle import binding as orc
    om = orc.OracleModel(wrap_model)
    pcm = synth.speech_like(11)
    plain = om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", token_timestamps=0))
    got = om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", token_timestamps=0, max_len=10))
    assert [(s["t0"], s["t1"], s["text"]) for s in got["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in plain["segments"]]
    om.close()


def test_oracle_suppress_non_speech_tokens(model_dir):
    from oracle import binding as orc
    path, hit = planted_model(model_dir)
    NON_SPEECH = globals()["NON_SPEECH"] | set(hit)
    om = orc.OracleModel(path)
    st = om.new_state(orc.MODE_F32)
    raw = np.random.default_rng(5).standard_normal(om.n_vocab).astype(np.float32)
    for hist, has_ts in (([], False), ([om.beg + 3, 700, 701], True), ([om.beg, 900, om.beg + 10, om.beg + 10], True)):
        _, off, _ = st.process_logits(raw, hist, has_ts, 0, orc.default_params(language="en"))
        _, on, _ = st.process_logits(raw, hist, has_ts, 0, orc.default_params(language="en", suppress_non_speech_tokens=1))
        extra = set(np.nonzero(np.isneginf(on) & ~np.isneginf(off))[0].tolist())
        already = {i for i in NON_SPEECH if np.isneginf(off[i])}        # a timestamp rule may have masked all text anyway
        assert extra == NON_SPEECH - already, (sorted(extra), hist)
        assert not (np.isneginf(off) & ~np.isneginf(on)).any()
        if not already:
            assert not np.isneginf(on[1005]) and not np.isneginf(on[1009]) and not np.isneginf(on[1012])
    om.close()


def test_oracle_suppress_non_speech_tokens_changes_a_transcription(model_dir):
    """Three of the ids the plain run emits carry non-speech symbols in this vocabulary: under the flag the stream keeps its prefix up to the first of
    them, takes another id there, and never emits any id of the list."""
    from oracle import binding as orc
    path, hit = planted_model(model_dir)
    om = orc.OracleModel(path)
    pcm = synth.speech_like(11)
    plain = [int(t) for t in om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en"))["trace"]]
    flagged = [int(t) for t in om.new_state(orc.MODE_F32).full(pcm, orc.default_params(language="en", suppress_non_speech_tokens=1))["trace"]]
    banned = NON_SPEECH | set(hit)
    assert all(h in plain for h in hit)
    first = min(plain.index(h) for h in hit)
    assert flagged[:first] == plain[:first] and flagged[first] != plain[first]
    assert not (set(flagged) & banned)
    om.close()


def ascii_vocab_model(model_dir):
    """toy.en with ids 0 .. 93 = the printable ASCII characters '!' .. '~' in order: the head of every GPT-2 byte-level vocabulary (the real ones are not here)."""
    path = os.path.join(model_dir, "toy.en-ascii-head.bin")
    if not os.path.exists(path):
        ggml_io.write_model(path, "toy.en", seed=9, vocab_overrides={i: bytes([33 + i]) for i in range(94)})
    return path


def test_non_speech_single_characters_match_hf_list(model_dir):
    """The single-ASCII-character part of the symbol table against an independent source: transformers' NON_SPEECH_TOKENS (the default suppress_tokens
    of the real checkpoints, tests/golden/hf_non_speech_ids.npz) lists, below id 94, exactly the characters this restatement masks -- and neither '-'
    nor the apostrophe."""
    from oracle import binding as orc
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hf_non_speech_ids.npz"))
    want = sorted(int(i) for i in g["en"] if i < 94)
    assert want == sorted(int(i) for i in g["multi"] if i < 94) and len(want) == 23
    om = orc.OracleModel(ascii_vocab_model(model_dir))
    st = om.new_state(orc.MODE_F32)
    raw = np.zeros(om.n_vocab, np.float32)
    raw[om.beg:] = -30.0          # keep the "timestamp mass beats every text token" rule out of the way: it would mask all text by itself
    _, off, _ = st.process_logits(raw, [om.beg + 3, 700], True, 0, orc.default_params(language="en"))
    _, on, _ = st.process_logits(raw, [om.beg + 3, 700], True, 0, orc.default_params(language="en", suppress_non_speech_tokens=1))
    got = sorted(int(i) for i in np.nonzero(np.isneginf(on[:94]) & ~np.isneginf(off[:94]))[0])
    assert got == want, "".join(chr(33 + i) for i in sorted(set(got) ^ set(want)))
    assert ord("-") - 33 not in got and ord("'") - 33 not in got
    om.close()
