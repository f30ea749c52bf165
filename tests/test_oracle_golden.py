"""CPU: the oracle against the committed golden vectors (tests/golden/hf_toy_golden.npz, made by make_golden.py from
HF transformers' Whisper on the seeded toy.en model) and against closed-form properties of whisper.cpp's mel."""
import os

import numpy as np
import pytest

from oracle import binding as orc
from speaksense_amd import ggml_io, synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_toy_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def gold_model(gold, model_dir):
    path = os.path.join(model_dir, "gold-toy.en.bin")
    ggml_io.write_model(path, "toy.en", seed=int(gold["seed_model"]))
    m = orc.OracleModel(path)
    yield m
    m.close()


def test_mel_matches_hf_feature_extractor(gold, gold_model):
    pcm = synth.speech_like(int(gold["seed_audio"]))
    mel = gold_model.log_mel(pcm)
    assert mel.shape == (80, 6000)  # 30 s of audio + 30 s of zero padding, hop 160
    cols = gold["mel_cols"]
    err = np.abs(mel[:, cols] - gold["hf_mel"]).max()
    assert err < 2e-4, err


def test_encoder_matches_hf(gold, gold_model):
    mel = np.zeros((80, 6000), np.float32)
    mel[:, :3000] = gold["hf_mel_in"].astype(np.float32)
    enc = gold_model.encode(mel, 0, orc.MODE_F32, gelu_erf=1)
    err = np.abs(enc[gold["enc_rows"]] - gold["enc"]).max()
    assert err < 1e-3 * float(gold["enc_absmax"]), err


def test_decoder_logits_match_hf(gold, gold_model):
    mel = np.zeros((80, 6000), np.float32)
    mel[:, :3000] = gold["hf_mel_in"].astype(np.float32)
    enc = gold_model.encode(mel, 0, orc.MODE_F32, gelu_erf=1)
    st = gold_model.new_state(orc.MODE_F32, gelu_erf=1)
    st.set_encoder(enc)
    toks = [int(t) for t in gold["tokens"]]
    tol = 2e-3 * float(gold["logit_std"])
    lg = st.decode(toks[:4], 0)                      # prompt pass (4 tokens at once) ...
    assert np.abs(lg[gold["topk"][3]] - gold["topv"][3]).max() < tol
    assert int(lg.argmax()) == int(gold["topk"][3][0])
    for i in range(4, len(toks)):                    # ... then KV-cached single-token steps
        lg = st.decode(toks[i:i + 1], i)
        assert np.abs(lg[gold["topk"][i]] - gold["topv"][i]).max() < tol, i
        assert int(lg.argmax()) == int(gold["topk"][i][0])
    st.close()


def test_mel_filterbank_is_slaney(gold):
    f80, f128 = ggml_io.mel_filters(80), ggml_io.mel_filters(128)
    assert f80.shape == (80, 201) and f128.shape == (128, 201)
    assert abs(float(f80.astype(np.float64).sum()) - float(gold["filt_sum"])) < 1e-6
    assert (f80 >= 0).all() and (f80.sum(axis=1) > 0).all()
    assert np.count_nonzero(f80[0]) <= 4   # triangular, narrow at low frequency


def test_mel_closed_form_cases(gold_model):
    # silence: every bin log10(1e-10) = -10 -> clamp leaves -10 -> (-10 + 4) / 4
    mel = gold_model.log_mel(synth.silence())
    assert mel.shape == (80, 6000) and np.all(mel == np.float32(-1.5))
    # frames that lie entirely in the 30 s zero padding sit at the clamp floor max - 8 (in log10 units, /4 after)
    pcm = synth.speech_like(2)
    mel = gold_model.log_mel(pcm)
    assert np.allclose(mel[:, 3100:], mel.max() - 2.0, atol=1e-6)
    assert mel.max() - mel.min() <= 2.0 + 1e-6
    # n_len follows whisper.cpp: (n + 480000 + 400 - 400) / 160
    for n in (16000, 80000, 480000, 481280):
        assert gold_model.log_mel(synth.speech_like(1, n)).shape[1] == (n + 480000) // 160


# ------------------------------------------------------------------------------------------------------------------------------------
# round 2 fixtures: more shapes, and the decoding rules
# ------------------------------------------------------------------------------------------------------------------------------------
SHAPES_GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_shapes_golden.npz")
RULES_GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_rules_golden.npz")


@pytest.mark.parametrize("name", ["toy", "tiny.en", "wide2"])
def test_oracle_matches_hf_on_more_shapes(name, model_dir):
    """128 mel bins + multilingual vocabulary/prompt (toy), the real tiny.en shape, and large-v3's width, head count and vocabulary (wide2):
    encoder rows and KV-cached decoder logits of the oracle (exact-f32 mode, erf GELU as HF) against HF transformers' Whisper."""
    g = np.load(SHAPES_GOLD)
    k = name.replace(".", "_")
    path = os.path.join(model_dir, f"gold-{name}.bin")
    ggml_io.write_model(path, name, seed=int(g[f"{k}_seed"]))
    om = orc.OracleModel(path)
    pcm = synth.speech_like(int(g["seed_audio"]))
    full = om.log_mel(pcm)
    if name == "toy":     # 128-bin filterbank against the HF feature extractor (away from the tail, where whisper.cpp zero-pads)
        assert full.shape[0] == 128
        a, b = full[:, g["toy_mel_cols"]], g["toy_hf_mel"]
        # the clamp floor is (global max - 8): HF's maximum sits in the reflect-padded last frames this path zero-pads, so the two floors differ
        # by ~1e-3; bins resting on a floor are compared to their own floor, everything above to 2e-4
        above = (a > a.min() + 2e-3) & (b > b.min() + 2e-3)
        assert above.mean() > 0.9 and np.abs(a - b)[above].max() < 2e-4
        assert abs(float(a.min()) - (float(full.max()) - 2.0)) < 1e-6
    mel = np.zeros_like(full)
    mel[:, :3000] = full[:, :3000].astype(np.float16).astype(np.float32)
    enc = om.encode(mel, 0, orc.MODE_F32, gelu_erf=1)
    err = np.abs(enc[g["enc_rows"]] - g[f"{k}_enc"]).max()
    assert err < 1e-3 * float(g[f"{k}_enc_absmax"]), err
    st = om.new_state(orc.MODE_F32, gelu_erf=1)
    st.set_encoder(enc)
    toks = [int(t) for t in g[f"{k}_tokens"]]
    n_prompt = int(g[f"{k}_n_prompt"])
    tol = 2e-3 * float(g[f"{k}_logit_std"])
    lg = st.decode(toks[:n_prompt], 0)
    assert np.abs(lg[g[f"{k}_topk"][n_prompt - 1]] - g[f"{k}_topv"][n_prompt - 1]).max() < tol
    for i in range(n_prompt, len(toks)):
        lg = st.decode(toks[i:i + 1], i)
        assert np.abs(lg[g[f"{k}_topk"][i]] - g[f"{k}_topv"][i]).max() < tol, i
        assert int(lg.argmax()) == int(g[f"{k}_topk"][i][0])
    st.close(); om.close()


@pytest.mark.parametrize("variant", ["wcpp_1_5", "openai_ts_rules"])
@pytest.mark.parametrize("tag,preset", [("en", "toy.en"), ("ml", "toy")])
def test_process_logits_matches_openai_rules(tag, preset, variant, model_dir):
    """whisper_process_logits (oracle restatement) against OpenAI's rules as HF transformers implements them (SuppressTokens*,
    WhisperTimeStampLogitsProcessor) on seeded logits x 12 token histories x 4 logit shapes, both vocabularies.  The two rule sets coincide
    except for three things, each excluded explicitly here and nowhere else:
      1. first step: OpenAI forces a timestamp (all text masked); whisper.cpp v1.5 only applies max_initial_ts -> the text region is not compared;
      2. monotonic timestamps: whisper.cpp masks ids < last timestamp (tid0 = seek_delta / 2); OpenAI masks <= last unless a pair is open
         -> the single id == last timestamp is not compared when OpenAI's "+1" branch is active;
      3. whisper.cpp's has_ts needs id > token_beg, so a history whose only timestamp is <|0.00|> masks nothing -- the same single id as in 2.
    variant "openai_ts_rules" = the oracle's COMPAT_OPENAI_TS_RULES switch (ledger rows 2-4), which restates OpenAI's form of
    exactly these three: under it NOTHING is excluded -- every mask bit and every top log-probability of every case must equal HF's."""
    from tests_golden_cases import rule_cases, rule_logits
    g = np.load(RULES_GOLD)
    path = os.path.join(model_dir, f"rules-{preset}.bin")
    ggml_io.write_model(path, preset, seed=1)
    om = orc.OracleModel(path)
    assert om.n_vocab == int(g[f"{tag}_n_vocab"])
    beg, eot = om.beg, om.eot
    openai = variant == "openai_ts_rules"
    st = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES if openai else 0)
    rng = np.random.default_rng(1234 + om.n_vocab)
    P = orc.default_params()
    n_cmp = 0
    for ci, (hist, trial) in enumerate(rule_cases(beg, eot)):
        raw = rule_logits(rng, om.n_vocab, beg, eot, trial)
        ts = [t for t in hist if t > beg]
        has_ts = bool(ts)
        seek_delta = 2 * (ts[-1] - beg) if ts else 3000
        _, lp, _ = st.process_logits(raw, hist, has_ts, seek_delta, P)
        mine = np.isinf(lp)
        gold = np.unpackbits(g[f"{tag}_mask"][ci])[:om.n_vocab].astype(bool)
        skip = np.zeros(om.n_vocab, bool)
        if not hist and not openai:
            skip[:beg] = True                                             # difference 1
        all_ts = [t for t in hist if t >= beg]
        if all_ts and not openai:
            pair_open = hist[-1] >= beg and not (len(hist) < 2 or hist[-2] >= beg)
            if not pair_open:
                skip[all_ts[-1]] = True                                   # differences 2 and 3
        cmp_ = ~skip
        assert np.array_equal(mine[cmp_], gold[cmp_]), (tag, hist, trial, np.nonzero(mine[cmp_] != gold[cmp_])[0][:8])
        # log-probabilities: relative to the most likely compared token (the skipped ids may carry mass that shifts the normaliser)
        top, topv = g[f"{tag}_top"][ci], g[f"{tag}_topv"][ci]
        keep = [j for j in range(len(top)) if cmp_[top[j]] and np.isfinite(topv[j])]
        if (hist or openai) and keep:
            j0 = keep[0]
            for j in keep:
                assert abs((lp[top[j]] - lp[top[j0]]) - (topv[j] - topv[j0])) < 2e-4 * max(1.0, abs(topv[j] - topv[j0])), (tag, hist, trial, int(top[j]))
            n_cmp += len(keep)
    assert n_cmp > 400
    st.close(); om.close()


# ------------------------------------------------------------------------------------------------------------------------------------
# round 5 fixtures: tanh GELU (the engine's and ggml's formula), and whole first windows of HF generate()
# ------------------------------------------------------------------------------------------------------------------------------------
TANH_GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_tanh_golden.npz")
GENERATE_GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_generate_golden.npz")


@pytest.mark.parametrize("name", ["toy", "tiny.en", "wide2"])
def test_oracle_default_gelu_matches_hf_gelu_new(name, model_dir):
    """The oracle in the mode every parity test uses (tanh GELU, ggml's formula) against HF Whisper with `activation_function="gelu_new"` on the
    seeded toy / tiny.en / wide2 models: log-mel columns (both filterbanks), encoder rows, KV-cached per-step logits.  The SAME vectors are
    applied to the HIP engine in tests/test_gpu_golden.py."""
    g = np.load(TANH_GOLD)
    k = name.replace(".", "_")
    path = os.path.join(model_dir, f"tanh-{name}.bin")
    ggml_io.write_model(path, name, seed=int(g[f"{k}_seed"]))
    om = orc.OracleModel(path)
    full = om.log_mel(synth.speech_like(int(g["seed_audio"])))
    if f"{k}_hf_mel" in g:
        a, b = full[:, g["mel_cols"]], g[f"{k}_hf_mel"]
        above = (a > a.min() + 2e-3) & (b > b.min() + 2e-3)       # bins on the clamp floor: the two floors differ by ~1e-3 (test above)
        assert above.mean() > 0.9 and np.abs(a - b)[above].max() < 2e-4
    mel = np.zeros_like(full)
    mel[:, :3000] = full[:, :3000].astype(np.float16).astype(np.float32)
    enc = om.encode(mel, 0, orc.MODE_F32)
    assert np.abs(enc[g["enc_rows"]] - g[f"{k}_enc"]).max() < 1e-3 * float(g[f"{k}_enc_absmax"])
    st = om.new_state(orc.MODE_F32)
    st.set_encoder(enc)
    toks = [int(t) for t in g[f"{k}_tokens"]]
    n_prompt = int(g[f"{k}_n_prompt"])
    tol = 2e-3 * float(g[f"{k}_logit_std"])
    lg = st.decode(toks[:n_prompt], 0)
    assert np.abs(lg[g[f"{k}_topk"][n_prompt - 1]] - g[f"{k}_topv"][n_prompt - 1]).max() < tol
    for i in range(n_prompt, len(toks)):
        lg = st.decode(toks[i:i + 1], i)
        assert np.abs(lg[g[f"{k}_topk"][i]] - g[f"{k}_topv"][i]).max() < tol, i
        assert int(lg.argmax()) == int(g[f"{k}_topk"][i][0])
    st.close()
    if f"{k}_lang_id" in g:     # whisper_lang_auto_detect == HF detect_language: argmax of the [sot] step's logits over the language tokens
        det = om.new_state(orc.MODE_F32).full(synth.speech_like(int(g["seed_audio"])), orc.default_params(language="en", detect_language=1))
        assert det["lang_id"] == int(g[f"{k}_lang_id"]) and float(g[f"{k}_lang_margin"]) > 1.0
    om.close()


def generate_cases():
    g = np.load(GENERATE_GOLD)
    return [dict(preset=str(g[f"c{i}_preset"]), seed=int(g[f"c{i}_seed"]), audio=int(g[f"c{i}_audio"]), ts_rate=float(g[f"c{i}_ts_rate"]),
                 language=str(g[f"c{i}_language"]), translate=int(g[f"c{i}_translate"]),
                 ids=[int(t) for t in g[f"c{i}_ids"]], n0=int(g[f"c{i}_n_window0"]), seg=list(zip(g[f"c{i}_seg_t0"].tolist(), g[f"c{i}_seg_t1"].tolist())))
            for i in range(int(g["n_cases"]))]


def generate_case_model(c, model_dir):
    path = os.path.join(model_dir, f"gen-{c['preset']}-{c['seed']}.bin")
    if not os.path.exists(path):
        ggml_io.write_model(path, c["preset"], seed=c["seed"], **dict(ggml_io.NATURAL, ts_rate=c["ts_rate"]))
    return path


@pytest.mark.parametrize("ci", range(10))
def test_oracle_full_matches_hf_generate_first_window(ci, model_dir):
    """SEQUENCE level (tests/golden/make_golden.py generate_fixture): the oracle's whole whisper_full loop -- prompt, every logits rule, greedy
    pick, the stopping rules (EOT; a timestamp within 1 s of the window's end), timestamp pairing into segments -- in exact-f32 mode with
    COMPAT_OPENAI_TS_RULES against HF transformers' `generate(return_timestamps=True, do_sample=False)` on the same seeded model and audio, with
    the parameters of /root/reference/src/asr/whisper.rs:131-173 at temperature 0: every id the loop samples in the first 30 s window must equal
    HF's, and the window's segments must carry HF's (start, end).  Cases 6 and 7 end by the window-end rule 13 / 33 ids before HF's EOT; cases 8 and 9
    put another language token and the translate task into the prompt ([sot, <|de|>, <|translate|>], [sot, <|fr|>, <|transcribe|>])."""
    c = generate_cases()[ci]
    om = orc.OracleModel(generate_case_model(c, model_dir))
    pcm = synth.speech_like(c["audio"])
    res = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language=c["language"], translate=c["translate"], temperature_inc=0.0))
    tr = [int(t) for t in res["trace"]]
    n0 = c["n0"]
    assert n0 >= 25 and sum(t >= om.beg for t in c["ids"][:n0]) >= 3, "fixture drifted: the window holds no timestamp pairs"
    assert tr[:n0] == c["ids"][:n0], f"first difference at {next(i for i in range(n0) if i >= len(tr) or tr[i] != c['ids'][i])}"
    assert len(tr) == n0 or res["n_encode"] > 1          # the loop ended the window exactly there (what follows belongs to the next window)
    assert [(s["t0"], s["t1"]) for s in res["segments"]][:len(c["seg"])] == c["seg"]
    assert res["n_fail"] == 0
    if ci == 8:     # the prompt matters on this fixture: the same audio with [sot, <|en|>, <|transcribe|>] gives another stream
        other = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language="en", temperature_inc=0.0))
        assert [int(t) for t in other["trace"]][:n0] != c["ids"][:n0]
    om.close()


GENERATE_LONG_GOLD = os.path.join(os.path.dirname(__file__), "golden", "hf_generate_long_golden.npz")


def generate_long_cases():
    g = np.load(GENERATE_LONG_GOLD)
    out = []
    for i in range(int(g["n_cases"])):
        wins = [dict(ids=[int(t) for t in g[f"c{i}_w{w}_ids"]], n=int(g[f"c{i}_w{w}_n"]), seek=int(g[f"c{i}_w{w}_seek"]),
                     seg=[(int(a), int(b)) for a, b in g[f"c{i}_w{w}_seg"]]) for w in range(int(g[f"c{i}_n_windows"]))]
        out.append(dict(preset=str(g[f"c{i}_preset"]), seed=int(g[f"c{i}_seed"]), audio=int(g[f"c{i}_audio"]), ts_rate=12.0, seconds=int(g["seconds"]), windows=wins))
    return out


def long_case_params(mod, c):
    """Parameters for a long fixture case: `duration_ms` ends the call 2 s after the last stored window could end -- far enough that no stored window feels
    it (whisper.cpp's window-end rule looks 1 s ahead), near enough that the 95 s of audio cost 3 - 4 windows instead of 7."""
    last = c["windows"][-1]["seek"]
    return mod.default_params(language="en", temperature_inc=0.0, duration_ms=10 * (last + 3000 + 200))


def check_windows_against_hf(res, c, beg):
    """The leading windows of a whisper_full result (oracle or engine) against the HF fixture: every id sampled in window w equals HF's, the windows' segments
    carry HF's absolute times, and the seek advance implied by them is HF's.  Returns None, or a description of the first difference."""
    tr, pos = [int(t) for t in res["trace"]], 0
    for wi, w in enumerate(c["windows"]):
        got = tr[pos:pos + w["n"]]
        if got != w["ids"][:w["n"]]:
            k = next((i for i in range(min(len(got), w["n"])) if got[i] != w["ids"][i]), min(len(got), w["n"]))
            return f"window {wi} (seek {w['seek']}): sampled id {k} differs from HF's"
        pos += w["n"]
    segs = [sg for w in c["windows"] for sg in w["seg"]]
    if [(s["t0"], s["t1"]) for s in res["segments"]][:len(segs)] != segs:
        return "segment times differ from HF's"
    if res["n_encode"] < len(c["windows"]):
        return "fewer windows than HF"
    return None


@pytest.mark.parametrize("ci", range(3))
def test_oracle_full_matches_hf_generate_across_windows(ci, model_dir):
    """SEQUENCE level across window boundaries (make_golden.py generate_long_fixture): 95 s of audio, HF `generate(condition_on_prev_tokens=True)` fed
    whisper.cpp's own padded log-mel.  Over the leading regular windows (2 - 3 per case) the oracle's whisper_full loop under COMPAT_OPENAI_TS_RULES |
    COMPAT_OPENAI_HISTORY samples HF's ids, advances `seek` to HF's frame, conditions the next window on `[prev] + history + [sot ..]` as HF does
    (otherwise the ids of window 1 would differ) and reports HF's absolute segment times.  /root/reference/src/asr/whisper.rs:75 with the parameters of
    :131-173 at temperature 0; the two flags cover the documented differences between whisper.cpp and OpenAI."""
    c = generate_long_cases()[ci]
    om = orc.OracleModel(generate_case_model(c, model_dir))
    pcm = synth.speech_like(c["audio"], 16000 * c["seconds"])
    P = long_case_params(orc, c)
    res = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES | orc.COMPAT_OPENAI_HISTORY).full(pcm, P)
    assert len(c["windows"]) >= 2 and all(w["n"] >= 15 for w in c["windows"]) and c["windows"][1]["seek"] > 0
    assert check_windows_against_hf(res, c, om.beg) is None, check_windows_against_hf(res, c, om.beg)
    om.close()
