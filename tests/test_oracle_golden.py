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
