"""Generates tests/golden/hf_toy_golden.npz.  Runs ONLY in the build container (needs transformers + torch-CPU).

What it pins: the CPU oracle's network arithmetic (conv stem, pre-LN blocks, bias-less K, scaling, tied logits,
positional embeddings, KV-cached decoding) and its log-mel front end, against HF transformers' Whisper -- a secondary
oracle, clearly NOT the reference (the reference's arithmetic is whisper.cpp, absent offline).
The model is the seeded synthetic `toy.en` file from speaksense_amd.ggml_io (regenerated bit-identically by the
tests from the seed), loaded into WhisperForConditionalGeneration; HF uses erf-GELU, so the oracle is compared
with gelu_erf=1 there.  whisper.cpp zero-pads where OpenAI reflect-pads, so mel is compared away from the last frames.

usage: python tests/golden/make_golden.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402

from speaksense_amd import ggml_io, synth  # noqa: E402

SEED_MODEL, SEED_AUDIO = 7, 1
ENC_ROWS = [0, 1, 2, 700, 1498, 1499]
TOKENS = [50257, 1234, 777, 50000, 42, 31337, 9, 50364 + 10]


def ggml_to_hf_state(hp, t):
    sd = {}
    sd["model.encoder.conv1.weight"] = t["encoder.conv1.weight"]
    sd["model.encoder.conv1.bias"] = t["encoder.conv1.bias"].reshape(-1)
    sd["model.encoder.conv2.weight"] = t["encoder.conv2.weight"]
    sd["model.encoder.conv2.bias"] = t["encoder.conv2.bias"].reshape(-1)
    sd["model.encoder.embed_positions.weight"] = t["encoder.positional_embedding"]
    sd["model.encoder.layer_norm.weight"] = t["encoder.ln_post.weight"]
    sd["model.encoder.layer_norm.bias"] = t["encoder.ln_post.bias"]
    sd["model.decoder.embed_positions.weight"] = t["decoder.positional_embedding"]
    sd["model.decoder.embed_tokens.weight"] = t["decoder.token_embedding.weight"]
    sd["proj_out.weight"] = t["decoder.token_embedding.weight"]
    sd["model.decoder.layer_norm.weight"] = t["decoder.ln.weight"]
    sd["model.decoder.layer_norm.bias"] = t["decoder.ln.bias"]

    def attn(src, dst):
        for a, b in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
            sd[f"{dst}.{b}.weight"] = t[f"{src}.{a}.weight"]
            if a != "key":
                sd[f"{dst}.{b}.bias"] = t[f"{src}.{a}.bias"]

    for side, n in (("encoder", hp.n_audio_layer), ("decoder", hp.n_text_layer)):
        for i in range(n):
            s, d = f"{side}.blocks.{i}", f"model.{side}.layers.{i}"
            attn(f"{s}.attn", f"{d}.self_attn")
            sd[f"{d}.self_attn_layer_norm.weight"] = t[f"{s}.attn_ln.weight"]
            sd[f"{d}.self_attn_layer_norm.bias"] = t[f"{s}.attn_ln.bias"]
            if side == "decoder":
                attn(f"{s}.cross_attn", f"{d}.encoder_attn")
                sd[f"{d}.encoder_attn_layer_norm.weight"] = t[f"{s}.cross_attn_ln.weight"]
                sd[f"{d}.encoder_attn_layer_norm.bias"] = t[f"{s}.cross_attn_ln.bias"]
            sd[f"{d}.final_layer_norm.weight"] = t[f"{s}.mlp_ln.weight"]
            sd[f"{d}.final_layer_norm.bias"] = t[f"{s}.mlp_ln.bias"]
            sd[f"{d}.fc1.weight"] = t[f"{s}.mlp.0.weight"]
            sd[f"{d}.fc1.bias"] = t[f"{s}.mlp.0.bias"]
            sd[f"{d}.fc2.weight"] = t[f"{s}.mlp.2.weight"]
            sd[f"{d}.fc2.bias"] = t[f"{s}.mlp.2.bias"]
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def main():
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "toy.en.bin")
    hp = ggml_io.write_model(path, "toy.en", seed=SEED_MODEL)
    hp, filt, vocab, tensors = ggml_io.read_model(path)   # f16-rounded weights, exactly what both loaders see
    cfg = WhisperConfig(vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, d_model=hp.n_audio_state, encoder_layers=hp.n_audio_layer,
                        encoder_attention_heads=hp.n_audio_head, decoder_layers=hp.n_text_layer, decoder_attention_heads=hp.n_text_head,
                        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state, max_source_positions=hp.n_audio_ctx,
                        max_target_positions=hp.n_text_ctx, activation_function="gelu", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0)
    model = WhisperForConditionalGeneration(cfg).eval()
    missing, unexpected = model.load_state_dict(ggml_to_hf_state(hp, tensors), strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m for m in missing) or not missing, missing

    pcm = synth.speech_like(SEED_AUDIO)
    fe = WhisperFeatureExtractor(feature_size=hp.n_mels)
    hf_mel = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"][0].astype(np.float32)  # [80, 3000]
    mel_in = hf_mel.astype(np.float16).astype(np.float32)   # the fixture stores the encoder input as f16: keep it self-consistent
    with torch.no_grad():
        enc = model.model.encoder(torch.from_numpy(mel_in)[None]).last_hidden_state[0].numpy()
        out = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([TOKENS]))
        logits = out.logits[0].numpy()
    topk = np.argsort(-logits, axis=1)[:, :16].astype(np.int32)
    topv = np.take_along_axis(logits, topk, axis=1).astype(np.float32)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_toy_golden.npz")
    np.savez_compressed(dst, seed_model=SEED_MODEL, seed_audio=SEED_AUDIO, enc_rows=np.array(ENC_ROWS), tokens=np.array(TOKENS, np.int32),
                        mel_cols=np.arange(0, 2990, 13), hf_mel=hf_mel[:, 0:2990:13].astype(np.float32), hf_mel_in=hf_mel.astype(np.float16),
                        enc=enc[ENC_ROWS].astype(np.float32), enc_absmax=np.float32(np.abs(enc).max()), topk=topk, topv=topv,
                        logit_std=np.float32(logits.std()), filt_sum=np.float64(filt.astype(np.float64).sum()))
    print("wrote", dst, os.path.getsize(dst), "bytes")


# ------------------------------------------------------------------------------------------------------------------------------------
# round 2: more shapes (128 mels + multilingual vocabulary, the real tiny.en, large-v3's width) and the logits RULES
# ------------------------------------------------------------------------------------------------------------------------------------
SHAPES = {"toy": 11, "tiny.en": 12, "wide2": 13}        # preset -> model seed


def hf_model_for(hp, tensors):
    cfg = WhisperConfig(vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, d_model=hp.n_audio_state, encoder_layers=hp.n_audio_layer,
                        encoder_attention_heads=hp.n_audio_head, decoder_layers=hp.n_text_layer, decoder_attention_heads=hp.n_text_head,
                        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state, max_source_positions=hp.n_audio_ctx,
                        max_target_positions=hp.n_text_ctx, activation_function="gelu", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0)
    model = WhisperForConditionalGeneration(cfg).eval()
    missing, unexpected = model.load_state_dict(ggml_to_hf_state(hp, tensors), strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m for m in missing) or not missing, missing
    return model


def shapes_fixture():
    """Encoder rows + per-step top-16 logits of HF Whisper on the seeded `toy` / `tiny.en` / `wide2` models.  The encoder input is the
    ORACLE's own whisper.cpp-style log-mel of the seeded audio (frames [0, 3000)), rounded to f16 so both sides see the same numbers; the
    test recomputes it, nothing but the outputs is stored.  For the 128-bin filterbank the HF feature extractor's mel is stored too."""
    from oracle import binding as orc
    out = {}
    tmp = tempfile.mkdtemp()
    for name, seed in SHAPES.items():
        path = os.path.join(tmp, f"{name}.bin")
        ggml_io.write_model(path, name, seed=seed)
        hp, filt, vocab, tensors = ggml_io.read_model(path)
        model = hf_model_for(hp, tensors)
        om = orc.OracleModel(path)
        pcm = synth.speech_like(SEED_AUDIO + 1)
        mel = om.log_mel(pcm)[:, :3000].astype(np.float16).astype(np.float32)
        multilingual = hp.n_vocab >= 51865
        sot = 50258 if multilingual else 50257
        beg = om.beg
        toks = ([sot, sot + 1 + 3, om.transcribe] if multilingual else [sot]) + [beg + 2, 1234, 777, 31000, 42, beg + 40, beg + 40, 9]
        with torch.no_grad():
            enc = model.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state[0].numpy()
            logits = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([toks])).logits[0].numpy()
        topk = np.argsort(-logits, axis=1)[:, :16].astype(np.int32)
        k = name.replace(".", "_")
        out[f"{k}_seed"] = seed
        out[f"{k}_tokens"] = np.array(toks, np.int32)
        out[f"{k}_n_prompt"] = 3 if multilingual else 1
        out[f"{k}_enc"] = enc[ENC_ROWS].astype(np.float32)
        out[f"{k}_enc_absmax"] = np.float32(np.abs(enc).max())
        out[f"{k}_topk"] = topk
        out[f"{k}_topv"] = np.take_along_axis(logits, topk, axis=1).astype(np.float32)
        out[f"{k}_logit_std"] = np.float32(logits.std())
        if name == "toy":     # 128 mel bins: the HF feature extractor on the same audio, sampled columns
            fe = WhisperFeatureExtractor(feature_size=hp.n_mels)
            hf_mel = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"][0].astype(np.float32)
            out["toy_mel_cols"] = np.arange(0, 2990, 13)
            out["toy_hf_mel"] = hf_mel[:, 0:2990:13].astype(np.float32)
        om.close()
        print(name, "enc absmax", float(np.abs(enc).max()), "logit std", float(logits.std()))
    out["seed_audio"] = SEED_AUDIO + 1
    out["enc_rows"] = np.array(ENC_ROWS)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_shapes_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests_golden_cases import rule_cases, rule_logits  # noqa: E402  (the same seeded cases the test replays)


def rules_fixture():
    """OpenAI's decoding rules as HF transformers implements them (SuppressTokens*, WhisperTimeStampLogitsProcessor) on seeded logits x token
    histories, for both vocabularies.  Stored per case: the -inf mask (packed bits) and the processed scores' log-softmax at the 24 most
    likely surviving tokens.  whisper.cpp restates the same rules with three documented differences (tests/test_oracle_golden.py)."""
    from transformers import GenerationConfig
    from transformers.generation.logits_process import SuppressTokensAtBeginLogitsProcessor, SuppressTokensLogitsProcessor, WhisperTimeStampLogitsProcessor
    out = {}
    for tag, n_vocab in (("en", 51864), ("ml", 51866)):
        multilingual = n_vocab >= 51865
        eot = 50257 if multilingual else 50256
        sot = eot + 1
        n_lang = n_vocab - 51765 - (1 if multilingual else 0)
        dt = n_lang - 98 if multilingual else 0
        translate, transcribe, solm, prev, nosp, not_, beg = [x + dt for x in (50357, 50358, 50359, 50360, 50361, 50362, 50363)]   # whisper.cpp's vocab offsets
        prompt = [sot, sot + 1, transcribe] if multilingual else [sot]
        gc = GenerationConfig(eos_token_id=eot, no_timestamps_token_id=not_, max_initial_timestamp_index=50)
        procs = [SuppressTokensLogitsProcessor([sot, nosp, solm, translate, transcribe, prev] + [sot + 1 + i for i in range(n_lang)]),
                 SuppressTokensAtBeginLogitsProcessor([220, eot], len(prompt)),
                 WhisperTimeStampLogitsProcessor(gc, len(prompt), _detect_timestamp_from_logprob=True)]
        rng = np.random.default_rng(1234 + n_vocab)
        masks, tops, topv = [], [], []
        for hist, trial in rule_cases(beg, eot):
            raw = rule_logits(rng, n_vocab, beg, eot, trial)
            ids = torch.tensor([prompt + hist])
            sc = torch.from_numpy(raw)[None].clone()
            for pr in procs:
                sc = pr(ids, sc)
            lp = torch.log_softmax(sc.float(), dim=-1)[0].numpy()
            masks.append(np.packbits(np.isinf(lp)))
            order = np.argsort(-lp)[:24].astype(np.int32)
            tops.append(order)
            topv.append(lp[order].astype(np.float32))
        out[f"{tag}_mask"] = np.stack(masks)
        out[f"{tag}_top"] = np.stack(tops)
        out[f"{tag}_topv"] = np.stack(topv)
        out[f"{tag}_n_vocab"] = n_vocab
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_rules_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


# ------------------------------------------------------------------------------------------------------------------------------------
# round 5: fixtures the HIP ENGINE can be held to directly (tests/test_gpu_golden.py), and a sequence-level one
# ------------------------------------------------------------------------------------------------------------------------------------
def hf_model_tanh(hp, tensors):
    """HF Whisper with `activation_function="gelu_new"` -- the tanh form, ggml's formula (/root/reference/resources/ggml-metal.metal:262-277
    kernel_gelu: 0.5 x (1 + tanh(sqrt(2/pi) x (1 + 0.044715 x^2)))) -- which is the only GELU the engine implements."""
    cfg = WhisperConfig(vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, d_model=hp.n_audio_state, encoder_layers=hp.n_audio_layer,
                        encoder_attention_heads=hp.n_audio_head, decoder_layers=hp.n_text_layer, decoder_attention_heads=hp.n_text_head,
                        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state, max_source_positions=hp.n_audio_ctx,
                        max_target_positions=hp.n_text_ctx, activation_function="gelu_new", dropout=0.0, attention_dropout=0.0,
                        activation_dropout=0.0)
    model = WhisperForConditionalGeneration(cfg).eval()
    missing, unexpected = model.load_state_dict(ggml_to_hf_state(hp, tensors), strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in m for m in missing) or not missing, missing
    return model


TANH_SHAPES = {"toy": 31, "tiny.en": 32, "wide2": 33}


def tanh_fixture():
    """hf_tanh_golden.npz: as shapes_fixture, with the tanh GELU, so that the same vectors apply to the oracle's default mode AND to the HIP
    engine (f16 operands): sampled columns of the HF feature extractor's log-mel for both filterbanks (80 and 128 bins), encoder rows, and the
    top-16 logits of every position of a prompt + text + timestamp sequence.  The encoder input is the oracle's whisper.cpp-style log-mel of
    the seeded audio (frames [0, 3000)) rounded to f16; the tests recompute it."""
    from oracle import binding as orc
    out = {}
    tmp = tempfile.mkdtemp()
    for name, seed in TANH_SHAPES.items():
        path = os.path.join(tmp, f"{name}.bin")
        ggml_io.write_model(path, name, seed=seed)
        hp, filt, vocab, tensors = ggml_io.read_model(path)
        model = hf_model_tanh(hp, tensors)
        om = orc.OracleModel(path)
        pcm = synth.speech_like(SEED_AUDIO + 2)
        mel = om.log_mel(pcm)[:, :3000].astype(np.float16).astype(np.float32)
        multilingual = hp.n_vocab >= 51865
        sot, beg = om.sot, om.beg
        toks = ([sot, sot + 1 + 7, om.transcribe] if multilingual else [sot]) + [beg + 1, 4321, 707, 30999, 24, beg + 33, beg + 33, 11, 2600]
        with torch.no_grad():
            enc = model.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state[0].numpy()
            logits = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([toks])).logits[0].numpy()
        topk = np.argsort(-logits, axis=1)[:, :16].astype(np.int32)
        k = name.replace(".", "_")
        out[f"{k}_seed"] = seed
        out[f"{k}_tokens"] = np.array(toks, np.int32)
        out[f"{k}_n_prompt"] = 3 if multilingual else 1
        out[f"{k}_enc"] = enc[ENC_ROWS].astype(np.float32)
        out[f"{k}_enc_absmax"] = np.float32(np.abs(enc).max())
        out[f"{k}_enc_rms"] = np.float32(np.sqrt((enc.astype(np.float64) ** 2).mean()))
        out[f"{k}_topk"] = topk
        out[f"{k}_topv"] = np.take_along_axis(logits, topk, axis=1).astype(np.float32)
        out[f"{k}_logit_std"] = np.float32(logits.std())
        if multilingual:   # language detection as HF does it (WhisperGenerationMixin.detect_language): the [sot] step's logits over the language tokens
            n_lang = hp.n_vocab - 51765 - 1
            with torch.no_grad():
                l0 = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([[sot]])).logits[0, -1].numpy()
            ll = l0[sot + 1: sot + 1 + n_lang]
            order = np.argsort(-ll)
            out[f"{k}_lang_id"] = int(order[0])
            out[f"{k}_lang_margin"] = np.float32(ll[order[0]] - ll[order[1]])
        if name in ("toy", "tiny.en"):     # 128 and 80 mel bins: the HF feature extractor on the same audio, sampled columns
            fe = WhisperFeatureExtractor(feature_size=hp.n_mels)
            hf_mel = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"][0].astype(np.float32)
            out[f"{k}_hf_mel"] = hf_mel[:, 0:2990:13].astype(np.float32)
        om.close()
        print(name, "enc absmax", float(np.abs(enc).max()), "logit std", float(logits.std()))
    out["seed_audio"] = SEED_AUDIO + 2
    out["enc_rows"] = np.array(ENC_ROWS)
    out["mel_cols"] = np.arange(0, 2990, 13)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_tanh_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


# (preset, model seed, audio seed, ts_rate): ts_rate 24 = timestamps advance 7.68 s per pair and pass 29 s before EOT, so the window ends by
# whisper.cpp's `seek + seek_delta + 100 >= seek_end` rule (on the FIRST timestamp of that pair) while HF decodes on
GENERATE_CASES = [("toy256", 41, 3, 12.0), ("toy256", 42, 8, 12.0), ("tiny.en", 43, 4, 12.0), ("tiny.en", 44, 9, 12.0), ("wide2", 45, 5, 12.0),
                  ("wide2", 46, 6, 12.0), ("toy256", 47, 10, 24.0), ("tiny.en", 48, 11, 24.0),
                  # cases 8, 9 (round 5): another language token and the translate task in the prompt -- [sot, <|de|>, <|translate|>], [sot, <|fr|>, <|transcribe|>]
                  ("toy256", 51, 16, 12.0, "de", True), ("wide2", 52, 17, 12.0, "fr", False)]
LANG_INDEX = {"en": 0, "de": 2, "fr": 6}


def window0_len(ids, beg, eot):
    """How many of HF's ids whisper.cpp's decode loop samples before it ends the FIRST window (seek 0, seek_end 3000): EOT, or a timestamp
    that brings seek_delta within 100 frames of the window's end (whisper_full_with_state: `has_ts && seek + seek_delta + 100 >= seek_end`)."""
    has_ts, seek_delta = False, 3000
    for i, t in enumerate(ids):
        if t > beg:
            seek_delta, has_ts = 2 * (t - beg), True
        if t == eot or (has_ts and seek_delta + 100 >= 3000):
            return i + 1
    return len(ids)


def generate_fixture():
    """hf_generate_golden.npz: SEQUENCE level.  HF `WhisperForConditionalGeneration.generate(return_timestamps=True, do_sample=False)` -- OpenAI's
    decoding rules as transformers implements them, greedy, KV-cached -- on seeded `natural`-style models (speaksense_amd.ggml_io.NATURAL: the
    decoder emits text, increasing timestamp pairs and ends with EOT after an audio-dependent number of tokens) with whisper.cpp's suppress
    lists (the parameters of /root/reference/src/asr/whisper.rs:131-173 at temperature 0).  Stored per case: every id of the FIRST 30 s window
    (the raw result of that window: the tail after the last closed timestamp pair included), how many of them whisper.cpp's loop samples
    before it ends the window, and the (start, end) of the window's segments in centiseconds as HF reports them.  The tests hold the oracle
    (exact-f32 mode, COMPAT_OPENAI_TS_RULES) to identical ids and segment times, and the HIP engine to identical ids or a forced replay.
    Later windows are not stored: HF zero-fills the log-mel past the audio, whisper.cpp continues in its own padded spectrogram.  (Tried in round 5 with
    62 s of audio, whisper.cpp's padded mel as `input_features`, `language` / `task` arguments and `condition_on_prev_tokens=True` -- what whisper.cpp
    does inside one call: window 0, the seek advance to 23.04 s and window 1's segment (23.12 s .. 49.92 s) agree with the oracle; window 2's first pick
    does not, because HF / OpenAI condition on the tokens of the SEGMENTS -- without the closing timestamp of the last pair -- where whisper.cpp keeps
    every token up to result_len: DESIGN.md section 2a row 11.)"""
    from transformers import GenerationConfig
    from oracle import binding as orc
    out = {}
    tmp = tempfile.mkdtemp()
    for ci, case in enumerate(GENERATE_CASES):
        name, seed, aseed, ts_rate = case[:4]
        lang, translate = (case[4], case[5]) if len(case) > 4 else ("en", False)
        path = os.path.join(tmp, f"{name}-{seed}.bin")
        ggml_io.write_model(path, name, seed=seed, **dict(ggml_io.NATURAL, ts_rate=ts_rate))
        hp, filt, vocab, tensors = ggml_io.read_model(path)
        model = hf_model_tanh(hp, tensors)
        om = orc.OracleModel(path)
        pcm = synth.speech_like(aseed)
        mel = om.log_mel(pcm)[:, :3000].astype(np.float32)
        multilingual = hp.n_vocab >= 51865
        eot, sot, beg = om.eot, om.sot, om.beg
        n_lang = hp.n_vocab - 51765 - (1 if multilingual else 0)
        prompt = [sot, sot + 1 + LANG_INDEX[lang], om.translate if translate else om.transcribe] if multilingual else [sot]
        suppress = [sot, om.nosp, om.solm, om.translate, om.transcribe, om.prev] + [sot + 1 + i for i in range(n_lang)]      # whisper_process_logits' list
        gc = GenerationConfig(eos_token_id=eot, pad_token_id=eot, bos_token_id=eot, decoder_start_token_id=sot, no_timestamps_token_id=om.not_,
                              max_initial_timestamp_index=50, suppress_tokens=suppress, begin_suppress_tokens=[220, eot], max_length=448,
                              is_multilingual=multilingual, return_timestamps=True, do_sample=False, num_beams=1)
        if multilingual:
            gc.lang_to_id = {"<|en|>": sot + 1}
            gc.task_to_id = {"transcribe": om.transcribe, "translate": om.translate}
        model.generation_config = gc
        with torch.no_grad():
            res = model.generate(input_features=torch.from_numpy(mel)[None], decoder_input_ids=torch.tensor([prompt]), return_timestamps=True,
                                 do_sample=False, num_beams=1, max_new_tokens=224, return_segments=True)
        segs = res["segments"][0]
        first = segs[0]["result"]
        raw = first["sequences"] if isinstance(first, dict) else first
        raw = [int(t) for t in (raw[0] if raw.dim() == 2 else raw)]
        if raw[:len(prompt)] == prompt:
            raw = raw[len(prompt):]
        w0 = [sg for sg in segs if sg["result"] is segs[0]["result"]]
        n0 = window0_len(raw, beg, eot)
        k = f"c{ci}"
        out[f"{k}_preset"], out[f"{k}_seed"], out[f"{k}_audio"], out[f"{k}_ts_rate"] = name, seed, aseed, ts_rate
        out[f"{k}_language"], out[f"{k}_translate"] = lang, int(translate)
        out[f"{k}_ids"] = np.array(raw, np.int32)
        out[f"{k}_n_window0"] = n0
        out[f"{k}_seg_t0"] = np.array([round(100 * float(sg["start"])) for sg in w0], np.int64)
        out[f"{k}_seg_t1"] = np.array([round(100 * float(sg["end"])) for sg in w0], np.int64)
        # the generator checks what the CPU test will check, so that a fixture that cannot be met is never committed
        ref = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language=lang, translate=int(translate), temperature_inc=0.0))
        tr = [int(t) for t in ref["trace"]]
        n_ts = sum(t >= beg for t in raw[:n0])
        print(name, seed, aseed, "HF window 0:", len(raw), "ids,", n0, "sampled by whisper.cpp's loop,", n_ts, "timestamps,", len(w0), "segments; oracle trace",
              len(tr), "equal prefix", tr[:n0] == raw[:n0], "segments", [(s["t0"], s["t1"]) for s in ref["segments"]][:len(w0)],
              list(zip(out[f"{k}_seg_t0"].tolist(), out[f"{k}_seg_t1"].tolist())))
        om.close()
    out["n_cases"] = len(GENERATE_CASES)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_generate_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


# (preset, model seed, audio seed, audio_ctx): shortened encoder contexts, multiples of 4 (the engine's restriction), none a multiple of 64
AUDIO_CTX_CASES = [("tiny.en", 61, 21, 752), ("toy256", 62, 22, 500), ("wide2", 63, 23, 1000)]


def audio_ctx_fixture():
    """hf_audio_ctx_golden.npz: `whisper_full_params.audio_ctx` < n_audio_ctx (whisper.cpp's exp_n_audio_ctx: the encoder covers the first audio_ctx
    positions only, the decoder attends over that many keys).  The independent statement of that semantics: an HF model whose `max_source_positions`
    IS audio_ctx -- the first audio_ctx rows of the positional embedding, 2 audio_ctx mel frames as `input_features` (the convolution's zero padding
    then sits right behind frame 2 audio_ctx - 1, as in whisper.cpp's truncated conv input).  Stored per case: encoder rows (the last rows included:
    they are the ones a wrong cut would change), top-16 logits of a teacher-forced sequence, and HF `generate`'s first-window ids / segment times."""
    from dataclasses import replace
    from transformers import GenerationConfig
    from oracle import binding as orc
    out = {}
    tmp = tempfile.mkdtemp()
    for ci, (name, seed, aseed, actx) in enumerate(AUDIO_CTX_CASES):
        path = os.path.join(tmp, f"{name}-{seed}.bin")
        ggml_io.write_model(path, name, seed=seed, **ggml_io.NATURAL)
        hp, filt, vocab, tensors = ggml_io.read_model(path)
        hp_s = ggml_io.HParams(*(hp.astuple()[:1] + (actx,) + hp.astuple()[2:]))
        assert hp_s.n_audio_ctx == actx and hp_s.n_audio_state == hp.n_audio_state
        t_s = dict(tensors)
        t_s["encoder.positional_embedding"] = tensors["encoder.positional_embedding"][:actx].copy()
        model = hf_model_tanh(hp_s, t_s)
        om = orc.OracleModel(path)
        pcm = synth.speech_like(aseed)
        mel = om.log_mel(pcm)[:, :2 * actx].astype(np.float32)
        multilingual = hp.n_vocab >= 51865
        eot, sot, beg = om.eot, om.sot, om.beg
        n_lang = hp.n_vocab - 51765 - (1 if multilingual else 0)
        prompt = [sot, sot + 1, om.transcribe] if multilingual else [sot]
        toks = prompt + [beg + 1, 4321, 707, 30999, 24, beg + 33, beg + 33, 11, 2600]
        rows = sorted({0, 1, 17, actx // 2, actx - 3, actx - 2, actx - 1})
        with torch.no_grad():
            enc = model.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state[0].numpy()
            logits = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([toks])).logits[0].numpy()
        topk = np.argsort(-logits, axis=1)[:, :16].astype(np.int32)
        suppress = [sot, om.nosp, om.solm, om.translate, om.transcribe, om.prev] + [sot + 1 + i for i in range(n_lang)]
        gc = GenerationConfig(eos_token_id=eot, pad_token_id=eot, bos_token_id=eot, decoder_start_token_id=sot, no_timestamps_token_id=om.not_,
                              max_initial_timestamp_index=50, suppress_tokens=suppress, begin_suppress_tokens=[220, eot], max_length=448,
                              is_multilingual=multilingual, return_timestamps=True, do_sample=False, num_beams=1)
        if multilingual:
            gc.lang_to_id = {"<|en|>": sot + 1}
            gc.task_to_id = {"transcribe": om.transcribe, "translate": om.translate}
        model.generation_config = gc
        with torch.no_grad():
            res = model.generate(input_features=torch.from_numpy(mel)[None], decoder_input_ids=torch.tensor([prompt]), return_timestamps=True,
                                 do_sample=False, num_beams=1, max_new_tokens=224, return_segments=True)
        segs = res["segments"][0]
        first = segs[0]["result"]
        raw = first["sequences"] if isinstance(first, dict) else first
        raw = [int(t) for t in (raw[0] if raw.dim() == 2 else raw)]
        if raw[:len(prompt)] == prompt:
            raw = raw[len(prompt):]
        w0 = [sg for sg in segs if sg["result"] is segs[0]["result"]]
        n0 = window0_len(raw, beg, eot)
        k = f"c{ci}"
        out[f"{k}_preset"], out[f"{k}_seed"], out[f"{k}_audio"], out[f"{k}_audio_ctx"] = name, seed, aseed, actx
        out[f"{k}_rows"], out[f"{k}_enc"] = np.array(rows, np.int32), enc[rows].astype(np.float32)
        out[f"{k}_enc_absmax"] = np.float32(np.abs(enc).max())
        out[f"{k}_tokens"], out[f"{k}_n_prompt"] = np.array(toks, np.int32), len(prompt)
        out[f"{k}_topk"], out[f"{k}_topv"] = topk, np.take_along_axis(logits, topk, axis=1).astype(np.float32)
        out[f"{k}_logit_std"] = np.float32(logits.std())
        out[f"{k}_ids"], out[f"{k}_n_window0"] = np.array(raw, np.int32), n0
        out[f"{k}_seg_t0"] = np.array([round(100 * float(sg["start"])) for sg in w0], np.int64)
        out[f"{k}_seg_t1"] = np.array([round(100 * float(sg["end"])) for sg in w0], np.int64)
        # what the CPU test will check
        e_or = om.encode(om.log_mel(pcm), 0, orc.MODE_F32, audio_ctx=actx)
        ref = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language="en", temperature_inc=0.0, audio_ctx=actx, duration_ms=30000))
        tr = [int(t) for t in ref["trace"]]
        print(name, seed, aseed, "audio_ctx", actx, "encoder max|oracle - HF| / absmax", float(np.abs(e_or - enc).max() / np.abs(enc).max()), "HF window 0:", len(raw), "ids,",
              n0, "sampled by whisper.cpp's loop; oracle equal prefix", tr[:n0] == raw[:n0], "segments", [(s_["t0"], s_["t1"]) for s_ in ref["segments"]][:len(w0)],
              list(zip(out[f"{k}_seg_t0"].tolist(), out[f"{k}_seg_t1"].tolist())))
        om.close()
    out["n_cases"] = len(AUDIO_CTX_CASES)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_audio_ctx_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


def non_speech_fixture():
    """hf_non_speech_ids.npz: transformers' NON_SPEECH_TOKENS / NON_SPEECH_TOKENS_MULTI (configuration_whisper.py: the default `suppress_tokens` of the
    English and multilingual checkpoints = openai/whisper tokenizer.py non_speech_tokens, as ids of the REAL vocabularies).  The vocabularies are not in
    the container, but the first 94 ids of every GPT-2 byte-level vocabulary are the printable ASCII characters '!' .. '~' in order (bytes_to_unicode), so
    the ids below 94 say which single ASCII characters are on the list -- the part of whisper.cpp's symbol table the tests can hold to an independent source."""
    import transformers.models.whisper.configuration_whisper as c
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_non_speech_ids.npz")
    np.savez_compressed(dst, en=np.array(c.NON_SPEECH_TOKENS, np.int32), multi=np.array(c.NON_SPEECH_TOKENS_MULTI, np.int32))
    print("wrote", dst, len(c.NON_SPEECH_TOKENS), len(c.NON_SPEECH_TOKENS_MULTI), "ids; single ASCII characters:",
          "".join(chr(33 + i) for i in c.NON_SPEECH_TOKENS if i < 94))


def languages_fixture():
    """hf_languages.txt: transformers' LANGUAGES (tokenization_whisper.py = openai/whisper tokenizer.py), one "code name" per line in id order: the
    language token of id i is sot + 1 + i, so the ORDER is part of the path (prompt_init, language detection)."""
    from transformers.models.whisper.tokenization_whisper import LANGUAGES
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_languages.txt")
    with open(dst, "w") as f:
        for code, name in LANGUAGES.items():
            f.write(f"{code} {name}\n")
    print("wrote", dst, len(LANGUAGES), "languages")


def large_v3_fixture():
    """hf_large_v3_golden.npz: the FULL-DEPTH shape (32 + 32 layers, d = 1280, 20 heads, 128 mels, 51 866 tokens) -- the seeded synthetic large-v3 model
    bench.py and tests/test_gpu_large_v3.py use (`write_model("large-v3", seed=0)`), loaded into HF with the tanh GELU: encoder rows, per-step top-16
    logits of a prompt + text + timestamp sequence, and the detected language.  ~10 minutes and ~20 GB of RAM on the build container's CPU; the tests
    apply the vectors to the HIP engine directly (and to the oracle under -m gpu only: its f32 pass over this model takes minutes)."""
    from oracle import binding as orc
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "large-v3.bin")
    ggml_io.write_model(path, "large-v3", seed=0)
    hp, filt, vocab, tensors = ggml_io.read_model(path)
    model = hf_model_tanh(hp, tensors)
    del tensors
    om = orc.OracleModel(path)
    pcm = synth.speech_like(SEED_AUDIO + 5)
    mel = om.log_mel(pcm)[:, :3000].astype(np.float16).astype(np.float32)
    sot, beg = om.sot, om.beg
    toks = [sot, sot + 1, om.transcribe, beg, 4321, 707, 30999, 24, beg + 33, beg + 33, 11]
    with torch.no_grad():
        enc = model.model.encoder(torch.from_numpy(mel)[None]).last_hidden_state[0].numpy()
        logits = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([toks])).logits[0].numpy()
        l0 = model(encoder_outputs=(torch.from_numpy(enc)[None],), decoder_input_ids=torch.tensor([[sot]])).logits[0, -1].numpy()
    n_lang = hp.n_vocab - 51765 - 1
    ll = l0[sot + 1: sot + 1 + n_lang]
    order = np.argsort(-ll)
    topk = np.argsort(-logits, axis=1)[:, :16].astype(np.int32)
    out = dict(seed=0, seed_audio=SEED_AUDIO + 5, tokens=np.array(toks, np.int32), n_prompt=3, enc_rows=np.array(ENC_ROWS), enc=enc[ENC_ROWS].astype(np.float32),
               enc_absmax=np.float32(np.abs(enc).max()), topk=topk, topv=np.take_along_axis(logits, topk, axis=1).astype(np.float32),
               logit_std=np.float32(logits.std()), lang_id=int(order[0]), lang_margin=np.float32(ll[order[0]] - ll[order[1]]))
    om.close()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_large_v3_golden.npz")
    np.savez_compressed(dst, **out)
    print("large-v3: enc absmax", float(np.abs(enc).max()), "logit std", float(logits.std()), "lang", int(order[0]), "margin", float(ll[order[0]] - ll[order[1]]))
    print("wrote", dst, os.path.getsize(dst), "bytes")


def generate_large_v3_fixture(audio_seeds=(5003, 5010)):
    """hf_generate_large_v3_golden.npz: the first-window sequence fixture at FULL depth -- HF `generate(return_timestamps=True, do_sample=False)` on the
    natural-EOT synthetic large-v3 model bench.py's `mode_n` and tests/test_gpu_bench_config.py use (`write_model("large-v3", seed=0, **NATURAL)`).
    Stored per audio: HF's ids of the first window, how many whisper.cpp's loop samples, HF's segment times.  The generator also runs the oracle (exact
    f32, COMPAT_OPENAI_TS_RULES; minutes per case) and reports whether it equals HF; the CPU suite does not (too slow), the GPU suite holds the ENGINE
    to the ids directly (tests/test_gpu_bench_config.py)."""
    from transformers import GenerationConfig
    from oracle import binding as orc
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "large-v3-natural.bin")
    ggml_io.write_model(path, "large-v3", seed=0, **ggml_io.NATURAL)
    hp, filt, vocab, tensors = ggml_io.read_model(path)
    model = hf_model_tanh(hp, tensors)
    del tensors
    om = orc.OracleModel(path)
    eot, sot, beg = om.eot, om.sot, om.beg
    n_lang = hp.n_vocab - 51765 - 1
    prompt = [sot, sot + 1, om.transcribe]
    suppress = [sot, om.nosp, om.solm, om.translate, om.transcribe, om.prev] + [sot + 1 + i for i in range(n_lang)]
    gc = GenerationConfig(eos_token_id=eot, pad_token_id=eot, bos_token_id=eot, decoder_start_token_id=sot, no_timestamps_token_id=om.not_,
                          max_initial_timestamp_index=50, suppress_tokens=suppress, begin_suppress_tokens=[220, eot], max_length=448,
                          is_multilingual=True, return_timestamps=True, do_sample=False, num_beams=1)
    gc.lang_to_id = {"<|en|>": sot + 1}
    gc.task_to_id = {"transcribe": om.transcribe, "translate": om.translate}
    model.generation_config = gc
    out = {"n_cases": len(audio_seeds)}
    for ci, aseed in enumerate(audio_seeds):
        pcm = synth.speech_like(aseed)
        mel = om.log_mel(pcm)[:, :3000].astype(np.float32)
        with torch.no_grad():
            res = model.generate(input_features=torch.from_numpy(mel)[None], decoder_input_ids=torch.tensor([prompt]), return_timestamps=True,
                                 do_sample=False, num_beams=1, max_new_tokens=224, return_segments=True)
        segs = res["segments"][0]
        first = segs[0]["result"]
        raw = first["sequences"] if isinstance(first, dict) else first
        raw = [int(t) for t in (raw[0] if raw.dim() == 2 else raw)]
        if raw[:len(prompt)] == prompt:
            raw = raw[len(prompt):]
        w0 = [sg for sg in segs if sg["result"] is segs[0]["result"]]
        n0 = window0_len(raw, beg, eot)
        k = f"c{ci}"
        out[f"{k}_audio"], out[f"{k}_ids"], out[f"{k}_n_window0"] = aseed, np.array(raw, np.int32), n0
        out[f"{k}_seg_t0"] = np.array([round(100 * float(sg["start"])) for sg in w0], np.int64)
        out[f"{k}_seg_t1"] = np.array([round(100 * float(sg["end"])) for sg in w0], np.int64)
        ref = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES).full(pcm, orc.default_params(language="en", temperature_inc=0.0, duration_ms=30000))
        tr = [int(t) for t in ref["trace"]]
        print("large-v3 natural, audio", aseed, "HF window 0:", len(raw), "ids,", n0, "sampled by whisper.cpp's loop,", sum(t >= beg for t in raw[:n0]), "timestamps; oracle equal prefix:",
              tr[:n0] == raw[:n0], "segments equal:", [(s_["t0"], s_["t1"]) for s_ in ref["segments"]][:len(w0)] == list(zip(out[f"{k}_seg_t0"].tolist(), out[f"{k}_seg_t1"].tolist())))
    om.close()
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_generate_large_v3_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


def wcpp_window(ids, beg, eot, seek, seek_end):
    """whisper.cpp's decode loop over HF's ids of ONE window that starts at frame `seek`: (ids it samples before it ends the window, seek_delta, regular) --
    regular = the window ends with EOT after at least one closed timestamp pair and nothing but that pair decides the advance (no single trailing
    timestamp, no window-end cut): the case in which OpenAI's / HF's seek advance and whisper.cpp's coincide."""
    has_ts, seek_delta, result_len, n = False, 3000, 0, len(ids)
    for i, t in enumerate(ids):
        if t > beg:
            seek_delta, has_ts, result_len = 2 * (t - beg), True, i + 1
        if t == eot or (has_ts and seek + seek_delta + 100 >= seek_end):
            n = i + 1
            break
    kept = ids[:result_len]
    regular = (n <= len(ids) and ids[n - 1] == eot and result_len >= 2 and kept[-1] >= beg and kept[-2] >= beg and kept[-1] == kept[-2] and
               seek + seek_delta + 100 < seek_end)
    return n, seek_delta, regular


LONG_CANDIDATES = [("wide2", 46, 6), ("toy256", 41, 3), ("tiny.en", 44, 9), ("toy256", 47, 12), ("tiny.en", 48, 13), ("wide2", 45, 7), ("toy256", 49, 14), ("tiny.en", 50, 15)]
LONG_SECONDS = 95


def generate_long_fixture():
    """hf_generate_long_golden.npz: the sequence-level fixture ACROSS window boundaries.  HF `generate(condition_on_prev_tokens=True)` -- inside one call
    whisper.cpp always conditions window n + 1 on `[prev] + history` (`no_context` only clears what a call starts with) -- on 95 s of seeded audio, fed
    whisper.cpp's own padded log-mel so that every window sees the frames whisper.cpp's seek loop sees.  Stored per case: for each of the leading REGULAR
    windows (closed timestamp pair, EOT, no end-of-audio effect: where the two implementations' seek advance coincides; at least two) the ids HF sampled,
    how many of them whisper.cpp's loop samples, and HF's absolute segment times.  The oracle meets them under COMPAT_OPENAI_TS_RULES | COMPAT_OPENAI_HISTORY
    (the history flag covers the one token by which the two histories differ); the generator refuses to store a case it does
    not meet.  What this pins beyond the first-window fixture: the seek advance, the `[prev] + history + [sot ..]` prompt of later windows, the per-window
    reset of the rule state, absolute segment times."""
    from transformers import GenerationConfig
    from oracle import binding as orc
    out, n_kept = {}, 0
    tmp = tempfile.mkdtemp()
    for name, seed, aseed in LONG_CANDIDATES:
        path = os.path.join(tmp, f"{name}-{seed}.bin")
        ggml_io.write_model(path, name, seed=seed, **ggml_io.NATURAL)
        hp, filt, vocab, tensors = ggml_io.read_model(path)
        model = hf_model_tanh(hp, tensors)
        om = orc.OracleModel(path)
        pcm = synth.speech_like(aseed, 16000 * LONG_SECONDS)
        mel = om.log_mel(pcm).astype(np.float32)                 # [n_mel][n_len], n_len = (n + 480000) / 160: 30 s of clamp-floor frames behind the audio
        seek_end = 1 + (len(pcm) + 200 - 400) // 160             # whisper.cpp's n_len_org
        multilingual = hp.n_vocab >= 51865
        eot, sot, beg = om.eot, om.sot, om.beg
        n_lang = hp.n_vocab - 51765 - (1 if multilingual else 0)
        init = [sot, sot + 1, om.transcribe] if multilingual else [sot]
        suppress = [sot, om.nosp, om.solm, om.translate, om.transcribe, om.prev] + [sot + 1 + i for i in range(n_lang)]
        gc = GenerationConfig(eos_token_id=eot, pad_token_id=eot, bos_token_id=eot, decoder_start_token_id=sot, no_timestamps_token_id=om.not_,
                              max_initial_timestamp_index=50, suppress_tokens=suppress, begin_suppress_tokens=[220, eot], max_length=448,
                              is_multilingual=multilingual, return_timestamps=True, do_sample=False, num_beams=1, prev_sot_token_id=om.prev)
        if multilingual:
            gc.lang_to_id = {"<|en|>": sot + 1}
            gc.task_to_id = {"transcribe": om.transcribe, "translate": om.translate}
        model.generation_config = gc
        with torch.no_grad():
            res = model.generate(input_features=torch.from_numpy(mel)[None], **(dict(language="en", task="transcribe") if multilingual else {}),
                                 return_timestamps=True, do_sample=False, num_beams=1, max_new_tokens=224, return_segments=True, condition_on_prev_tokens=True)
        wins = []
        for sg in res["segments"][0]:
            if not wins or sg["result"] is not wins[-1]["result"]:
                r = sg["result"]["sequences"] if isinstance(sg["result"], dict) else sg["result"]
                r = [int(t) for t in (r[0] if r.dim() == 2 else r)]
                k0 = max(i for i in range(len(r) - len(init) + 1) if r[i:i + len(init)] == init) + len(init)      # the new ids follow the LAST [sot ..]
                wins.append(dict(result=sg["result"], ids=r[k0:], seg=[]))
            wins[-1]["seg"].append((round(100 * float(sg["start"])), round(100 * float(sg["end"]))))
        seek, keep = 0, []
        for w in wins:
            n, seek_delta, regular = wcpp_window(w["ids"], beg, eot, seek, seek_end)
            first_ts = next((t for t in w["ids"] if t >= beg), None)
            consistent = first_ts is not None and w["seg"][0][0] == seek + 2 * (first_ts - beg)     # HF's own seek of this window is the one the rule gives
            if not (regular and consistent and seek + 3000 + 100 < seek_end):
                break
            keep.append(dict(ids=w["ids"], n=n, seg=w["seg"], seek=seek))
            seek += seek_delta
        ref = om.new_state(orc.MODE_F32, compat=orc.COMPAT_OPENAI_TS_RULES | orc.COMPAT_OPENAI_HISTORY).full(pcm, orc.default_params(language="en", temperature_inc=0.0))
        tr, pos, ok = [int(t) for t in ref["trace"]], 0, True
        for w in keep:
            ok = ok and tr[pos:pos + w["n"]] == w["ids"][:w["n"]]
            pos += w["n"]
        segs = [sg for w in keep for sg in w["seg"]]
        ok = ok and [(s_["t0"], s_["t1"]) for s_ in ref["segments"]][:len(segs)] == segs
        print(name, seed, aseed, "HF windows", len(wins), "regular leading windows", len(keep), [w["n"] for w in keep], "seeks", [w["seek"] for w in keep], "oracle equal:", ok)
        om.close()
        if len(keep) >= 2 and ok and n_kept < 3:
            k = f"c{n_kept}"
            out[f"{k}_preset"], out[f"{k}_seed"], out[f"{k}_audio"], out[f"{k}_n_windows"] = name, seed, aseed, len(keep)
            for wi, w in enumerate(keep):
                out[f"{k}_w{wi}_ids"] = np.array(w["ids"], np.int32)
                out[f"{k}_w{wi}_n"] = w["n"]
                out[f"{k}_w{wi}_seek"] = w["seek"]
                out[f"{k}_w{wi}_seg"] = np.array(w["seg"], np.int64)
            n_kept += 1
    out["n_cases"], out["seconds"] = n_kept, LONG_SECONDS
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hf_generate_long_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes,", n_kept, "cases")


if __name__ == "__main__":
    which = sys.argv[1:] or ["toy", "shapes", "rules", "tanh", "generate", "generate_long", "audio_ctx", "non_speech", "languages"]
    if "generate_long" in which:
        generate_long_fixture()
    if "audio_ctx" in which:
        audio_ctx_fixture()
    if "non_speech" in which:
        non_speech_fixture()
    if "languages" in which:
        languages_fixture()
    if "large_v3" in which:      # not part of the default list: ~3 minutes
        large_v3_fixture()
    if "generate_large_v3" in which:      # not part of the default list: ~15 minutes (two oracle windows at full depth in exact f32)
        generate_large_v3_fixture()
    if "tanh" in which:
        tanh_fixture()
    if "generate" in which:
        generate_fixture()
    if "toy" in which:
        main()
    if "shapes" in which:
        shapes_fixture()
    if "rules" in which:
        rules_fixture()
