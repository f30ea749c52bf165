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


if __name__ == "__main__":
    main()
