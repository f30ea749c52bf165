"""Cross-check the oracle's network arithmetic against HF transformers' Whisper (runs ONLY in the build container:
transformers/torch-CPU are available here, not on the GPU box).  Seeded random weights, toy width.
Secondary oracle, clearly not the reference: it validates architecture (bias-less K, scaling,
pre-LN, tied logits, conv stem, positional embeddings), not whisper.cpp's exact numerics.

usage: python tests/golden/hf_crosscheck.py      -> prints max abs errors, exits non-zero on mismatch
"""
import os, sys, tempfile
import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from transformers import WhisperConfig, WhisperForConditionalGeneration, WhisperFeatureExtractor
from speaksense_amd import ggml_io, synth
from oracle import binding as orc


def hf_to_ggml_names(hp: ggml_io.HParams, sd):
    t = {}
    t["encoder.conv1.weight"] = sd["model.encoder.conv1.weight"]
    t["encoder.conv1.bias"] = sd["model.encoder.conv1.bias"]
    t["encoder.conv2.weight"] = sd["model.encoder.conv2.weight"]
    t["encoder.conv2.bias"] = sd["model.encoder.conv2.bias"]
    t["encoder.positional_embedding"] = sd["model.encoder.embed_positions.weight"]
    t["encoder.ln_post.weight"] = sd["model.encoder.layer_norm.weight"]
    t["encoder.ln_post.bias"] = sd["model.encoder.layer_norm.bias"]
    t["decoder.positional_embedding"] = sd["model.decoder.embed_positions.weight"]
    t["decoder.token_embedding.weight"] = sd["model.decoder.embed_tokens.weight"]
    t["decoder.ln.weight"] = sd["model.decoder.layer_norm.weight"]
    t["decoder.ln.bias"] = sd["model.decoder.layer_norm.bias"]

    def attn(dst, src):
        for a, b in (("query", "q_proj"), ("key", "k_proj"), ("value", "v_proj"), ("out", "out_proj")):
            t[f"{dst}.{a}.weight"] = sd[f"{src}.{b}.weight"]
            if a != "key":
                t[f"{dst}.{a}.bias"] = sd[f"{src}.{b}.bias"]

    for side, n in (("encoder", hp.n_audio_layer), ("decoder", hp.n_text_layer)):
        for i in range(n):
            s = f"model.{side}.layers.{i}"
            d = f"{side}.blocks.{i}"
            attn(f"{d}.attn", f"{s}.self_attn")
            t[f"{d}.attn_ln.weight"] = sd[f"{s}.self_attn_layer_norm.weight"]
            t[f"{d}.attn_ln.bias"] = sd[f"{s}.self_attn_layer_norm.bias"]
            if side == "decoder":
                attn(f"{d}.cross_attn", f"{s}.encoder_attn")
                t[f"{d}.cross_attn_ln.weight"] = sd[f"{s}.encoder_attn_layer_norm.weight"]
                t[f"{d}.cross_attn_ln.bias"] = sd[f"{s}.encoder_attn_layer_norm.bias"]
            t[f"{d}.mlp_ln.weight"] = sd[f"{s}.final_layer_norm.weight"]
            t[f"{d}.mlp_ln.bias"] = sd[f"{s}.final_layer_norm.bias"]
            t[f"{d}.mlp.0.weight"] = sd[f"{s}.fc1.weight"]
            t[f"{d}.mlp.0.bias"] = sd[f"{s}.fc1.bias"]
            t[f"{d}.mlp.2.weight"] = sd[f"{s}.fc2.weight"]
            t[f"{d}.mlp.2.bias"] = sd[f"{s}.fc2.bias"]
    return {k: v.detach().float().numpy() for k, v in t.items()}


def main():
    torch.manual_seed(0)
    hp = ggml_io.PRESETS["toy.en"]
    cfg = WhisperConfig(vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, d_model=hp.n_audio_state,
                        encoder_layers=hp.n_audio_layer, encoder_attention_heads=hp.n_audio_head,
                        decoder_layers=hp.n_text_layer, decoder_attention_heads=hp.n_text_head,
                        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state,
                        max_source_positions=hp.n_audio_ctx, max_target_positions=hp.n_text_ctx,
                        activation_function="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0)
    model = WhisperForConditionalGeneration(cfg).eval()
    with torch.no_grad():  # make weights non-trivial and exactly f16-representable where the file stores f16
        for n, p in model.named_parameters():
            if p.dim() >= 2 and "embed_positions" not in n:
                p.copy_((p * 3.0).half().float())
            elif "bias" in n:
                p.copy_(0.05 * torch.randn_like(p))
    tensors = hf_to_ggml_names(hp, model.state_dict())
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "toy_hf.bin")
    ggml_io.write_model(path, hp, tensors=tensors)
    om = orc.OracleModel(path)

    # ---- mel: whisper.cpp zero-pads where OpenAI reflect-pads, so compare away from the last frames ----
    pcm = synth.speech_like(1)
    fe = WhisperFeatureExtractor(feature_size=hp.n_mels)
    hf_mel = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"][0]  # [80,3000]
    o_mel = om.log_mel(pcm)
    err_mel = np.abs(o_mel[:, :2990] - hf_mel[:, :2990]).max()
    print(f"mel   max|oracle - HF| (frames 0..2989) = {err_mel:.3e}")

    # ---- encoder (erf GELU to match HF) ----
    mel_in = o_mel[:, :3000]
    with torch.no_grad():
        hf_enc = model.model.encoder(torch.from_numpy(mel_in)[None]).last_hidden_state[0].numpy()
    o_enc = om.encode(o_mel, 0, orc.MODE_F32, gelu_erf=1)
    err_enc = np.abs(o_enc - hf_enc).max()
    print(f"enc   max|oracle - HF| = {err_enc:.3e}  (|enc|max {np.abs(hf_enc).max():.2f})")

    # ---- decoder: prompt of 4 tokens then 3 single-token steps with KV cache ----
    st = om.new_state(orc.MODE_F32, gelu_erf=1)
    st.set_encoder(hf_enc)
    toks = [om.sot, 1234, 777, 50000, 42, 31337, 9]
    with torch.no_grad():
        out = model(encoder_outputs=(torch.from_numpy(hf_enc)[None],), decoder_input_ids=torch.tensor([toks]))
        hf_logits = out.logits[0].numpy()
    lg = st.decode(toks[:4], 0)
    errs = [np.abs(lg - hf_logits[3]).max()]
    for i in range(4, 7):
        lg = st.decode(toks[i:i + 1], i)
        errs.append(np.abs(lg - hf_logits[i]).max())
    err_dec = max(errs)
    print(f"dec   max|oracle - HF| logits = {err_dec:.3e}  (|logit|max {np.abs(hf_logits).max():.2f})")
    ok = err_mel < 2e-4 and err_enc < 2e-3 and err_dec < 2e-3
    print("HF cross-check:", "OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
