"""ggml legacy model-file I/O (the `ggml-*.bin` files SpeakSense points ASR_MODEL_PATH at).

The reference loads these through `WhisperContext::new_with_params` (/root/reference/src/asr/whisper.rs:21-28),
i.e. whisper.cpp's `whisper_model_load` (third party, not in tree; format restated in SURVEY.md §8 a-2).
No real weights exist offline, so tests and bench.py synthesise seeded random models in exactly that format:
the product loader (csrc/model.cpp) and the oracle loader (oracle/whisper_oracle.cpp) both parse the same file.

numpy only -- this module must work on the GPU box (no transformers / no reference there).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, asdict

import numpy as np

GGML_MAGIC = 0x67676D6C
GGML_TYPE_F32 = 0
GGML_TYPE_F16 = 1
GGML_TYPE_Q4_0, GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1, GGML_TYPE_Q8_0 = 2, 3, 6, 7, 8
# whisper.cpp's quantize tool: file ftype -> type of the quantised (2-D) tensors; everything else stays f16 / f32
FTYPE_TO_QTYPE = {2: GGML_TYPE_Q4_0, 3: GGML_TYPE_Q4_1, 7: GGML_TYPE_Q8_0, 8: GGML_TYPE_Q5_0, 9: GGML_TYPE_Q5_1}
FTYPE_BY_NAME = {"f32": 0, "f16": 1, "q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}
Q_BLOCK_BYTES = {GGML_TYPE_Q4_0: 18, GGML_TYPE_Q4_1: 20, GGML_TYPE_Q5_0: 22, GGML_TYPE_Q5_1: 24, GGML_TYPE_Q8_0: 34}
QUANT_SKIP = ("encoder.conv1.bias", "encoder.conv2.bias", "encoder.positional_embedding", "decoder.positional_embedding")


@dataclass
class HParams:
    n_vocab: int = 51864
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4
    n_mels: int = 80
    ftype: int = 1

    def astuple(self):
        return tuple(asdict(self).values())


# write_model(path, shape, **NATURAL): see natural_tensor.  tools/natural_preset_stats.py measures fallbacks / lengths / distinct streams on the oracle.
NATURAL = dict(style="natural", logit_gain=16.0, mlp_boost=4.0, ctrl_rho=0.75, cross_gain=0.2, cross_q_gain=8.0, pos_scale=0.4, self_gain=0.2,
               ts_period=16, ts_rate=12.0, ts_align=0.9, eot_start=4, eot_ramp=160.0, eot_mean=50.0)

PRESETS = {
    # real shapes (SURVEY.md §8 header)
    "tiny.en": HParams(51864, 1500, 384, 6, 4, 448, 384, 6, 4, 80, 1),
    "base.en": HParams(51864, 1500, 512, 8, 6, 448, 512, 8, 6, 80, 1),
    "large-v3": HParams(51866, 1500, 1280, 20, 32, 448, 1280, 20, 32, 128, 1),
    # the other shapes /root/reference/script/download-ggml-model.sh:36-48 can fetch
    "large-v3-turbo": HParams(51866, 1500, 1280, 20, 32, 448, 1280, 20, 4, 128, 1),     # 32 encoder / 4 decoder layers
    "medium": HParams(51865, 1500, 1024, 16, 24, 448, 1024, 16, 24, 80, 1),             # v2-era multilingual vocabulary: 99 languages, ids shifted by one
    "medium.en": HParams(51864, 1500, 1024, 16, 24, 448, 1024, 16, 24, 80, 1),
    "small.en": HParams(51864, 1500, 768, 12, 12, 448, 768, 12, 12, 80, 1),
    "small.en-tdrz": HParams(51864, 1500, 768, 12, 12, 448, 768, 12, 12, 80, 1),        # same file layout; fine-tuned to emit [_SOLM_] at speaker turns (tdrz_enable)
    # toy shapes for fast CPU tests: same context sizes / vocabulary rules, tiny width
    "toy.en": HParams(51864, 1500, 128, 2, 2, 448, 128, 2, 2, 80, 1),
    "toy": HParams(51866, 1500, 128, 2, 2, 448, 128, 2, 2, 128, 1),
    # width 256: the smallest shape that takes the 256x256-tile GEMM path (N % 256 == 0, M >= 1024)
    "toy256": HParams(51866, 1500, 256, 4, 2, 448, 256, 4, 2, 128, 1),
    # large-v3's width, head count, mel count and vocabulary with 2 layers per stack: every large-v3 kernel configuration (split-K plans,
    # LayerNorm register tiling, 20-head attention) at 1/16 of the oracle's cost
    "wide2": HParams(51866, 1500, 1280, 20, 2, 448, 1280, 20, 2, 128, 1),
    # the same idea for the other widths of the download script: medium (d = 1024, 16 heads, 51865-token vocabulary), small.en / small.en-tdrz
    # (d = 768, 12 heads) and large-v3-turbo's asymmetry (more encoder than decoder layers)
    "medium2": HParams(51865, 1500, 1024, 16, 2, 448, 1024, 16, 2, 80, 1),
    "small2.en": HParams(51864, 1500, 768, 12, 2, 448, 768, 12, 2, 80, 1),
    "turbo41": HParams(51866, 1500, 1280, 20, 4, 448, 1280, 20, 1, 128, 1),
}


# ----------------------------------------------------------------------------------------------
# mel filterbank (slaney scale + slaney norm, as shipped inside every ggml whisper file)
# ----------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    log_t = f >= min_log_hz
    mels = np.where(log_t, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)
    return mels


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    log_t = m >= min_log_mel
    freqs = np.where(log_t, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)
    return freqs


def mel_filters(n_mels: int, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """[n_mels, n_fft/2+1] float32, identical to librosa.filters.mel(sr, n_fft, n_mels) (slaney)."""
    n_bins = n_fft // 2 + 1
    fftfreqs = np.linspace(0, sr / 2, n_bins)
    mel_pts = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2), n_mels + 2))
    fdiff = np.diff(mel_pts)
    ramps = mel_pts[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_pts[2 : n_mels + 2] - mel_pts[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


# ----------------------------------------------------------------------------------------------
# ggml block quantisation, QK = 32 (the `*-q5_0` / `*-q5_1` files of /root/reference/script/download-ggml-model.sh:28-51 and the other types
# whisper.cpp's quantize tool writes): numpy restatement of quantize_row_q*_reference / dequantize_row_q*
# ----------------------------------------------------------------------------------------------
def quantize_blocks(x: np.ndarray, qtype: int) -> bytes:
    """x: float32, size a multiple of 32 (rows are multiples of 32 long) -> the ggml block stream."""
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 32)
    nb = x.shape[0]
    if qtype == GGML_TYPE_Q8_0:
        amax = np.abs(x).max(axis=1)
        d = (amax / 127.0).astype(np.float32)
        inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
        q = np.round(x * inv[:, None]).astype(np.int8)
        out = np.zeros((nb, 34), np.uint8)
        out[:, 0:2] = d.astype("<f2").view(np.uint8).reshape(nb, 2)
        out[:, 2:] = q.view(np.uint8)
        return out.tobytes()
    if qtype in (GGML_TYPE_Q4_0, GGML_TYPE_Q5_0):
        half, top = (8, 15) if qtype == GGML_TYPE_Q4_0 else (16, 31)
        idx = np.abs(x).argmax(axis=1)
        mx = x[np.arange(nb), idx]
        d = (mx / -float(half)).astype(np.float32)
        inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
        q = np.minimum(top, (x * inv[:, None] + (half + 0.5)).astype(np.int8)).astype(np.uint8)       # MIN(top, (int8_t)(x*id + half.5))
        dd, mm = d, None
    else:
        top = 15 if qtype == GGML_TYPE_Q4_1 else 31
        mn, mxv = x.min(axis=1), x.max(axis=1)
        d = ((mxv - mn) / float(top)).astype(np.float32)
        inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
        q = ((x - mn[:, None]) * inv[:, None] + 0.5).astype(np.uint8)
        dd, mm = d, mn.astype(np.float32)
    lo, hi = q[:, :16], q[:, 16:]
    qs = ((lo & 0x0F) | ((hi & 0x0F) << 4)).astype(np.uint8)
    parts = [dd.astype("<f2").view(np.uint8).reshape(nb, 2)]
    if mm is not None:
        parts.append(mm.astype("<f2").view(np.uint8).reshape(nb, 2))
    if qtype in (GGML_TYPE_Q5_0, GGML_TYPE_Q5_1):
        j = np.arange(16, dtype=np.uint32)
        qh = (((lo.astype(np.uint32) & 0x10) >> 4) << j[None, :]).sum(axis=1, dtype=np.uint32) | \
             (((hi.astype(np.uint32) & 0x10) >> 4) << (j[None, :] + 16)).sum(axis=1, dtype=np.uint32)
        parts.append(qh.astype("<u4").view(np.uint8).reshape(nb, 4))
    parts.append(qs)
    return np.concatenate(parts, axis=1).tobytes()


def dequantize_blocks(buf, qtype: int, n: int) -> np.ndarray:
    bb = Q_BLOCK_BYTES[qtype]
    nb = n // 32
    b = np.frombuffer(buf, np.uint8, nb * bb).reshape(nb, bb)
    d = b[:, 0:2].copy().view("<f2").astype(np.float32).reshape(nb, 1)
    off = 2
    m = None
    if qtype in (GGML_TYPE_Q4_1, GGML_TYPE_Q5_1):
        m = b[:, 2:4].copy().view("<f2").astype(np.float32).reshape(nb, 1)
        off = 4
    if qtype == GGML_TYPE_Q8_0:
        return (b[:, 2:].copy().view(np.int8).astype(np.float32) * d).reshape(-1)
    if qtype in (GGML_TYPE_Q5_0, GGML_TYPE_Q5_1):
        qh = b[:, off:off + 4].copy().view("<u4").reshape(nb, 1)
        off += 4
        j = np.arange(16, dtype=np.uint32)[None, :]
        xh0 = ((qh >> j) << 4) & 0x10
        xh1 = (qh >> (j + 12)) & 0x10
    else:
        xh0 = xh1 = 0
    qs = b[:, off:off + 16].astype(np.uint32)
    x0 = ((qs & 0x0F) | xh0).astype(np.float32)
    x1 = ((qs >> 4) | xh1).astype(np.float32)
    q = np.concatenate([x0, x1], axis=1)
    if m is not None:
        return (q * d + m).astype(np.float32).reshape(-1)
    sub = 8.0 if qtype == GGML_TYPE_Q4_0 else 16.0
    return ((q - sub) * d).astype(np.float32).reshape(-1)


# ----------------------------------------------------------------------------------------------
# synthetic vocabulary: the file carries the n_vocab_base "text" tokens; whisper.cpp synthesises
# names for the specials above them (restated in both loaders)
# ----------------------------------------------------------------------------------------------
_SYL = ["ka", "to", "mi", "ra", "ne", "so", "lu", "vi", "pa", "chi", "ng", "er", "an", "th", "ou", "we"]


def synth_vocab(n_vocab: int) -> list[bytes]:
    """Deterministic fake BPE table with the properties the path relies on: a " " token exists
    (suppress_blank looks it up), ids < eot are printable UTF-8, some CJK tokens exist so the
    reference's punctuation/promo post-pass (whisper.rs:41-43,175-201) is exercised."""
    multilingual = n_vocab >= 51865
    n_base = 50257 if multilingual else 50256  # tokens stored in the file = [0, eot)
    toks: list[bytes] = []
    cjk = ["吗", "呢", "什么", "好", "太", "啊", "。", "！", "？", "，", "订阅", "點贊", "真是", "怎么"]
    for i in range(n_base):
        if i == 220:
            toks.append(b" ")
        elif i < 256:
            toks.append(bytes([33 + (i % 90)]) + b"%d" % i)
        elif i % 97 == 0:
            toks.append(cjk[(i // 97) % len(cjk)].encode("utf-8"))
        else:
            a, b, c = i % 16, (i // 16) % 16, (i // 256) % 16
            word = _SYL[a] + _SYL[b] + (_SYL[c] if i % 3 == 0 else "")
            toks.append(((" " if i % 2 == 0 else "") + word).encode("utf-8"))
    return toks


# ----------------------------------------------------------------------------------------------
# tensor inventory (names as in whisper.cpp's loader; shapes in PyTorch order)
# ----------------------------------------------------------------------------------------------
def tensor_specs(hp: HParams):
    """Yield (name, shape, ggml_type). 2-D+ weights are f16 when ftype==1, the rest f32."""
    wt = GGML_TYPE_F16 if hp.ftype == 1 else GGML_TYPE_F32
    da, dt = hp.n_audio_state, hp.n_text_state
    yield "encoder.positional_embedding", (hp.n_audio_ctx, da), GGML_TYPE_F32
    yield "encoder.conv1.weight", (da, hp.n_mels, 3), wt
    yield "encoder.conv1.bias", (da, 1), GGML_TYPE_F32
    yield "encoder.conv2.weight", (da, da, 3), wt
    yield "encoder.conv2.bias", (da, 1), GGML_TYPE_F32
    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}."
        yield p + "attn_ln.weight", (da,), GGML_TYPE_F32
        yield p + "attn_ln.bias", (da,), GGML_TYPE_F32
        yield p + "attn.query.weight", (da, da), wt
        yield p + "attn.query.bias", (da,), GGML_TYPE_F32
        yield p + "attn.key.weight", (da, da), wt
        yield p + "attn.value.weight", (da, da), wt
        yield p + "attn.value.bias", (da,), GGML_TYPE_F32
        yield p + "attn.out.weight", (da, da), wt
        yield p + "attn.out.bias", (da,), GGML_TYPE_F32
        yield p + "mlp_ln.weight", (da,), GGML_TYPE_F32
        yield p + "mlp_ln.bias", (da,), GGML_TYPE_F32
        yield p + "mlp.0.weight", (4 * da, da), wt
        yield p + "mlp.0.bias", (4 * da,), GGML_TYPE_F32
        yield p + "mlp.2.weight", (da, 4 * da), wt
        yield p + "mlp.2.bias", (da,), GGML_TYPE_F32
    yield "encoder.ln_post.weight", (da,), GGML_TYPE_F32
    yield "encoder.ln_post.bias", (da,), GGML_TYPE_F32
    yield "decoder.positional_embedding", (hp.n_text_ctx, dt), GGML_TYPE_F32
    yield "decoder.token_embedding.weight", (hp.n_vocab, dt), wt
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}."
        for a in ("attn", "cross_attn"):
            yield p + a + "_ln.weight", (dt,), GGML_TYPE_F32
            yield p + a + "_ln.bias", (dt,), GGML_TYPE_F32
            yield p + a + ".query.weight", (dt, dt), wt
            yield p + a + ".query.bias", (dt,), GGML_TYPE_F32
            yield p + a + ".key.weight", (dt, dt), wt
            yield p + a + ".value.weight", (dt, dt), wt
            yield p + a + ".value.bias", (dt,), GGML_TYPE_F32
            yield p + a + ".out.weight", (dt, dt), wt
            yield p + a + ".out.bias", (dt,), GGML_TYPE_F32
        yield p + "mlp_ln.weight", (dt,), GGML_TYPE_F32
        yield p + "mlp_ln.bias", (dt,), GGML_TYPE_F32
        yield p + "mlp.0.weight", (4 * dt, dt), wt
        yield p + "mlp.0.bias", (4 * dt,), GGML_TYPE_F32
        yield p + "mlp.2.weight", (dt, 4 * dt), wt
        yield p + "mlp.2.bias", (dt,), GGML_TYPE_F32
    yield "decoder.ln.weight", (dt,), GGML_TYPE_F32
    yield "decoder.ln.bias", (dt,), GGML_TYPE_F32


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> np.ndarray:
    inc = np.log(max_timescale) / (channels // 2 - 1)
    inv = np.exp(-inc * np.arange(channels // 2))
    t = np.arange(length)[:, None] * inv[None, :]
    return np.concatenate([np.sin(t), np.cos(t)], axis=1).astype(np.float32)


def _special_ids(hp: HParams):
    multilingual = hp.n_vocab >= 51865
    eot = 50257 if multilingual else 50256
    beg = 50363 + ((hp.n_vocab - 51765 - 1 - 98) if multilingual else 0)
    n_prompt = 3 if multilingual else 1
    return eot, beg, n_prompt


_CTRL = 64   # natural style: channels [d - 64, d) carry EOT and the timestamp tokens (and nothing else); [d - 128, d - 64) is ballast no token reads


def natural_tensor(name: str, shape, hp: HParams, rng: np.random.Generator, logit_gain: float, ctx: dict):
    """`style="natural"` (write_model(**NATURAL)): synthetic weights whose decoder behaves like a transcriber under whisper.cpp's FULL rules --
    stays at temperature 0 (peaked next-token distribution, no loops: passes the logprob and entropy checks), emits increasing timestamp pairs and
    ends with EOT after a number of tokens that depends on the audio -- so that natural-EOT decoding (bench Mode N) and parity at data-dependent
    lengths have something meaningful to run on.  Returns None for tensors that keep the default random draw.  Construction:
      * the text state splits into T (text), a 64-channel ballast B and a 64-channel control subspace C.  Text-token embeddings live in T only,
        timestamp tokens and EOT in C only, nothing in B;
      * the next token is a well-mixed function of (current token, position, audio): block 0's MLP reads LN(E_tok + P_pos + cross-attention(audio))
        and its boosted, zero-mean output dominates the residual stream from there on -- no self-prediction through the tied embeddings and no
        slow-moving history average, the two things that make a random decoder loop.  The audio enters BEFORE that non-linearity (block 0's
        cross-attention, sharply peaked so that it returns the value rows of a few encoder positions, not their average), so it changes the map
        instead of adding the same bonus to the same tokens at every step;
      * every position's embedding carries a control vector of CONSTANT norm R in B + C (LayerNorm statistics, hence the scale of everything
        else, do not depend on the position): mostly ballast, plus a component along EOT growing linearly with the position (EOT wins around
        position eot_start + eot_ramp) and, at pair positions, along the embedding of THE timestamp that position should emit (beg + ts_rate x
        tokens so far: timestamps grow with the text, are never near ties, and close / open segments in pairs).  Block 0's LayerNorms are deaf
        to B + C (gain 0) and compensate the norm the control vector adds (gain x comp on T);
      * one text token in `eot_mean` is a "sentence ender": channel 0 of T is a flag that only those tokens' embeddings set, hidden unit 0 of
        block 0's MLP reads nothing but that flag and writes a large push along EOT, so the token after an ender is EOT with probability ~1.
        The stream is a well-mixed, audio-dependent walk over the vocabulary, so it meets an ender after a geometrically distributed number of
        steps: a different transcript length for every audio, no threshold to calibrate; the slow EOT ramp only guarantees an end.  (Tried first
        and dropped: a rank-1 term on the last cross-attention output -- a random encoder's cross-attention output has a large component
        common to all audios, large-v3 ended all 32 chunks after two tokens, r04_e -- and a random read-out of block 0's hidden units along EOT,
        whose firing rate differed by orders of magnitude between model seeds.)"""
    d = hp.n_text_state
    if d < 256:
        raise ValueError("natural style needs n_text_state >= 256")
    eot, beg, n_prompt = _special_ids(hp)
    T = d - 2 * _CTRL
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    last = f"decoder.blocks.{hp.n_text_layer - 1}."
    pos_scale, boost, rho = ctx.get("pos_scale", 0.4), ctx.get("mlp_boost", 4.0), ctx.get("ctrl_rho", 0.75)
    sx = 0.6 * boost * np.sqrt(T)                        # per-element std of block 0's MLP output = of the final residual stream on T
    R = rho * sx * np.sqrt(d)                            # norm of the control vector
    a_star = 4.3 * np.sqrt(T / d) / (rho * np.sqrt(d))   # alignment with EOT at which the EOT logit reaches the expected best text logit
    xT2 = T * (0.25 + pos_scale ** 2)                    # |E_tok + P_pos|^2 on T
    comp = float(np.sqrt((xT2 + R * R) / xT2))
    if name == "decoder.ln.weight":
        g = logit_gain * np.sqrt(T / d + rho * rho) / (0.5 * np.sqrt(T))     # text logits get std `logit_gain`
        return (g * (1.0 + 0.05 * rng.standard_normal(shape, dtype=np.float32))).astype(np.float32)
    if name in ("decoder.blocks.0.attn_ln.weight", "decoder.blocks.0.cross_attn_ln.weight", "decoder.blocks.0.mlp_ln.weight"):
        g = comp * (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32))
        g[T:] = 0.0
        return g.astype(np.float32)
    if name == "decoder.positional_embedding":
        pe = pos_scale * rng.standard_normal(shape, dtype=np.float32)
        pe[:, T:] = 0.0
        pe[:, 0] = 0.0                                   # channel 0 is the sentence-ender flag: only token embeddings write it
        # every control direction sums to zero over the channels: LayerNorm subtracts the channel mean, and a control vector of norm R with a
        # non-zero mean would shift every text channel by the same large amount (it switched the sentence-ender flag on for every token)
        cu, cn = rng.standard_normal(_CTRL), rng.standard_normal(_CTRL)
        u = np.zeros(d, np.float32); u[d - _CTRL:] = cu - cu.mean(); u /= np.linalg.norm(u)
        n = np.zeros(d, np.float32); n[T:d - _CTRL] = cn - cn.mean(); n /= np.linalg.norm(n)
        ctx.update(u_eot=u, n_bal=n, pe_plain=pe, R=R, a_star=a_star)
        return None                                      # finished by the token-embedding draw (the timestamp pushes point at timestamp embeddings)
    if name == "decoder.token_embedding.weight":
        te = 0.5 * rng.standard_normal(shape, dtype=np.float32)
        te[:, T:] = 0.0
        te[:, 0] = 0.0
        te *= (0.5 * np.sqrt(T)) / np.linalg.norm(te, axis=1, keepdims=True)   # equal norms: among 50 k candidates the longer rows would win again and again
        enders = np.nonzero(rng.random(eot) < 1.0 / ctx.get("eot_mean", 70.0))[0]
        enders = enders[enders > 255]                    # (byte tokens stay ordinary)
        te[enders, 0] = 2.0
        te[beg:] = 0.0
        ts_rows = 0.5 * np.sqrt(d / _CTRL) * rng.standard_normal((shape[0] - beg, _CTRL), dtype=np.float32)
        te[beg:, d - _CTRL:] = ts_rows - ts_rows.mean(axis=1, keepdims=True)
        te[beg:] -= np.outer(te[beg:] @ ctx["u_eot"], ctx["u_eot"])       # a timestamp push never moves the EOT logit
        te[eot] = 0.5 * np.sqrt(d) * ctx["u_eot"]
        pe, u, n = ctx.pop("pe_plain"), ctx["u_eot"], ctx["n_bal"]
        period, rate = int(ctx["ts_period"]), ctx.get("ts_rate", 25.0)
        for p in range(pe.shape[0]):
            s = np.zeros(d, np.float32)
            i = p - (n_prompt - 1)                       # index of the token sampled FROM position p
            if i >= 0 and (i == 0 or i % period in (period - 2, period - 1)):
                k = 0 if i == 0 else min(shape[0] - beg - 1, int(rate * (i // period + 1) * period))
                s += ctx.get("ts_align", 0.9) * te[beg + k] / np.linalg.norm(te[beg + k])
            else:
                s += min(0.95, max(0.0, a_star * (p - ctx["eot_start"]) / ctx.get("eot_ramp", 60.0))) * u
            pe[p] += R * (s + np.sqrt(max(0.0, 1.0 - float(s @ s))) * n)
        ctx["pe_final"] = pe.astype(np.float32)
        return te.astype(np.float32)
    if name == "decoder.blocks.0.mlp.2.weight":
        w = boost * np.sqrt(d) * rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in) * np.sqrt(T / d)
        w -= w.mean(axis=1, keepdims=True)               # the hidden units' common mean (GELU > 0 on average) maps to 0: no token is favoured at every step
        w[T:] = 0.0                                      # nothing is written into the ballast / control channels ...
        w[0] = 0.0                                       # ... nor into the flag channel ...
        w[:, 0] = 3.0 * a_star * R / 6.0 * ctx["u_eot"]  # ... except by hidden unit 0 (~6 after an ender, ~0 otherwise): a push of 3 x the EOT threshold
        return w.astype(np.float32)
    if name == "decoder.blocks.0.mlp.0.weight":
        w = rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)
        w[:, 0] = 0.0                                    # the flag is invisible to the successor map ...
        w[0] = 0.0
        w[0, 0] = 2.0                                    # ... and is all hidden unit 0 sees (z_0 ~ 3 for an ender)
        return w.astype(np.float32)
    if name in ("decoder.blocks.0.attn.out.weight", "decoder.blocks.0.cross_attn.out.weight"):
        gain = ctx.get("self_gain", 0.2) if name.endswith("attn.out.weight") and "cross" not in name else ctx.get("cross_gain", 0.2)
        w = gain * rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)
        w[0] = 0.0                                       # the attention outputs of block 0 leave the flag channel alone
        return w.astype(np.float32)
    if name == "decoder.blocks.0.cross_attn.query.weight":
        return (ctx.get("cross_q_gain", 8.0) * rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)
    return None


def synth_tensor(name: str, shape, hp: HParams, rng: np.random.Generator, logit_gain: float, ctx: dict) -> np.ndarray:
    """Seeded random weights with sane scales so activations stay O(1) through 32 layers, plus a little
    hand-built structure so that a random decoder behaves like a transcriber instead of a fixed point:

    * `logit_gain` = target std of the logits over the vocabulary (token embedding per-element std 0.5, final
      LayerNorm gain chosen to match).  Around 9 the greedy choice has a top-2 margin ~2 and p_top ~0.5, which
      keeps whisper.cpp's logprob/entropy fallback from firing on every window.
    * tied embeddings make a random decoder predict its own input; the last block's self-attention is built as
      an anti-repetition head (-c * mean of the history) and its MLP output is boosted to drown the direct path.
    * timestamp-token embeddings share a common direction, and the learned positional embedding pushes along it
      every `ts_period` positions (-> timestamp pairs) and along the EOT embedding late in the sequence
      (-> natural termination), so segments / seek / EOT logic is exercised.
    """
    d = hp.n_text_state
    eot, beg, n_prompt = _special_ids(hp)
    if name == "encoder.positional_embedding":
        return sinusoids(shape[0], shape[1])
    if ctx.get("style") == "natural":
        if name == "decoder.positional_embedding":
            natural_tensor(name, shape, hp, rng, logit_gain, ctx)
            return None
        t = natural_tensor(name, shape, hp, rng, logit_gain, ctx)
        if t is not None:
            return t
        if name.startswith("decoder.") and len(shape) == 2 and not name.endswith("embedding.weight"):   # no structured heads / boosts of the default style
            fan_in = int(np.prod(shape[1:]))
            return (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)
    if name == "decoder.ln.weight":
        g = logit_gain / (0.5 * np.sqrt(shape[0]))
        return (g * (1.0 + 0.05 * rng.standard_normal(shape, dtype=np.float32))).astype(np.float32)
    if name.endswith("_ln.weight") or name.endswith("ln_post.weight"):
        return (1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if name.endswith(".bias"):
        return (0.02 * rng.standard_normal(shape, dtype=np.float32)).astype(np.float32)
    if name == "decoder.positional_embedding":
        pe = rng.standard_normal(shape, dtype=np.float32)
        ctx["u_ts"] = rng.standard_normal(d).astype(np.float32)
        ctx["u_ts"] /= np.linalg.norm(ctx["u_ts"])
        ctx["u_eot"] = rng.standard_normal(d).astype(np.float32)
        ctx["u_eot"] /= np.linalg.norm(ctx["u_eot"])
        sx = 0.3 * np.sqrt(d)  # ~ per-element std of the final residual stream (boosted MLP)
        period = ctx["ts_period"]
        for p in range(shape[0]):
            k = (p - (n_prompt - 1)) % period
            if p >= n_prompt - 1 and k in (0, period - 1):
                pe[p] += 6.0 * sx * ctx["u_ts"]
            ramp = (p - ctx["eot_start"]) / 10.0
            if ramp > 0:
                pe[p] += min(ramp, 3.0) * 6.0 * sx * ctx["u_eot"]
        return pe.astype(np.float32)
    if name == "decoder.token_embedding.weight":
        te = 0.5 * rng.standard_normal(shape, dtype=np.float32)
        te[beg:] = 0.35 * rng.standard_normal((shape[0] - beg, d), dtype=np.float32) + 0.35 * np.sqrt(d) * ctx["u_ts"]
        te[eot] = 0.5 * np.sqrt(d) * ctx["u_eot"]
        return te.astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    last = f"decoder.blocks.{hp.n_text_layer - 1}."
    if name in (last + "attn.value.weight", last + "attn.out.weight"):
        eye = np.eye(shape[0], dtype=np.float32)
        return eye if name.endswith("value.weight") else (-4.0 * eye)
    if name == last + "mlp.2.weight":
        boost = np.sqrt(hp.n_text_state) / 2
        return (boost * rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)
    if name.endswith("cross_attn.out.weight") or name.endswith("cross_attn.query.weight"):
        return (3.0 * rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)  # make the audio matter
    return (rng.standard_normal(shape, dtype=np.float32) / np.sqrt(fan_in)).astype(np.float32)


def write_model(path: str, hp: HParams | str, seed: int = 0, logit_gain: float = 9.0, tensors: dict | None = None,
                ts_period: int = 12, eot_start: int = 40, ftype: int | str | None = None, vocab_overrides: dict | None = None, **style_kw):
    """Write a ggml legacy whisper model. `tensors` (name -> ndarray in PyTorch layout) overrides the
    synthetic draw -- used by the HF cross-check to export a transformers model's weights.
    `ftype` ("q5_0", "q5_1", "q8_0", "q4_0", "q4_1" or the ggml number): what whisper.cpp's quantize tool produces from the f16 file -- every
    2-D weight except the positional embeddings becomes block-quantised, the header's ftype says which type (+ 2000: GGML_QNT_VERSION 2).
    `vocab_overrides` (id -> bytes): replaces entries of the synthetic vocabulary, e.g. to plant whisper.cpp's non-speech symbols (weights are unaffected)."""
    if isinstance(hp, str):
        hp = PRESETS[hp]
    if isinstance(ftype, str):
        ftype = FTYPE_BY_NAME[ftype]
    qtype = FTYPE_TO_QTYPE.get(ftype) if ftype is not None else None
    if ftype is not None:
        hp = HParams(*(hp.astuple()[:10] + ((2000 + ftype) if qtype is not None else ftype,)))
    base_ftype = 1 if qtype is not None else hp.ftype
    spec_hp = HParams(*(hp.astuple()[:10] + (base_ftype,)))
    rng = np.random.default_rng(seed)
    ctx = {"ts_period": ts_period, "eot_start": eot_start}
    ctx.update(style_kw)    # style="natural" and its knobs: NATURAL / natural_tensor
    vocab = synth_vocab(hp.n_vocab)
    for i, t in (vocab_overrides or {}).items():
        vocab[i] = t
    filt = mel_filters(hp.n_mels)
    with open(path, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.astuple()))
        f.write(struct.pack("<2i", filt.shape[0], filt.shape[1]))
        f.write(filt.astype("<f4").tobytes())
        f.write(struct.pack("<i", len(vocab)))
        for t in vocab:
            f.write(struct.pack("<I", len(t)))
            f.write(t)
        specs = list(tensor_specs(spec_hp))
        drawn = {}
        if ctx.get("style") == "natural" and tensors is None:
            # the positional embedding is finished by the token-embedding draw: draw every tensor first, in file order (one seeded stream)
            for name, shape, ttype in specs:
                drawn[name] = synth_tensor(name, shape, hp, rng, logit_gain, ctx)
            drawn["decoder.positional_embedding"] = ctx["pe_final"]
        for name, shape, ttype in specs:
            if tensors is not None and name in tensors:
                data = np.asarray(tensors[name], dtype=np.float32).reshape(shape)
            elif name in drawn:
                data = drawn.pop(name)
            else:
                data = synth_tensor(name, shape, hp, rng, logit_gain, ctx)
            nb = name.encode()
            quant = qtype is not None and len(shape) == 2 and name.endswith("weight") and name not in QUANT_SKIP and shape[-1] % 32 == 0
            if quant:       # the tool quantises the f16 file's values
                data = data.astype(np.float16).astype(np.float32)
                ttype = qtype
            f.write(struct.pack("<3i", len(shape), len(nb), ttype))
            for i in range(len(shape)):
                f.write(struct.pack("<i", shape[len(shape) - 1 - i]))  # ggml order: ne[0] fastest
            f.write(nb)
            if quant:
                f.write(quantize_blocks(data, qtype))
            else:
                f.write(data.astype("<f2" if ttype == GGML_TYPE_F16 else "<f4").tobytes())
    return hp


def read_model(path: str):
    """Parse a ggml legacy whisper file -> (HParams, filters[n_mel,n_fft], vocab list[bytes], {name: float32 ndarray})."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    (magic,) = struct.unpack_from("<I", buf, off)
    off += 4
    if magic != GGML_MAGIC:
        raise ValueError("bad ggml magic")
    hp = HParams(*struct.unpack_from("<11i", buf, off))
    off += 44
    n_mel, n_fft = struct.unpack_from("<2i", buf, off)
    off += 8
    filt = np.frombuffer(buf, "<f4", n_mel * n_fft, off).reshape(n_mel, n_fft).copy()
    off += 4 * n_mel * n_fft
    (nv,) = struct.unpack_from("<i", buf, off)
    off += 4
    vocab = []
    for _ in range(nv):
        (ln,) = struct.unpack_from("<I", buf, off)
        off += 4
        vocab.append(bytes(buf[off : off + ln]))
        off += ln
    tensors = {}
    while off < len(buf):
        n_dims, nlen, ttype = struct.unpack_from("<3i", buf, off)
        off += 12
        ne = struct.unpack_from(f"<{n_dims}i", buf, off)
        off += 4 * n_dims
        name = bytes(buf[off : off + nlen]).decode()
        off += nlen
        n = int(np.prod(ne))
        if ttype == GGML_TYPE_F16:
            data = np.frombuffer(buf, "<f2", n, off).astype(np.float32)
            off += 2 * n
        elif ttype == GGML_TYPE_F32:
            data = np.frombuffer(buf, "<f4", n, off).copy()
            off += 4 * n
        elif ttype in Q_BLOCK_BYTES:
            data = dequantize_blocks(buf[off : off + n // 32 * Q_BLOCK_BYTES[ttype]], ttype, n)
            off += n // 32 * Q_BLOCK_BYTES[ttype]
        else:
            raise ValueError(f"unsupported ggml tensor type {ttype} for {name}")
        tensors[name] = data.reshape(tuple(reversed(ne)))
    return hp, filt, vocab, tensors
