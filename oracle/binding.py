"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY; never imported by speaksense_amd/)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MODE_F32, MODE_GGML_F16, MODE_BF16, MODE_FP8 = 0, 1, 2, 3   # FP8: GGML_F16 + e4m3 encoder-block / cross-KV projections (whisper_oracle.cpp header)
# Version-dependent whisper.cpp behaviour (the ledger); 0 = v1.5.0 .. v1.5.4, what whisper-rs-sys 0.9.0 vendors.  Same values as
# SS_COMPAT_* in include/speaksense.h.
COMPAT_RNG_STATE = 1         # <= v1.4.x: one std::mt19937(0) in whisper_state shared by all best_of decoders
COMPAT_OPENAI_TS_RULES = 2   # OpenAI's timestamp rules where whisper.cpp's differ (forced first timestamp, `<=` monotonic rule, <|0.00|> counts)
COMPAT_OPENAI_HISTORY = 4    # later windows of a call conditioned on the SEGMENTS' tokens (no closing timestamp of the last pair), <= 222 of them


class OrcOpts(C.Structure):
    _fields_ = [("mode", C.c_int32), ("gelu_erf", C.c_int32), ("n_threads", C.c_int32)]


class FullParams(C.Structure):
    # mirrors `struct FullParams` in whisper_oracle.cpp (the whisper_full_params fields whisper.rs:131-173 sets)
    _fields_ = [
        ("best_of", C.c_int32), ("temperature", C.c_float), ("temperature_inc", C.c_float), ("entropy_thold", C.c_float),
        ("logprob_thold", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float),
        ("no_context", C.c_int32), ("single_segment", C.c_int32), ("no_timestamps", C.c_int32), ("suppress_blank", C.c_int32),
        ("tdrz_enable", C.c_int32), ("print_special", C.c_int32), ("max_tokens", C.c_int32), ("n_max_text_ctx", C.c_int32),
        ("audio_ctx", C.c_int32), ("translate", C.c_int32), ("fixed_steps", C.c_int32), ("language", C.c_char * 8),
        ("offset_ms", C.c_int32), ("duration_ms", C.c_int32), ("detect_language", C.c_int32), ("prompt_n_tokens", C.c_int32),
        ("prompt_tokens", C.c_void_p), ("initial_prompt", C.c_char_p),
        ("token_timestamps", C.c_int32), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float),
        ("suppress_non_speech_tokens", C.c_int32), ("max_len", C.c_int32), ("split_on_word", C.c_int32),
    ]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "whisper_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_hparams.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_special_tokens.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_token_str.restype = C.c_char_p
        L.orc_token_str.argtypes = [C.c_void_p, C.c_int]
        L.orc_log_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(OrcOpts), C.c_void_p]
        L.orc_encode_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(OrcOpts), C.c_int, C.c_void_p]
        L.orc_state_new.restype = C.c_void_p
        L.orc_state_new.argtypes = [C.c_void_p, C.POINTER(OrcOpts)]
        L.orc_state_free.argtypes = [C.c_void_p]
        L.orc_state_set_compat.argtypes = [C.c_void_p, C.c_int]
        L.orc_state_rng_peek.restype = C.c_uint32
        L.orc_state_rng_peek.argtypes = [C.c_void_p, C.c_int]
        L.orc_state_set_encoder.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_state_set_encoder_ctx.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_state_cross_kv.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_process_logits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(FullParams), C.c_void_p, C.c_void_p]
        L.orc_full_default_params.argtypes = [C.POINTER(FullParams)]
        L.orc_full.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FullParams)]
        L.orc_full_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FullParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_full_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(FullParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_n_trace.argtypes = [C.c_void_p]
        L.orc_trace.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_n_sampled.argtypes = [C.c_void_p]
        L.orc_e4m3_round.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_encode_fp8_first_quant.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_e8m0_exponent.argtypes = [C.c_float]
        L.orc_quantize_rows_f8.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_lang_id.argtypes = [C.c_void_p]
        L.orc_lang_code_to_id.argtypes = [C.c_char_p]
        L.orc_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int]
        L.orc_sampled.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_n_segments.argtypes = [C.c_void_p]
        L.orc_segment_text.restype = C.c_char_p
        L.orc_segment_text.argtypes = [C.c_void_p, C.c_int]
        L.orc_segment_t0.restype = C.c_int64
        L.orc_segment_t0.argtypes = [C.c_void_p, C.c_int]
        L.orc_segment_t1.restype = C.c_int64
        L.orc_segment_t1.argtypes = [C.c_void_p, C.c_int]
        L.orc_segment_speaker_turn_next.argtypes = [C.c_void_p, C.c_int]
        L.orc_segment_n_tokens.argtypes = [C.c_void_p, C.c_int]
        L.orc_segment_tokens.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_signal_energy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_token_times_chunk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_float, C.c_float] + [C.c_void_p] * 3
        L.orc_voice_length.restype = C.c_float
        L.orc_voice_length.argtypes = [C.c_char_p]
        L.orc_n_tokens.argtypes = [C.c_void_p]
        L.orc_tokens.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_mel_of_state.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_time_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        # cap OpenMP: the GPU box reports 256 logical CPUs; hundreds of spinning threads on tiny decode-step loops
        # (or a cgroup quota below the CPU count) make the oracle orders of magnitude slower
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(default_threads())
        _LIB = L
    return _LIB


_THREADS_CAP = 16


def set_thread_cap(n: int):
    """Tests that run the oracle at full large-v3 depth raise the cap (the default 16 keeps hundreds of threads from spinning on tiny loops)."""
    global _THREADS_CAP
    _THREADS_CAP = max(1, int(n))
    if _LIB is not None:
        _LIB.orc_set_threads(default_threads())


def default_threads() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(_THREADS_CAP, n))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class OracleModel:
    def __init__(self, path: str):
        self.L = lib()
        self.h = self.L.orc_load(path.encode())
        if not self.h:
            raise RuntimeError(f"oracle: cannot load {path}")
        hp = np.zeros(11, np.int32)
        self.L.orc_hparams(self.h, _p(hp))
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer, self.n_text_ctx,
         self.n_text_state, self.n_text_head, self.n_text_layer, self.n_mels, self.ftype) = [int(x) for x in hp]
        st = np.zeros(9, np.int32)
        self.L.orc_special_tokens(self.h, _p(st))
        (self.eot, self.sot, self.translate, self.transcribe, self.solm, self.prev, self.nosp, self.not_, self.beg) = [int(x) for x in st]

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def token_str(self, i: int) -> bytes:
        return self.L.orc_token_str(self.h, i)

    def tokenize(self, text) -> list:
        b = text.encode("utf-8") if isinstance(text, str) else text
        ids = np.zeros(max(16, 2 * len(b)), np.int32)
        n = self.L.orc_tokenize(self.h, b, _p(ids), len(ids))
        assert n >= 0
        return [int(x) for x in ids[:n]]

    def token_times_chunk(self, pcm, segments, token_lists, thold_pt=0.01, thold_ptsum=0.01):
        """Token-level timestamps of one chunk from GIVEN token data: `segments` = [(t0, t1)], `token_lists` = per segment dict(ids, tid, pt, ptsum)
        (e.g. speaksense_amd.binding.Session.token_times()).  Returns per segment dict(t0, t1, vlen)."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        st0 = np.array([s[0] for s in segments], np.int64); st1 = np.array([s[1] for s in segments], np.int64)
        nt = np.array([len(t["ids"]) for t in token_lists], np.int32)
        cat = lambda k, dt: np.ascontiguousarray(np.concatenate([np.asarray(t[k], dt) for t in token_lists]) if token_lists else np.zeros(0, dt), dt)
        ids, tid, pt, ps = cat("ids", np.int32), cat("tid", np.int32), cat("pt", np.float32), cat("ptsum", np.float32)
        n = int(nt.sum())
        o0, o1, ov = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.float32)
        self.L.orc_token_times_chunk(self.h, _p(pcm), len(pcm), len(segments), _p(st0), _p(st1), _p(nt), _p(ids), _p(tid), _p(pt), _p(ps),
                                     thold_pt, thold_ptsum, _p(o0), _p(o1), _p(ov))
        out, at = [], 0
        for k in nt:
            out.append(dict(t0=o0[at:at + k].copy(), t1=o1[at:at + k].copy(), vlen=ov[at:at + k].copy())); at += k
        return out

    def log_mel(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, np.float32)
        n_len = self.L.orc_mel_n_len(len(pcm))
        out = np.empty((self.n_mels, n_len), np.float32)
        self.L.orc_log_mel(self.h, _p(pcm), len(pcm), _p(out), n_len)
        return out

    def encode(self, mel: np.ndarray, seek: int = 0, mode: int = MODE_F32, gelu_erf: int = 0, audio_ctx: int = 0) -> np.ndarray:
        """audio_ctx > 0 (whisper_full_params.audio_ctx): only the first audio_ctx positions, [audio_ctx][n_audio_state]."""
        mel = np.ascontiguousarray(mel, np.float32)
        o = OrcOpts(mode, gelu_erf, default_threads())
        if audio_ctx > 0:
            out = np.empty((audio_ctx, self.n_audio_state), np.float32)
            self.L.orc_encode_ctx(self.h, _p(mel), mel.shape[1], seek, C.byref(o), int(audio_ctx), _p(out))
            return out
        out = np.empty((self.n_audio_ctx, self.n_audio_state), np.float32)
        self.L.orc_encode(self.h, _p(mel), mel.shape[1], seek, C.byref(o), _p(out))
        return out

    def encode_fp8_first_quant(self, mel: np.ndarray, seek: int = 0) -> np.ndarray:
        """FP8 mode: LayerNorm 1 of encoder block 0 as the e4m3 projections see it (code x 2^s per element), [n_audio_ctx][n_audio_state]."""
        mel = np.ascontiguousarray(mel, np.float32)
        out = np.empty((self.n_audio_ctx, self.n_audio_state), np.float32)
        if self.L.orc_encode_fp8_first_quant(self.h, _p(mel), mel.shape[1], seek, default_threads(), _p(out)) != 0:
            raise RuntimeError("oracle: the FP8 encoder made no e4m3 projection")
        return out

    def time_sample(self, pcm, mode, n_enc_layers, n_cross_layers, n_dec_steps, n_threads):
        pcm = np.ascontiguousarray(pcm, np.float32)
        out = np.zeros(5, np.float64)
        self.L.orc_time_sample(self.h, _p(pcm), len(pcm), mode, n_enc_layers, n_cross_layers, n_dec_steps, n_threads, _p(out))
        return dict(mel_s=out[0], stem_s=out[1], enc_layer_s=out[2], cross_layer_s=out[3], dec_step_s=out[4])

    def new_state(self, mode: int = MODE_F32, gelu_erf: int = 0, compat: int = 0) -> "OracleState":
        return OracleState(self, mode, gelu_erf, compat)


def default_params(**kw) -> FullParams:
    """whisper_full_default_params(GREEDY) + the reference's build_params (whisper.rs:131-173) + stream-mode
    overrides (whisper.rs:65-69): greedy best_of 5, temperature 0 (+0.2 ladder), no_context, max_initial_ts 1.0."""
    p = FullParams()
    lib().orc_full_default_params(C.byref(p))
    p.no_context = 1
    for k, v in kw.items():
        if k in ("language", "initial_prompt"):
            v = v.encode() if isinstance(v, str) else v
        if k == "prompt_tokens":
            arr = np.ascontiguousarray(v, np.int32)
            p._keep = arr                      # the struct holds a raw pointer
            p.prompt_tokens = arr.ctypes.data
            p.prompt_n_tokens = len(arr)
            continue
        setattr(p, k, v)
    return p


class OracleState:
    def __init__(self, model: OracleModel, mode: int, gelu_erf: int, compat: int = 0):
        self.m = model
        self.L = model.L
        o = OrcOpts(mode, gelu_erf, default_threads())
        self.h = self.L.orc_state_new(model.h, C.byref(o))
        self.compat = int(compat)
        self.L.orc_state_set_compat(self.h, self.compat)

    def rng_peek(self, decoder: int = 0) -> int:
        """Next output of the generator decoder `decoder` would draw from (a copy is advanced, not the generator): two states whose samplers
        consumed the same number of draws agree on it."""
        return int(self.L.orc_state_rng_peek(self.h, decoder))

    def close(self):
        if self.h:
            self.L.orc_state_free(self.h)
            self.h = None

    def set_encoder(self, enc: np.ndarray):
        """enc: [n_audio_ctx][n_audio_state], or fewer rows = the output of a shortened context (whisper_full_params.audio_ctx)."""
        enc = np.ascontiguousarray(enc, np.float32)
        if enc.shape[0] < self.m.n_audio_ctx:
            if self.L.orc_state_set_encoder_ctx(self.h, _p(enc), int(enc.shape[0])) != 0:
                raise RuntimeError("oracle: bad audio_ctx")
            return
        self.L.orc_state_set_encoder(self.h, _p(enc))

    def cross_kv(self, il: int):
        k = np.empty((self.m.n_audio_ctx, self.m.n_text_state), np.float32)
        v = np.empty_like(k)
        self.L.orc_state_cross_kv(self.h, il, _p(k), _p(v))
        return k, v

    def decode(self, tokens, n_past: int) -> np.ndarray:
        t = np.ascontiguousarray(tokens, np.int32)
        out = np.empty(self.m.n_vocab, np.float32)
        self.L.orc_decode(self.h, _p(t), len(t), n_past, _p(out))
        return out

    def process_logits(self, raw: np.ndarray, hist, has_ts: bool, seek_delta: int, params: FullParams):
        raw = np.ascontiguousarray(raw, np.float32)
        h = np.ascontiguousarray(hist, np.int32)
        lp = np.empty(self.m.n_vocab, np.float32)
        o5 = np.zeros(5, np.float32)
        tid = self.L.orc_process_logits(self.h, _p(raw), _p(h), len(h), int(has_ts), seek_delta, C.byref(params), _p(lp), _p(o5))
        return tid, lp, o5

    def full(self, pcm: np.ndarray, params: FullParams | None = None, forced=None, trace=None):
        """`forced` (test hook): token ids of another implementation; greedy step g takes forced[g] instead of the argmax and the result
        carries `forced_gap[g]` = logprob(oracle's best) - logprob(forced[g]) and `forced_best[g]` = the oracle's own pick at that step.
        `trace` (test hook): EVERY id the other implementation sampled, in whisper_sample_token call order (failed attempts and losing best_of
        decoders included); sampled (t > 0) calls are replayed too: `trace_kind[g]` 0 = greedy (gap as above), 1 = sampled, `trace_gap[g]` then
        is the distance of the uniform the oracle drew (same generator, same position) to trace[g]'s interval of the oracle's cumulative distribution,
        and `trace_sens[g]` = F (1 - F) / T at the interval boundary nearer to the uniform: a logit perturbation of +-delta moves that boundary by
        at most 2 delta trace_sens[g] (first order)."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        params = params or default_params()
        extra = {}
        if trace is not None:
            ids = np.ascontiguousarray(trace, np.int32)
            gaps = np.zeros(max(1, len(ids)), np.float32)
            best = np.zeros(max(1, len(ids)), np.int32)
            kind = np.zeros(max(1, len(ids)), np.int32)
            sens = np.zeros(max(1, len(ids)), np.float32)
            used = np.zeros(1, np.int32)
            rc = self.L.orc_full_trace(self.h, _p(pcm), len(pcm), C.byref(params), _p(ids), len(ids), _p(gaps), _p(best), _p(kind), _p(sens), _p(used))
            k = int(used[0])
            extra = dict(trace_gap=gaps[:k].copy(), trace_best=best[:k].copy(), trace_kind=kind[:k].copy(), trace_sens=sens[:k].copy())
        elif forced is None:
            rc = self.L.orc_full(self.h, _p(pcm), len(pcm), C.byref(params))
        else:
            ids = np.ascontiguousarray(forced, np.int32)
            gaps = np.zeros(max(1, len(ids)), np.float32)
            best = np.zeros(max(1, len(ids)), np.int32)
            used = np.zeros(1, np.int32)
            rc = self.L.orc_full_forced(self.h, _p(pcm), len(pcm), C.byref(params), _p(ids), len(ids), _p(gaps), _p(best), _p(used))
            extra = dict(forced_gap=gaps[: int(used[0])].copy(), forced_best=best[: int(used[0])].copy())
        if rc != 0:
            raise RuntimeError(f"orc_full -> {rc}")
        segs = []
        for i in range(self.L.orc_n_segments(self.h)):
            segs.append(dict(text=self.L.orc_segment_text(self.h, i), t0=self.L.orc_segment_t0(self.h, i),
                             t1=self.L.orc_segment_t1(self.h, i), speaker_turn_next=bool(self.L.orc_segment_speaker_turn_next(self.h, i))))
            k = self.L.orc_segment_n_tokens(self.h, i)
            tid, tt0, tt1, vl = np.zeros(k, np.int32), np.zeros(k, np.int64), np.zeros(k, np.int64), np.zeros(k, np.float32)
            if k:
                self.L.orc_segment_tokens(self.h, i, _p(tid), _p(tt0), _p(tt1), _p(vl))
            # whisper_full_get_token_data per segment: ids, token-level t0 / t1 (10 ms units, -1 = not computed), voice length
            segs[-1]["token_times"] = dict(ids=tid, t0=tt0, t1=tt1, vlen=vl)
        n = self.L.orc_n_tokens(self.h)
        ids = np.zeros(n, np.int32)
        plog = np.zeros(n, np.float32)
        if n:
            self.L.orc_tokens(self.h, _p(ids), _p(plog))
        ns = self.L.orc_n_sampled(self.h)
        sampled = np.zeros(ns, np.int32)
        if ns:
            self.L.orc_sampled(self.h, _p(sampled))
        extra["sampled"] = sampled
        nt = self.L.orc_n_trace(self.h)
        tr = np.zeros(nt, np.int32)
        if nt:
            self.L.orc_trace(self.h, _p(tr))
        extra["trace"] = tr
        extra["lang_id"] = int(self.L.orc_lang_id(self.h))
        c = np.zeros(3, np.int32)
        self.L.orc_counters(self.h, _p(c))
        return dict(segments=segs, tokens=ids, plog=plog, n_encode=int(c[0]), n_decode=int(c[1]), n_fail=int(c[2]), **extra)


def signal_energy(pcm: np.ndarray, hw: int = 32) -> np.ndarray:
    """whisper.cpp get_signal_energy: mean |x| over a centred window of 2 hw + 1 samples (token-level timestamps)."""
    pcm = np.ascontiguousarray(pcm, np.float32)
    out = np.zeros(len(pcm), np.float32)
    if len(pcm):
        lib().orc_signal_energy(_p(pcm), len(pcm), hw, _p(out))
    return out


def voice_length(text: bytes) -> float:
    return float(lib().orc_voice_length(text))


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """FP8 mode primitive: nearest OCP e4m3 value (ties to even code, saturating at 448)."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().orc_e4m3_round(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), x.size)
    return out


def e8m0_exponent(amax: float) -> int:
    return int(lib().orc_e8m0_exponent(C.c_float(amax)))


def quantize_rows_f8(a: np.ndarray, f16_first: bool = False) -> np.ndarray:
    """FP8 mode primitive: per (row, 64-column block) power-of-two scale + e4m3, returned dequantised."""
    a = np.ascontiguousarray(a, np.float32)
    out = np.empty_like(a)
    lib().orc_quantize_rows_f8(a.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), a.shape[0], a.shape[1], int(f16_first))
    return out
