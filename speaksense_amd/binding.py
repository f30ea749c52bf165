"""ctypes binding of libspeaksense_hip.so (include/speaksense.h).  Product path: raises if the HIP library is
missing or no GPU is visible -- there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS_LIB_PATH") or os.path.join(_HERE, "libspeaksense_hip.so")   # SS_LIB_PATH: A/B of two builds on one box (tools/experiments)
_LIB = None

DTYPE_BF16, DTYPE_F16, DTYPE_FP8 = 0, 1, 2   # FP8: the f16 engine with e4m3 encoder / cross-KV projections
# SS_COMPAT_* (include/speaksense.h): which variant of a whisper.cpp-version-dependent behaviour the engine reproduces; 0 = whisper.cpp v1.5.x
COMPAT_RNG_STATE, COMPAT_OPENAI_TS_RULES, COMPAT_OPENAI_HISTORY = 1, 2, 4


class EngineOpts(C.Structure):
    _fields_ = [("device", C.c_int32), ("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_decoders", C.c_int32),
                ("batch_wait_us", C.c_int32), ("n_lanes", C.c_int32), ("compat", C.c_int32), ("reserved", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("best_of", C.c_int32), ("temperature", C.c_float), ("temperature_inc", C.c_float), ("entropy_thold", C.c_float),
                ("logprob_thold", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float), ("no_context", C.c_int32),
                ("single_segment", C.c_int32), ("no_timestamps", C.c_int32), ("suppress_blank", C.c_int32), ("tdrz_enable", C.c_int32),
                ("print_special", C.c_int32), ("max_tokens", C.c_int32), ("audio_ctx", C.c_int32), ("translate", C.c_int32),
                ("fixed_steps", C.c_int32), ("language", C.c_char * 8),
                ("n_max_text_ctx", C.c_int32), ("offset_ms", C.c_int32), ("duration_ms", C.c_int32), ("detect_language", C.c_int32),
                ("prompt_tokens", C.c_void_p), ("prompt_n_tokens", C.c_int32), ("token_timestamps", C.c_int32), ("initial_prompt", C.c_char_p),
                ("thold_pt", C.c_float), ("thold_ptsum", C.c_float),
                ("suppress_non_speech_tokens", C.c_int32), ("max_len", C.c_int32), ("split_on_word", C.c_int32)]


ABI_VERSION = 6   # include/speaksense.h SS_ABI_VERSION


class DenoiseConfig(C.Structure):   # DenoiseConfig, /root/reference/src/audio/mod.rs:41-61
    _fields_ = [("frame_size", C.c_int32), ("overlap", C.c_float), ("strength", C.c_float), ("noise_gate", C.c_float),
                ("enable_noise_reduction", C.c_int32), ("threshold", C.c_float)]


class SpeakSenseError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"speaksense error {code}: {msg}")
        self.code = code


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950)")
        L = C.CDLL(LIB_PATH)
        vp, i32, f32p = C.c_void_p, C.c_int32, C.c_void_p
        # the library copies ss_params / ss_engine_opts by value: a layout this binding does not share would be read out of bounds (speaksense.h SS_ABI_VERSION)
        if (L.ss_abi_version(), L.ss_sizeof_params(), L.ss_sizeof_engine_opts()) != (ABI_VERSION, C.sizeof(Params), C.sizeof(EngineOpts)):
            raise ImportError(f"{LIB_PATH}: ABI {L.ss_abi_version()} with ss_params of {L.ss_sizeof_params()} / ss_engine_opts of {L.ss_sizeof_engine_opts()} bytes; "
                              f"this binding was written for ABI {ABI_VERSION}, {C.sizeof(Params)} / {C.sizeof(EngineOpts)} bytes -- rebuild the library")
        L.ss_last_error.restype = C.c_char_p
        L.ss_default_params.argtypes = [C.POINTER(Params)]
        L.ss_engine_create.argtypes = [C.c_char_p, C.POINTER(EngineOpts), C.POINTER(vp)]
        L.ss_engine_free.argtypes = [vp]
        L.ss_engine_hparams.argtypes = [vp, vp]
        L.ss_engine_special_tokens.argtypes = [vp, vp]
        L.ss_engine_token_str.restype = C.c_char_p
        L.ss_engine_token_str.argtypes = [vp, i32]
        L.ss_session_create.restype = vp
        L.ss_session_create.argtypes = [vp]
        L.ss_session_free.argtypes = [vp]
        L.ss_transcribe_batch.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), i32, C.POINTER(Params), i32]
        L.ss_transcribe.argtypes = [vp, f32p, i32, C.POINTER(Params)]
        L.ss_submit.argtypes = [vp, f32p, i32, C.POINTER(Params), C.POINTER(vp)]
        L.ss_wait.argtypes = [vp]
        L.ss_ticket_ready.argtypes = [vp]
        L.ss_result_n_segments.argtypes = [vp]
        L.ss_result_segment_text.restype = C.c_char_p
        L.ss_result_segment_text.argtypes = [vp, i32]
        L.ss_result_segment_t0.restype = C.c_int64
        L.ss_result_segment_t0.argtypes = [vp, i32]
        L.ss_result_segment_t1.restype = C.c_int64
        L.ss_result_segment_t1.argtypes = [vp, i32]
        L.ss_result_segment_speaker_turn_next.argtypes = [vp, i32]
        L.ss_result_segment_n_tokens.argtypes = [vp, i32]
        L.ss_result_segment_token.argtypes = [vp, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.ss_result_segment_token_times.argtypes = [vp, i32, i32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_float)]
        L.ss_result_n_tokens.argtypes = [vp]
        L.ss_result_tokens.argtypes = [vp, vp, vp]
        L.ss_result_n_sampled_tokens.argtypes = [vp]
        L.ss_result_sampled_tokens.argtypes = [vp, vp]
        L.ss_result_n_trace_tokens.argtypes = [vp]
        L.ss_result_trace_tokens.argtypes = [vp, vp]
        L.ss_result_counters.argtypes = [vp, vp]
        L.ss_result_lang_id.argtypes = [vp]
        L.ss_engine_tokenize.argtypes = [vp, C.c_char_p, vp, i32]
        L.ss_model_tokenize.argtypes = [C.c_char_p, C.c_char_p, vp, i32]
        L.ss_session_rng_draws.argtypes = [vp]
        L.ss_session_rng_draws.restype = C.c_int64
        L.ss_session_rng_discard.argtypes = [vp, C.c_int64]
        L.ss_engine_lane_counters.argtypes = [vp, i32, vp]
        L.ss_engine_mem_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.ss_session_rng_draws_decoder.argtypes = [vp, i32]
        L.ss_session_rng_draws_decoder.restype = C.c_int64
        L.ss_mel_n_len.argtypes = [i32]
        L.ss_log_mel.argtypes = [vp, f32p, i32, f32p, i32]
        L.ss_signal_energy.argtypes = [vp, f32p, i32, f32p]
        L.ss_encode.argtypes = [vp, f32p, i32, i32, f32p]
        L.ss_encode_ctx.argtypes = [vp, f32p, i32, i32, i32, f32p]
        L.ss_session_set_encoder.argtypes = [vp, f32p]
        L.ss_session_set_encoder_ctx.argtypes = [vp, f32p, i32]
        L.ss_session_decode.argtypes = [vp, vp, i32, i32, f32p]
        L.ss_engine_set_encoder_window.argtypes = [vp, i32, f32p]
        L.ss_engine_fp8_first_quant.argtypes = [vp, f32p, i32, i32, C.c_void_p, C.c_void_p]
        L.ss_engine_decode_rows.argtypes = [vp, vp, vp, vp, vp, i32, vp, i32, f32p]
        L.ss_process_logits.argtypes = [vp, f32p, vp, i32, i32, i32, C.POINTER(Params), f32p]
        L.ss_process_logits_row.argtypes = [vp, f32p, vp, i32, i32, i32, C.POINTER(Params), f32p, f32p]
        L.ss_default_denoise_config.argtypes = [C.POINTER(DenoiseConfig)]
        L.ss_denoise_audio.argtypes = [vp, f32p, i32, C.POINTER(DenoiseConfig), i32, f32p, C.POINTER(i32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ss_resample_max_out.argtypes = [C.c_int64, i32]
        L.ss_resample_max_out.restype = C.c_int64
        L.ss_resample_stream.argtypes = [vp, f32p, C.c_int64, i32, f32p, C.c_int64, C.POINTER(C.c_int64), vp, C.POINTER(C.c_float)]
        L.ss_preprocess_n_out.argtypes = [C.c_int64]
        L.ss_preprocess_n_out.restype = C.c_int64
        L.ss_preprocess_stream.argtypes = [vp, f32p, C.c_int64, vp, i32, i32, C.POINTER(DenoiseConfig), f32p, f32p, C.POINTER(C.c_float)]
        L.ss_engine_last_timing.argtypes = [vp, f32p]
        L.ss_engine_last_counters.argtypes = [vp, vp]
        L.ss_engine_totals.argtypes = [vp, vp, vp, C.POINTER(i32)]
        L.ss_pool_create.argtypes = [C.c_char_p, vp, i32, C.POINTER(EngineOpts), C.POINTER(vp)]
        L.ss_pool_free.argtypes = [vp]
        L.ss_pool_n_engines.argtypes = [vp]
        L.ss_pool_engine.restype = vp
        L.ss_pool_engine.argtypes = [vp, i32]
        L.ss_pool_session_create.restype = vp
        L.ss_pool_session_create.argtypes = [vp]
        L.ss_pool_submit.argtypes = [vp, vp, f32p, i32, C.POINTER(Params), C.POINTER(vp)]
        L.ss_pool_last_engine.argtypes = [vp]
        L.ss_pool_pick.argtypes = [vp, i32, C.c_uint32]
        L.ss_submit_ex.argtypes = [vp, f32p, i32, C.POINTER(Params), i32, C.POINTER(vp)]
        L.ss_engine_probe_gemm.argtypes = [vp, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_double)]
        L.ss_engine_selftest_gemm.argtypes = [vp, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ss_e4m3_from_f32.argtypes = [vp, vp, C.c_int64]
        L.ss_engine_selftest_gemm_ex.argtypes = [vp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise SpeakSenseError(rc, lib().ss_last_error().decode(errors="replace"))


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw) -> Params:
    p = Params()
    lib().ss_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("language", "initial_prompt") and isinstance(v, str):
            v = v.encode()
        if k == "prompt_tokens":
            arr = np.ascontiguousarray(v, np.int32)
            p._keep = arr                      # the struct holds a raw pointer; the library copies the tokens during the call
            p.prompt_tokens = arr.ctypes.data
            p.prompt_n_tokens = len(arr)
            continue
        setattr(p, k, v)
    return p


_LANGS = ("en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg lt la mi ml cy sk te fa lv bn sr az "
          "sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt "
          "haw ln ha ba jw su yue").split()


def lang_code(lang_id: int) -> str:
    """whisper_lang_str"""
    return _LANGS[lang_id]


def model_tokenize(model_path: str, text) -> list:
    """whisper_tokenize with only the file's vocabulary loaded (host only, no GPU)."""
    b = text.encode("utf-8") if isinstance(text, str) else text
    ids = np.zeros(max(16, 2 * len(b)), np.int32)
    n = lib().ss_model_tokenize(model_path.encode(), b, _p(ids), len(ids))
    if n < 0:
        raise SpeakSenseError(n, lib().ss_last_error().decode(errors="replace"))
    return [int(x) for x in ids[:n]]


class Engine:
    def __init__(self, model_path: str, device: int = 0, dtype: int = DTYPE_F16, max_batch: int = 8, max_decoders: int = 5,
                 batch_wait_us: int = 2000, n_lanes: int = 0, compat: int = 0):
        self.L = lib()
        o = EngineOpts(device, dtype, max_batch, max_decoders, batch_wait_us, n_lanes, compat)
        self.compat = compat
        h = C.c_void_p()
        self.model_path = model_path
        _check(self.L.ss_engine_create(model_path.encode(), C.byref(o), C.byref(h)))
        self.h = h
        hp = np.zeros(11, np.int32)
        self.L.ss_engine_hparams(self.h, _p(hp))
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer, self.n_text_ctx, self.n_text_state,
         self.n_text_head, self.n_text_layer, self.n_mels, self.ftype) = [int(x) for x in hp]
        st = np.zeros(9, np.int32)
        self.L.ss_engine_special_tokens(self.h, _p(st))
        (self.eot, self.sot, self.translate, self.transcribe, self.solm, self.prev, self.nosp, self.not_, self.beg) = [int(x) for x in st]
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "h", None):
            self.L.ss_engine_free(self.h)
            self.h = None

    def lane_counters(self, lane: int):
        cnt = np.zeros(6, np.int64)
        _check(self.L.ss_engine_lane_counters(self.h, lane, _p(cnt)))
        return dict(decoder_passes=int(cnt[0]), decoder_rows=int(cnt[1]), encoder_windows=int(cnt[2]), admitted=int(cnt[3]), started_midway=int(cnt[4]),
                    graph_evictions=int(cnt[5]))

    def mem_info(self):
        """(free, total) bytes of the engine's device (hipMemGetInfo)."""
        f, t = C.c_int64(), C.c_int64()
        _check(self.L.ss_engine_mem_info(self.h, C.byref(f), C.byref(t)))
        return int(f.value), int(t.value)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def token_str(self, i: int) -> bytes:
        return self.L.ss_engine_token_str(self.h, i)

    def tokenize(self, text) -> list:
        b = text.encode("utf-8") if isinstance(text, str) else text
        ids = np.zeros(max(16, 2 * len(b)), np.int32)
        n = self.L.ss_engine_tokenize(self.h, b, _p(ids), len(ids))
        if n < 0:
            raise SpeakSenseError(n, lib().ss_last_error().decode(errors="replace"))
        return [int(x) for x in ids[:n]]

    def new_session(self) -> "Session":
        return Session(self)

    # ---- stage hooks ----
    def signal_energy(self, pcm: np.ndarray) -> np.ndarray:
        """whisper.cpp get_signal_energy(pcm, n, 32) on the device (token-level timestamps)."""
        pcm = np.ascontiguousarray(pcm, np.float32)
        out = np.zeros(len(pcm), np.float32)
        _check(self.L.ss_signal_energy(self.h, _p(pcm), len(pcm), _p(out)))
        return out

    def log_mel(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, np.float32)
        n_len = self.L.ss_mel_n_len(len(pcm))
        out = np.empty((self.n_mels, n_len), np.float32)
        _check(self.L.ss_log_mel(self.h, _p(pcm), len(pcm), _p(out), n_len))
        return out

    def encode(self, mel: np.ndarray, seek: int = 0, audio_ctx: int = 0) -> np.ndarray:
        """audio_ctx > 0 (Params.audio_ctx): the first audio_ctx positions only, [audio_ctx][n_audio_state]."""
        mel = np.ascontiguousarray(mel, np.float32)
        if audio_ctx:
            out = np.empty((audio_ctx, self.n_audio_state), np.float32)
            _check(self.L.ss_encode_ctx(self.h, _p(mel), mel.shape[1], seek, int(audio_ctx), _p(out)))
            return out
        out = np.empty((self.n_audio_ctx, self.n_audio_state), np.float32)
        _check(self.L.ss_encode(self.h, _p(mel), mel.shape[1], seek, _p(out)))
        return out

    def set_encoder_window(self, window: int, enc: np.ndarray):
        """Stage hook: fill cross-KV cache slot `window` (< max_batch) from an encoder output [n_audio_ctx][n_audio_state]."""
        enc = np.ascontiguousarray(enc, np.float32)
        assert enc.shape == (self.n_audio_ctx, self.n_audio_state)
        _check(self.L.ss_engine_set_encoder_window(self.h, int(window), _p(enc)))

    def fp8_first_quant(self, mel: np.ndarray, seek: int = 0):
        """fp8 engines: (codes uint8 [n_audio_ctx][n_audio_state], exponent bytes uint8 [n_audio_ctx][n_audio_state / 64]) at the first quantisation
        point (LayerNorm 1 of encoder block 0); value = e4m3(code) * 2^(exp - 127)."""
        mel = np.ascontiguousarray(mel, np.float32)
        codes = np.empty((self.n_audio_ctx, self.n_audio_state), np.uint8)
        exps = np.empty((self.n_audio_ctx, self.n_audio_state // 64), np.uint8)
        _check(self.L.ss_engine_fp8_first_quant(self.h, _p(mel), mel.shape[1], int(seek), codes.ctypes.data_as(C.c_void_p), exps.ctypes.data_as(C.c_void_p)))
        return codes, exps

    def decode_rows(self, token, pos, slot, cross, sample_rows) -> np.ndarray:
        """Stage hook: ONE decoder pass over len(token) rows (row i = token[i] at position pos[i] of self-KV slot slot[i], attending to
        cross-KV window cross[i]) -> raw logits [len(sample_rows)][n_vocab] of the listed rows."""
        t, p, sl, cr, sr = (np.ascontiguousarray(a, np.int32) for a in (token, pos, slot, cross, sample_rows))
        assert len(t) == len(p) == len(sl) == len(cr)
        out = np.empty((len(sr), self.n_vocab), np.float32)
        _check(self.L.ss_engine_decode_rows(self.h, _p(t), _p(p), _p(sl), _p(cr), len(t), _p(sr), len(sr), _p(out)))
        return out

    def process_logits(self, raw, hist, has_ts: bool, seek_delta: int, params: Params | None = None, want_row: bool = False):
        """want_row: also `logprobs` = the processed log-softmax row [n_vocab], -inf where a rule masks the id (ss_process_logits_row)."""
        raw = np.ascontiguousarray(raw, np.float32)
        h = np.ascontiguousarray(hist, np.int32)
        out = np.zeros(6, np.float32)
        pp = C.byref(params) if params is not None else None
        row = None
        if want_row:
            row = np.empty(self.n_vocab, np.float32)
            _check(self.L.ss_process_logits_row(self.h, _p(raw), _p(h), len(h), int(has_ts), seek_delta, pp, _p(out), _p(row)))
        else:
            _check(self.L.ss_process_logits(self.h, _p(raw), _p(h), len(h), int(has_ts), seek_delta, pp, _p(out)))
        r = dict(id=int(out[0]), p=float(out[1]), plog=float(out[2]), tid=int(out[3]), pt=float(out[4]), ptsum=float(out[5]))
        if want_row:
            r["logprobs"] = row
        return r

    def transcribe_batch(self, sessions, pcms, params: Params | None = None, device_ptrs=None):
        """pcms: list of np.float32 arrays (host), or with device_ptrs=[(ptr, n), ...] device buffers already in HBM."""
        n = len(sessions)
        sh = (C.c_void_p * n)(*[s.h for s in sessions])
        if device_ptrs is not None:
            pp = (C.c_void_p * n)(*[int(p) for p, _ in device_ptrs])
            nn = (C.c_int32 * n)(*[int(k) for _, k in device_ptrs])
            on_dev = 1
        else:
            pcms = [np.ascontiguousarray(p, np.float32) for p in pcms]
            pp = (C.c_void_p * n)(*[p.ctypes.data for p in pcms])
            nn = (C.c_int32 * n)(*[len(p) for p in pcms])
            on_dev = 0
        _check(self.L.ss_transcribe_batch(self.h, sh, pp, nn, n, C.byref(params) if params is not None else None, on_dev))
        return [s.result() for s in sessions]

    def denoise_audio(self, samples, config: "DenoiseConfig | None" = None, force_type: int = -1):
        """`denoise_audio(samples, &config)` (src/audio/mod.rs:507) -> (out, noise_type, normalized_variance, device_ms)."""
        x = np.ascontiguousarray(samples, np.float32)
        out = np.empty_like(x)
        nt, nv, ms = C.c_int32(), C.c_float(), C.c_float()
        _check(self.L.ss_denoise_audio(self.h, _p(x), len(x), C.byref(config) if config is not None else None, force_type, _p(out),
                                       C.byref(nt), C.byref(nv), C.byref(ms)))
        return out, nt.value, nv.value, ms.value

    def resample_stream(self, mono, from_rate: int):
        """rubato SincFixedIn per 4096-sample read (src/audio/mod.rs:235-257) over a whole mono stream -> (samples @16 kHz, chunk_lens, device_ms)."""
        x = np.ascontiguousarray(mono, np.float32)
        cap = self.L.ss_resample_max_out(len(x), from_rate)
        out = np.empty(max(cap, 1), np.float32)
        lens = np.zeros(max(len(x) // 4096, 1), np.int32)
        n_out, ms = C.c_int64(), C.c_float()
        _check(self.L.ss_resample_stream(self.h, _p(x), len(x), from_rate, _p(out), cap, C.byref(n_out), lens.ctypes.data_as(C.c_void_p), C.byref(ms)))
        return out[: n_out.value].copy(), lens[: len(x) // 4096].copy(), ms.value

    def preprocess_stream(self, samples, chunk_len: int = 4096, chunk_lens=None, config: "DenoiseConfig | None" = None):
        """`StreamAudioProcessor` over a whole mono 16 kHz stream (src/audio/mod.rs:67-155) -> (frames [n_frames, 2048], gains, device_ms)."""
        x = np.ascontiguousarray(samples, np.float32)
        n_out = self.L.ss_preprocess_n_out(len(x))
        out = np.empty(max(n_out, 0), np.float32)
        gains = np.empty(max(n_out // 2048, 0), np.float32)
        ms = C.c_float()
        cl = np.ascontiguousarray(chunk_lens, np.int32) if chunk_lens is not None else None
        _check(self.L.ss_preprocess_stream(self.h, _p(x), len(x), cl.ctypes.data_as(C.c_void_p) if cl is not None else None,
                                           len(cl) if cl is not None else 0, chunk_len, C.byref(config) if config is not None else None,
                                           _p(out), _p(gains), C.byref(ms)))
        return out.reshape(-1, 2048), gains, ms.value

    def selftest_gemm(self, M: int, N: int, K: int, kind: int):
        """Tiled GEMM vs the on-device reference -> (max |diff|, max |ref|).  kind: 0 store, 1 GELU, 2 f32 residual, 6 f32 store."""
        err, ref = C.c_float(), C.c_float()
        _check(self.L.ss_engine_selftest_gemm(self.h, M, N, K, kind, C.byref(err), C.byref(ref)))
        return err.value, ref.value

    def selftest_gemm_ex(self, M: int, N: int, K: int, kind: int, fp8: bool = False, reps: int = 0):
        """As selftest_gemm, optionally the e4m3 kernel (fp8=True; kinds 0 -> T, 1 GELU -> e4m3, 2 f32 residual, 5 -> f32) and a timing of `reps`
        back-to-back launches -> (max |diff|, max |ref|, average ms per launch)."""
        err, ref, ms = C.c_float(), C.c_float(), C.c_float()
        _check(self.L.ss_engine_selftest_gemm_ex(self.h, M, N, K, kind, int(fp8), reps, C.byref(err), C.byref(ref), C.byref(ms)))
        return err.value, ref.value, ms.value

    def last_timing(self):
        t = np.zeros(4, np.float32)
        self.L.ss_engine_last_timing(self.h, _p(t))
        c = np.zeros(4, np.int64)
        self.L.ss_engine_last_counters(self.h, _p(c))
        return dict(mel_ms=float(t[0]), encode_ms=float(t[1]), decode_ms=float(t[2]), total_ms=float(t[3]), decoder_passes=int(c[0]),
                    decoder_rows=int(c[1]), encoder_windows=int(c[2]))

    def totals(self):
        """Cumulative device time / work over all lanes since the engine was created (ss_engine_totals)."""
        ms = np.zeros(4, np.float64)
        cnt = np.zeros(6, np.int64)
        nl = C.c_int32()
        _check(self.L.ss_engine_totals(self.h, _p(ms), _p(cnt), C.byref(nl)))
        return dict(mel_ms=float(ms[0]), encode_ms=float(ms[1]), decode_ms=float(ms[2]), total_ms=float(ms[3]), decoder_passes=int(cnt[0]),
                    decoder_rows=int(cnt[1]), encoder_windows=int(cnt[2]), admitted=int(cnt[3]), started_midway=int(cnt[4]), n_lanes=int(nl.value))

    def probe_gemm(self, batch: int, reps: int):
        ms, fl = C.c_float(), C.c_double()
        _check(self.L.ss_engine_probe_gemm(self.h, batch, reps, C.byref(ms), C.byref(fl)))
        return ms.value, fl.value


class Session:
    def __init__(self, eng: Engine):
        self.eng = eng
        self.L = eng.L
        self.h = self.L.ss_session_create(eng.h)
        if not self.h:
            raise SpeakSenseError(-1, "ss_session_create failed")
        self._pending = 0      # tickets submitted on this session and not yet waited for

    def close(self):
        if getattr(self, "h", None):
            self.L.ss_session_free(self.h)
            self.h = None

    def transcribe(self, pcm: np.ndarray, params: Params | None = None):
        pcm = np.ascontiguousarray(pcm, np.float32)
        _check(self.L.ss_transcribe(self.h, _p(pcm), len(pcm), C.byref(params) if params is not None else None))
        return self.result()

    def submit(self, pcm: np.ndarray, params: Params | None = None):
        pcm = np.ascontiguousarray(pcm, np.float32)
        t = C.c_void_p()
        _check(self.L.ss_submit(self.h, _p(pcm), len(pcm), C.byref(params) if params is not None else None, C.byref(t)))
        self._pending += 1
        return t

    def submit_device(self, ptr: int, n: int, params: Params | None = None):
        """Async submit of PCM already resident on the engine's GPU (device pointer); the buffer must outlive wait()."""
        t = C.c_void_p()
        _check(self.L.ss_submit_ex(self.h, C.c_void_p(int(ptr)), int(n), C.byref(params) if params is not None else None, 1, C.byref(t)))
        self._pending += 1
        return t

    def wait(self, ticket):
        """Blocks until the chunk is done; returns its results -- or None when ANOTHER ticket of this session is still outstanding: the results
        live on the session ("valid until its next transcribe/submit", include/speaksense.h), and with a second chunk queued or running the engine is
        writing them.  Reading then is a data race, and the bulk getters size their arrays from an earlier call: a heap overflow in the caller."""
        self._pending -= 1
        _check(self.L.ss_wait(ticket))
        return self.result() if self._pending <= 0 else None

    def ready(self, ticket) -> bool:
        """Non-blocking: has the chunk behind `ticket` completed (wait() will return at once)?"""
        return bool(self.L.ss_ticket_ready(ticket))

    def token_times(self):
        """whisper_full_get_token_data per segment: [dict(ids, t0, t1, vlen)], token-level times in 10 ms units (-1 when Params.token_timestamps was 0)."""
        out = []
        for i in range(self.L.ss_result_n_segments(self.h)):
            k = self.L.ss_result_segment_n_tokens(self.h, i)
            ids, tids, t0, t1, vl = np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros(k, np.int64), np.zeros(k, np.int64), np.zeros(k, np.float32)
            pt, ptsum = np.zeros(k, np.float32), np.zeros(k, np.float32)
            a, b, v, tid, tt, o4 = C.c_int64(), C.c_int64(), C.c_float(), C.c_int32(), C.c_int32(), (C.c_float * 4)()
            for q in range(k):
                _check(self.L.ss_result_segment_token(self.h, i, q, C.byref(tid), C.byref(tt), o4))
                _check(self.L.ss_result_segment_token_times(self.h, i, q, C.byref(a), C.byref(b), C.byref(v)))
                ids[q], tids[q], t0[q], t1[q], vl[q], pt[q], ptsum[q] = tid.value, tt.value, a.value, b.value, v.value, o4[2], o4[3]
            out.append(dict(ids=ids, tid=tids, pt=pt, ptsum=ptsum, t0=t0, t1=t1, vlen=vl))
        return out

    def result(self):
        segs = []
        for i in range(self.L.ss_result_n_segments(self.h)):
            first_id, first_tid, o4 = C.c_int32(-1), C.c_int32(-1), (C.c_float * 4)()
            if self.L.ss_result_segment_n_tokens(self.h, i) > 0:
                self.L.ss_result_segment_token(self.h, i, 0, C.byref(first_id), C.byref(first_tid), o4)
            segs.append(dict(text=self.L.ss_result_segment_text(self.h, i), t0=self.L.ss_result_segment_t0(self.h, i),
                             t1=self.L.ss_result_segment_t1(self.h, i),
                             speaker_turn_next=bool(self.L.ss_result_segment_speaker_turn_next(self.h, i)),
                             first_id=int(first_id.value)))   # whisper.cpp derives t0 from the first token's `tid` (argmax over timestamps)
        n = self.L.ss_result_n_tokens(self.h)
        ids = np.zeros(n, np.int32)
        plog = np.zeros(n, np.float32)
        if n:
            self.L.ss_result_tokens(self.h, _p(ids), _p(plog))
        ns = self.L.ss_result_n_sampled_tokens(self.h)
        sampled = np.zeros(ns, np.int32)
        if ns:
            self.L.ss_result_sampled_tokens(self.h, _p(sampled))
        nt = self.L.ss_result_n_trace_tokens(self.h)
        trace = np.zeros(nt, np.int32)
        if nt:
            self.L.ss_result_trace_tokens(self.h, _p(trace))
        c = np.zeros(4, np.int32)
        self.L.ss_result_counters(self.h, _p(c))
        return dict(segments=segs, tokens=ids, plog=plog, sampled=sampled, trace=trace, n_encode=int(c[0]), n_decode=int(c[1]), n_fail=int(c[2]),
                    n_windows=int(c[3]), lang_id=int(self.L.ss_result_lang_id(self.h)))

    def rng_draws(self, decoder: int = 0) -> int:
        """Invocations so far of the std::mt19937 the session carries from chunk to chunk (decoder 0's generator; whisper_state::rng under
        COMPAT_RNG_STATE), consumed only by temperature-fallback sampling.  decoder >= 1: that decoder's own generator since the last chunk began."""
        return int(self.L.ss_session_rng_draws_decoder(self.h, decoder)) if decoder else int(self.L.ss_session_rng_draws(self.h))

    def rng_discard(self, n: int):
        _check(self.L.ss_session_rng_discard(self.h, int(n)))

    def set_encoder(self, enc: np.ndarray):
        """enc: [n_audio_ctx][n_audio_state], or fewer rows = the output of a shortened context (Engine.encode(audio_ctx=...))."""
        enc = np.ascontiguousarray(enc, np.float32)
        if enc.shape[0] < self.eng.n_audio_ctx:
            _check(self.L.ss_session_set_encoder_ctx(self.h, _p(enc), int(enc.shape[0])))
            return
        _check(self.L.ss_session_set_encoder(self.h, _p(enc)))

    def decode(self, tokens, n_past: int) -> np.ndarray:
        t = np.ascontiguousarray(tokens, np.int32)
        out = np.empty(self.eng.n_vocab, np.float32)
        _check(self.L.ss_session_decode(self.h, _p(t), len(t), n_past, _p(out)))
        return out


class Pool:
    """One model on several GPUs of a node (ss_pool_*): N engines, chunks routed to the least-loaded one, ties round-robin."""

    def __init__(self, model_path: str, device_ids, dtype: int = DTYPE_F16, max_batch: int = 8, max_decoders: int = 5, batch_wait_us: int = 2000,
                 n_lanes: int = 0, compat: int = 0):
        self.L = lib()
        ids = np.ascontiguousarray(device_ids, np.int32)
        o = EngineOpts(0, dtype, max_batch, max_decoders, batch_wait_us, n_lanes, compat)
        h = C.c_void_p()
        _check(self.L.ss_pool_create(model_path.encode(), _p(ids), len(ids), C.byref(o), C.byref(h)))
        self.h = h
        self.n_engines = int(self.L.ss_pool_n_engines(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.L.ss_pool_free(self.h)
            self.h = None

    def new_session(self) -> "PoolSession":
        return PoolSession(self)


class PoolSession(Session):
    def __init__(self, pool: Pool):
        self.pool = pool
        self.L = pool.L
        self.eng = None
        self.h = self.L.ss_pool_session_create(pool.h)
        if not self.h:
            raise SpeakSenseError(-1, "ss_pool_session_create failed")
        self._pending = 0

    def submit(self, pcm: np.ndarray, params: Params | None = None):
        pcm = np.ascontiguousarray(pcm, np.float32)
        t = C.c_void_p()
        _check(self.L.ss_pool_submit(self.pool.h, self.h, _p(pcm), len(pcm), C.byref(params) if params is not None else None, C.byref(t)))
        self._pending += 1
        return t

    def transcribe(self, pcm: np.ndarray, params: Params | None = None):
        return self.wait(self.submit(pcm, params))

    def last_engine(self) -> int:
        return int(self.L.ss_pool_last_engine(self.h))


def pool_pick(load, cursor: int) -> int:
    a = np.ascontiguousarray(load, np.int32)
    return int(lib().ss_pool_pick(_p(a), len(a), cursor & 0xFFFFFFFF))


def e4m3_from_f32(x: np.ndarray) -> np.ndarray:
    """OCP e4m3 codes (uint8) of float32 values: the host converter the fp8 engine quantises weights with.  Needs no GPU."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.uint8)
    _check(lib().ss_e4m3_from_f32(_p(x), _p(out), x.size))
    return out
