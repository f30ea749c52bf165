"""Host-side mirror of the reference's `src/asr` interface, over the C ABI.

Same names, argument meaning and error behaviour as
  /root/reference/src/asr/mod.rs:9-73       AsrParams, TranscribeSegment, TranscribeResult, trait AsrEngine
  /root/reference/src/asr/whisper.rs:16-223 WhisperAsr::{new, create_state, transcribe_with_state, transcribe},
                                            is_promotional_text, add_punctuation, build_params
The Rust toolchain is not available in this image, so this Python class is the compiled-and-tested host layer;
the Rust shim a maintainer would drop into the reference is in speaksense_amd/rust/ (source only) and INTEGRATION.md.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import binding

# whisper.rs:9-14
PROMOTIONAL_TEXT = [
    "请不吝点赞", "請不吝點贊", "點贊", "訂閱", "订阅", "打赏", "打賞", "打賞支持明鏡與點點欄目", "打赏支持明镜与点点栏目",
    "並且按下小鈴鐺才能收到最新消息哦!", "請按讚、訂閱、分享!", "明镜需要您的支持 欢迎收看订阅明镜",
    "請按讚,訂閱,分享,打開小鈴鐺,並且按下小鈴鐺才能收到最新消息謝謝觀看",
    "請按讚,訂閱,分享,打開小鈴鐺,並且按下小鈴鐺才能收到最新消息哦!",
]


@dataclass
class AsrParams:  # mod.rs:9-42
    language: Optional[str] = None
    speaker_diarization: bool = False
    stream_mode: bool = False
    min_segment_length: int = 10  # set by callers, never read by the engine (mod.rs:14) -- kept for parity

    def set_language(self, language): self.language = language
    def set_speaker_diarization(self, enable): self.speaker_diarization = enable
    def set_stream_mode(self, enable): self.stream_mode = enable
    def set_min_segment_length(self, length): self.min_segment_length = length


@dataclass
class TranscribeSegment:  # mod.rs:44-50
    text: str
    speaker_id: int
    start: float
    end: float


@dataclass
class TranscribeResult:  # mod.rs:52-56
    segments: List[TranscribeSegment] = field(default_factory=list)
    full_text: str = ""


def is_promotional_text(text: str) -> bool:  # whisper.rs:41-43
    return any(p in text for p in PROMOTIONAL_TEXT)


def add_punctuation(text: str) -> str:  # whisper.rs:175-201
    if text.endswith(("。", "！", "？", "，")):
        return text
    contains_question = any(k in text for k in ("吗", "呢", "什么", "为何", "怎么"))
    contains_exclaim = any(k in text for k in ("啊", "哇", "太", "真", "好", "真是"))
    if contains_question:
        return text + "？"
    if contains_exclaim:
        return text + "！"
    return text + " "


class WhisperAsr:
    """`WhisperAsr::new(model_path)` (whisper.rs:21-28) -- loads the ggml model onto one MI355X."""

    def __init__(self, model_path: str, device: int = 0, dtype: int = binding.DTYPE_F16, max_batch: int = 8, batch_across_callers: bool = False,
                 batch_wait_us: int = 2000, compat: int = 0):
        # batch_across_callers: transcribe_with_state goes through ss_submit/ss_wait, so chunks that concurrent callers (one thread
        # per gRPC stream, asr.rs:164) hand in at about the same time share one device batch; results are identical either way.
        self.batch_across_callers = batch_across_callers
        self.params_hook = None    # benchmarks only: callable(binding.Params) applied after build_params (e.g. Mode F fixed_steps on random weights)
        try:
            self.engine = binding.Engine(model_path, device=device, dtype=dtype, max_batch=max_batch, batch_wait_us=batch_wait_us, compat=compat)
        except binding.SpeakSenseError as e:
            raise RuntimeError(f"failed to open whisper model: {e}") from e  # whisper.rs:24

    def create_state(self) -> binding.Session:  # whisper.rs:30-39
        return self.engine.new_session()

    def build_params(self, ap: AsrParams) -> binding.Params:  # whisper.rs:131-173 + 60-71
        p = binding.default_params()
        p.tdrz_enable = 1 if ap.speaker_diarization else 0
        p.no_context = 0           # build_params: set_no_context(false) ...
        if ap.language is not None:
            p.language = ap.language.encode()
        if ap.stream_mode:         # ... overridden by stream mode (whisper.rs:65-69)
            p.single_segment = 0
            p.no_context = 1
            p.audio_ctx = 0
        if getattr(self, "params_hook", None) is not None:
            self.params_hook(p)
        return p

    def _collect(self, res: dict, user_params: AsrParams) -> TranscribeResult:  # whisper.rs:77-128
        out = TranscribeResult()
        current_speaker = 0
        segs = res["segments"]
        for i, s in enumerate(segs):
            text = s["text"].decode("utf-8")  # strict: invalid UTF-8 errors the whole chunk (whisper.rs:85)
            if is_promotional_text(text):
                continue
            if i > 0 and segs[i - 1]["speaker_turn_next"]:
                current_speaker += 1
            processed = add_punctuation(text)
            seg = TranscribeSegment(processed, current_speaker, float(s["t0"]), float(s["t1"]))
            if user_params.stream_mode:
                if i == len(segs) - 1:
                    out.segments.append(seg)
                    out.full_text = processed
            else:
                out.segments.append(seg)
                out.full_text += processed
        return out

    def transcribe_with_state(self, state: binding.Session, audio, user_params: AsrParams) -> TranscribeResult:  # whisper.rs:45-129
        if self.batch_across_callers:
            res = state.wait(state.submit(np.asarray(audio, np.float32), self.build_params(user_params)))
            if res is None:   # binding.Session.wait withholds results while another ticket of the session is outstanding
                raise RuntimeError("transcribe_with_state: the state has another chunk in flight -- a state serves one caller at a time "
                                   "(the reference guards it with a Mutex, /root/reference/src/asr/whisper.rs:38,51)")
        else:
            res = state.transcribe(np.asarray(audio, np.float32), self.build_params(user_params))
        return self._collect(res, user_params)

    def transcribe(self, audio, params: AsrParams) -> TranscribeResult:  # mod.rs:69-72
        return self.transcribe_with_state(self.create_state(), audio, params)

    def transcribe_many(self, states, audios, user_params: AsrParams) -> List[TranscribeResult]:
        """Batched form the reference lacks (its REST worker is strictly serial, transcribe.rs:105-125):
        one device batch over independent chunks, same per-chunk results."""
        res = self.engine.transcribe_batch(states, [np.asarray(a, np.float32) for a in audios], self.build_params(user_params))
        return [self._collect(r, user_params) for r in res]
