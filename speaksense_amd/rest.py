"""Host-side mirror of the reference's REST transcription pipeline (SURVEY.md §8f "next" #3): schemas and the 30 s chunker.

Schemas (serde JSON shapes): TranscribeRequest /root/reference/src/web/handlers/asr.rs:29-46; the handler's TaskConfig defaults :77-92;
TaskConfig / TaskParams / TranscribeParams / TaskStatus / TaskResult / TranscribeResult / TranscribeSegment / CallbackType
/root/reference/src/schedule/types.rs:36-156; callback payload {task_id, status, data} /root/reference/src/schedule/callback/mod.rs:28-33,
68-95.  Processor: /root/reference/src/schedule/processors/transcribe.rs:31-166 (process_audio) and :186-203 (validate_params).
No HTTP server, task queue or SQLite here (out of scope); `TranscribeProcessor.process_audio` is the part that forms the work units and
calls the engine.  The reference transcribes a file's chunks strictly one after another on one state (transcribe.rs:100-142); both callers
run in stream mode (no_context), so the chunks are independent and `batched=True` hands all of a file's chunks to the engine at once.
"""
from __future__ import annotations

import json
import wave
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import asr as asr_mod

BUFFER_SIZE = 16000 * 30      # transcribe.rs:104
READ_CHUNK = 4096             # audio/mod.rs:184 (interleaved samples per read)


@dataclass
class TranscribeRequest:      # web/handlers/asr.rs:36-46
    path: str
    path_type: str            # "Url" | "Local"
    callback_url: str
    language: Optional[str] = None
    speaker_diarization: bool = False
    emotion_recognition: bool = False
    filter_dirty_words: bool = False

    @staticmethod
    def from_json(text: str) -> "TranscribeRequest":
        d = json.loads(text)
        for k in ("path", "path_type", "callback_url", "speaker_diarization", "emotion_recognition", "filter_dirty_words"):
            if k not in d:
                raise ValueError(f"missing field `{k}`")       # serde: only Option<> fields may be absent
        if d["path_type"] not in ("Url", "Local"):
            raise ValueError(f"unknown variant `{d['path_type']}`, expected `Url` or `Local`")
        return TranscribeRequest(d["path"], d["path_type"], d["callback_url"], d.get("language"), bool(d["speaker_diarization"]),
                                 bool(d["emotion_recognition"]), bool(d["filter_dirty_words"]))


def task_config_from_request(req: TranscribeRequest) -> dict:    # web/handlers/asr.rs:77-92, serde layout of types.rs:36-156
    return {
        "task_type": "Transcribe", "input_path": req.path, "path_type": req.path_type,
        "callback_type": {"type": "Http", "config": {"url": req.callback_url}},
        "params": {"type": "Transcribe", "params": {"language": req.language, "speaker_diarization": req.speaker_diarization,
                                                     "emotion_recognition": req.emotion_recognition, "filter_dirty_words": req.filter_dirty_words}},
        "priority": "Normal", "retry_count": 0, "max_retries": 3, "timeout": None,
    }


def http_response(code: int, message: str, data) -> dict:       # utils/http.rs HttpResponse<T>
    return {"code": code, "message": message, "data": data}


def validate_params(params: dict) -> None:                       # transcribe.rs:186-203
    if params.get("type") != "Transcribe":
        raise ValueError("Invalid task params type")
    lang = params["params"].get("language")
    if lang is not None and lang not in ("zh", "en", "ja"):
        raise ValueError(f"Unsupported language: {lang}")


@dataclass
class TaskSegment:            # types.rs:132-138
    text: str
    speaker_id: Optional[int]
    start_time: float
    end_time: float


@dataclass
class TaskTranscribeResult:   # types.rs:126-130
    text: str = ""
    segments: List[TaskSegment] = field(default_factory=list)

    def task_result(self) -> dict:   # TaskResult::Transcribe, #[serde(tag = "type", content = "result")]
        return {"type": "Transcribe", "result": {"text": self.text, "segments": [
            {"text": s.text, "speaker_id": s.speaker_id, "start_time": s.start_time, "end_time": s.end_time} for s in self.segments]}}


def callback_on_complete(task_id: str, result: TaskTranscribeResult) -> dict:    # callback/mod.rs:78-85
    return {"task_id": task_id, "status": "Completed", "data": result.task_result()}


def callback_on_error(task_id: str, error: str) -> dict:                          # callback/mod.rs:87-94
    return {"task_id": task_id, "status": {"Failed": error}, "data": error}


def read_wav_i16(path: str):
    """hound WavReader::samples::<i16>() as parse_audio_file_stream uses it (audio/mod.rs:165-190): interleaved i16 / 32768."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("Unsupported bits per sample: expected 16 bits")
        ch, sr = w.getnchannels(), w.getframerate()
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    return raw.astype(np.float32) / np.float32(32768.0), ch, sr


def convert_to_mono(samples: np.ndarray, num_channels: int) -> np.ndarray:      # audio/mod.rs:391-397, per 4096-sample read chunk
    out = []
    for i in range(0, len(samples), READ_CHUNK):
        c = samples[i : i + READ_CHUNK]
        n_full = len(c) // num_channels
        acc = np.zeros(n_full, np.float32)
        body = c[: n_full * num_channels].reshape(n_full, num_channels)
        for k in range(num_channels):
            acc = (acc + body[:, k]).astype(np.float32)
        out.append((acc / np.float32(num_channels)).astype(np.float32))
        if len(c) % num_channels:
            out.append(np.array([np.sum(c[n_full * num_channels:], dtype=np.float32) / np.float32(num_channels)], np.float32))
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def rest_chunks(frames: np.ndarray) -> List[np.ndarray]:         # transcribe.rs:100-142
    out, buf, n = [], [], 0
    for f in frames:
        buf.append(f); n += len(f)
        if n >= BUFFER_SIZE:
            out.append(np.concatenate(buf)); buf, n = [], 0
    if buf:
        out.append(np.concatenate(buf))
    return out


class TranscribeProcessor:    # transcribe.rs:21-166
    def __init__(self, asr: asr_mod.WhisperAsr, batched: bool = True):
        self.asr, self.batched = asr, batched

    def preprocess(self, interleaved: np.ndarray, channels: int, sample_rate: int) -> np.ndarray:
        """parse_audio_file_stream (audio/mod.rs:158-233) -> the [n_frames, 2048] callbacks, on the engine's GPU."""
        eng = self.asr.engine
        finish = True
        if sample_rate != 16000:
            # rubato's SincFixedIn::process wants exactly 4096 frames: multi-channel files (mono chunks of 4096/channels) fail on the first
            # read and a short last read fails too; either way the file's processing ends there without finish() (mod.rs:196-217, 92-96)
            if channels != 1 or len(interleaved) < READ_CHUNK:
                return np.zeros((0, 2048), np.float32)
            mono, chunk_lens, _ = eng.resample_stream(interleaved, sample_rate)
            finish = len(interleaved) % READ_CHUNK == 0
            chunk_len = READ_CHUNK
            keep = chunk_lens > 0
            chunk_lens = chunk_lens[keep]
        else:
            mono = convert_to_mono(interleaved, channels)
            chunk_lens, chunk_len = None, READ_CHUNK // channels
            if READ_CHUNK % channels:      # read chunks do not align with channel groups: explicit lengths
                chunk_lens = np.array([len(convert_to_mono(interleaved[i : i + READ_CHUNK], channels)) for i in range(0, len(interleaved), READ_CHUNK)], np.int32)
        if len(mono) == 0:
            return np.zeros((0, 2048), np.float32)
        frames, _, _ = eng.preprocess_stream(mono, chunk_len, chunk_lens)
        if not finish and len(mono) % 2048:
            frames = frames[:-1]
        return frames

    def process_samples(self, interleaved: np.ndarray, channels: int, sample_rate: int, language: Optional[str], speaker_diarization: bool):
        frames = self.preprocess(np.asarray(interleaved, np.float32), channels, sample_rate)
        p = asr_mod.AsrParams(language=language, speaker_diarization=speaker_diarization, stream_mode=True)   # transcribe.rs:66-70
        chunks = rest_chunks(frames)
        if self.batched and chunks:
            # The reference runs the chunks one after another on ONE state.  With no_context they share nothing but the state's sampler
            # (std::mt19937, never reseeded): a chunk that falls back to temperature sampling draws from a generator advanced by all earlier
            # chunks' draws.  All chunks go out as one batch on fresh sessions; a chunk that sampled although earlier chunks had already
            # drawn is redone with its generator advanced to the serial position.  Chunks that never sample (the normal case) need no redo.
            states = [self.asr.create_state() for _ in chunks]
            results = self.asr.transcribe_many(states, chunks, p)
            prefix = 0
            for k, st in enumerate(states):
                draws = st.rng_draws()
                if prefix > 0 and draws > 0:
                    st = self.asr.create_state()
                    st.rng_discard(prefix)
                    results[k] = self.asr.transcribe_with_state(st, chunks[k], p)
                    draws = st.rng_draws() - prefix
                prefix += draws
        elif self.batched:
            results = []
        else:
            state = self.asr.create_state()
            results = [self.asr.transcribe_with_state(state, c, p) for c in chunks]
        out = TaskTranscribeResult()
        for r in results:                                  # transcribe.rs:146-165
            out.text += r.full_text
            out.segments += [TaskSegment(s.text, s.speaker_id, s.start, s.end) for s in r.segments]
        return out

    def process_audio(self, task_config: dict) -> TaskTranscribeResult:
        validate_params(task_config["params"])
        if task_config["path_type"] != "Local":
            raise RuntimeError("Failed to download audio: no network transport in this build (PathType::Url is the service's job)")
        if not task_config["input_path"].lower().endswith(".wav"):
            raise RuntimeError("FFmpeg conversion is the service's job: hand in 16-bit PCM WAV")     # audio/mod.rs:314-340
        samples, ch, sr = read_wav_i16(task_config["input_path"])
        pr = task_config["params"]["params"]
        return self.process_samples(samples, ch, sr, pr.get("language"), pr.get("speaker_diarization", False))
