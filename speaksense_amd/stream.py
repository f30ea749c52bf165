"""Host-side mirror of the reference's gRPC streaming handler (SURVEY.md §8f "next" #2), in-process.

Follows /root/reference/src/grpc/handlers/asr.rs:
  constants :13-18 (CHUNK_SIZE = 160 000 BYTES = 5 s of PCM16, OVERLAP_SIZE = 16 000 bytes = 0.5 s),
  StreamContext :24-60 (block-relative -> absolute milliseconds, monotonic repair),
  process_text :69-136 (what to emit given the previous text), the per-stream loop :146-280
  (base64 -> bytes -> PCM16LE / 32767 -> denoise_audio -> transcribe_with_state -> de-dup -> response).
Wire format: proto/asr.proto:22-44 (TranscribeRequest{end, audio(base64), device_id}, TranscribeResponse{end, text, device_id, segments}).
There is no tonic/gRPC server here (network plumbing is out of scope): `GrpcStreamSession.feed` is what the handler does per request
message, so a C++/Rust server only has to forward messages.  Quirks are kept: segment times arrive in centiseconds and are treated as
seconds (asr.rs:39-43), language is hard-wired to "zh" with stream mode (asr.rs:154-157), the final flush uses a fresh state and no
denoiser (asr.rs:247).
"""
from __future__ import annotations

import base64
import re
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import asr as asr_mod

SAMPLE_RATE = 16000
CHUNK_SIZE = SAMPLE_RATE * 10   # bytes
OVERLAP_SIZE = SAMPLE_RATE      # bytes
_SENT_SPLIT = re.compile("[。！？.!?]")
_SENT_END = "。！？.!?"


@dataclass
class StreamContext:  # asr.rs:24-60
    block_index: int = 0
    last_text: str = ""
    last_end_time: float = 0.0

    def calculate_segment_time(self, segment_start: float, segment_end: float):
        block_base_time = self.block_index * 5.0
        abs_start = int((block_base_time + segment_start) * 1000.0)
        abs_end = int((block_base_time + segment_end) * 1000.0)
        last_end_ms = int(self.last_end_time * 1000.0)
        if abs_start < last_end_ms:
            diff = last_end_ms - abs_start
            abs_start = last_end_ms
            abs_end += diff
        self.last_end_time = abs_end / 1000.0
        return abs_start, abs_end

    def next_block(self):
        self.block_index += 1


def process_text(new_text: str, last_text: str, segments) -> Optional[str]:  # asr.rs:69-136 (lengths are UTF-8 byte lengths, as in Rust)
    if not last_text:
        return new_text
    if segments:
        last_segment = segments[-1]
        if last_segment.text not in last_text:
            return last_segment.text
    nb, lb = len(new_text.encode("utf-8")), len(last_text.encode("utf-8"))
    if nb > lb and new_text.startswith(last_text):
        added = new_text[len(last_text):]
        if added.strip():
            return added.strip()
    if nb > lb * 2 or lb > nb * 2:
        return new_text
    if new_text != last_text:
        new_s = [s for s in _SENT_SPLIT.split(new_text) if s.strip()]
        last_s = [s for s in _SENT_SPLIT.split(last_text) if s.strip()]
        if len(new_s) > len(last_s):
            content = "".join(new_s[len(last_s):]).strip()
            if content:
                if new_text and new_text[-1] in _SENT_END:
                    content += new_text[-1]
                return content
        elif new_s and last_s:
            if new_s[-1].strip() != last_s[-1].strip():
                result = new_s[-1].strip()
                if new_text and new_text[-1] in _SENT_END:
                    result += new_text[-1]
                return result
    return None


def pcm16_bytes_to_f32(buf: bytes) -> np.ndarray:  # asr.rs:188-194 / 236-245
    n = len(buf) // 2
    out = np.frombuffer(buf[: 2 * n], dtype="<i2").astype(np.float32) / np.float32(32767.0)
    if len(buf) % 2:
        out = np.concatenate([out, np.zeros(1, np.float32)])   # a dangling byte becomes 0.0 (chunks(2) with len 1)
    return out


@dataclass
class Segment:       # proto/asr.proto:40-44
    start: int
    end: int
    text: bytes


@dataclass
class TranscribeResponse:   # proto/asr.proto:33-38
    end: int
    text: bytes
    device_id: str
    segments: List[Segment] = field(default_factory=list)


class GrpcStreamSession:
    """One bidi stream: state created once (asr.rs:164), requests fed one by one, responses returned per request."""

    def __init__(self, engine: asr_mod.WhisperAsr):
        self.engine = engine
        self.params = asr_mod.AsrParams(language="zh", stream_mode=True, min_segment_length=5)   # asr.rs:154-157
        self.state = engine.create_state()
        self.ctx = StreamContext()
        self.buf = bytearray()
        self.device_id = ""
        self.done = False

    def feed(self, audio_b64: bytes, end: int = 0, device_id: str = "") -> List[TranscribeResponse]:
        if self.done:
            return []
        out: List[TranscribeResponse] = []
        if not self.device_id:
            self.device_id = device_id
        try:
            self.buf.extend(base64.b64decode(audio_b64, validate=True))
        except Exception:
            return out                      # "Failed to decode audio" -> continue (asr.rs:177-183)
        if len(self.buf) >= CHUNK_SIZE:
            float_data = pcm16_bytes_to_f32(bytes(self.buf[:CHUNK_SIZE]))
            denoised = self.engine.engine.denoise_audio(float_data)[0]
            try:
                result = self.engine.transcribe_with_state(self.state, denoised, self.params)
                for seg in result.segments:
                    new_text = process_text(seg.text, self.ctx.last_text, [seg])
                    if new_text is not None:
                        self.ctx.last_text = seg.text
                        s, e = self.ctx.calculate_segment_time(seg.start, seg.end)
                        out.append(TranscribeResponse(0, new_text.encode(), self.device_id, [Segment(s, e, seg.text.encode())]))
                self.ctx.next_block()
            except Exception:
                pass                        # "ASR processing failed" is logged and the stream continues (asr.rs:228)
            del self.buf[: CHUNK_SIZE - OVERLAP_SIZE]
        if end == 1 and len(self.buf) > 0:
            float_data = pcm16_bytes_to_f32(bytes(self.buf))
            try:
                result = self.engine.transcribe(float_data, self.params)     # fresh state, no denoise (asr.rs:247)
                final_text = process_text(result.full_text, self.ctx.last_text, result.segments)
                if final_text is not None:
                    segs = []
                    for seg in result.segments:
                        s, e = self.ctx.calculate_segment_time(seg.start, seg.end)
                        segs.append(Segment(s, e, seg.text.encode()))
                    out.append(TranscribeResponse(1, final_text.encode(), self.device_id, segs))
            except Exception:
                pass
            self.done = True
        return out


def client_messages(pcm_f32: np.ndarray, message_bytes: int = 32 * 1024):
    """What examples/asr_client.rs:142,166-212 sends: PCM16LE, 32 KiB raw chunks, each base64-encoded; `end = 1` on the last one."""
    pcm16 = np.clip(np.round(np.asarray(pcm_f32, np.float64) * 32767.0), -32768, 32767).astype("<i2").tobytes()
    msgs = [pcm16[i : i + message_bytes] for i in range(0, len(pcm16), message_bytes)] or [b""]
    return [(base64.b64encode(m), 1 if i == len(msgs) - 1 else 0) for i, m in enumerate(msgs)]


def serve_streams(engine: asr_mod.WhisperAsr, streams, device_ids=None):
    """Config #4's shape in-process: one thread per stream (the reference runs one tokio task per stream, asr.rs:160), all sharing one
    engine.  `streams[i]` is the list of (base64 message, end) a client sends.  With `engine.batch_across_callers` the 5 s chunks of
    different streams meet in the engine's batch former.  Returns the per-stream response lists."""
    import threading
    results = [None] * len(streams)
    errors = []

    def run(i):
        try:
            sess = GrpcStreamSession(engine)
            out = []
            for m, end in streams[i]:
                out.extend(sess.feed(m, end, device_ids[i] if device_ids else f"stream-{i}"))
            results[i] = out
        except Exception as e:   # pragma: no cover - surfaced below
            errors.append((i, e))

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(streams))]
    for t in th: t.start()
    for t in th: t.join()
    if errors:
        raise RuntimeError(f"stream {errors[0][0]} failed: {errors[0][1]}")
    return results
