"""CPU: the gRPC streaming handler's host logic (speaksense_amd/stream.py) against hand-derived expectations from
/root/reference/src/grpc/handlers/asr.rs:24-136."""
import base64

import numpy as np

from speaksense_amd import stream
from speaksense_amd.asr import TranscribeSegment as Seg


def test_constants():
    assert stream.CHUNK_SIZE == 160000 and stream.OVERLAP_SIZE == 16000     # bytes: 5 s and 0.5 s of PCM16 (asr.rs:13-18)


def test_calculate_segment_time():
    c = stream.StreamContext()
    assert c.calculate_segment_time(0.0, 250.0) == (0, 250000)            # centiseconds treated as seconds (kept quirk)
    assert c.last_end_time == 250.0
    c.next_block()
    # block 1 starts at 5 s; a segment starting before the previous end is pushed after it and keeps its duration
    assert c.calculate_segment_time(10.0, 60.0) == (250000, 65000 + (250000 - 15000))
    c2 = stream.StreamContext(block_index=3)
    assert c2.calculate_segment_time(1.5, 2.25) == (16500, 17250)


def test_process_text_rules():
    P = stream.process_text
    s = lambda t: [Seg(t, 0, 0.0, 1.0)]
    assert P("你好", "", s("你好")) == "你好"                                  # first text passes through
    assert P("abc def", "abc", s("zzz")) == "zzz"                             # last segment not contained in last_text -> the segment
    assert P("abc def", "abc", s("abc")) == "def"                             # pure extension -> the added part, trimmed
    assert P("abc   ", "abc", s("abc")) is None                               # only whitespace added, similar length, one sentence
    assert P("a", "abcdefgh", s("a")) == "a"                                  # very different lengths -> the new text
    assert P("一。二。三。", "一。二。", s("一")) == "三。"                       # more sentences -> the new ones + final punctuation
    assert P("今天好。后天坏", "今天好。明天", s("今天好")) == "后天坏"             # same count, last sentence changed
    assert P("same.", "same.", s("same.")) is None
    # byte lengths, as Rust: 3 CJK chars (9 bytes) vs 4 ASCII
    assert P("abcd", "你好吗", s("abcd")) == "abcd"                            # segment not in last_text
    assert P("你好吗你", "你好吗", s("你好吗")) == "你"


def test_pcm16_decode_and_client_messages():
    x = np.array([0.0, 0.5, -0.5, 1.0, -1.0], np.float32)
    msgs = stream.client_messages(x, message_bytes=4)
    assert [e for _, e in msgs] == [0, 0, 1]
    raw = b"".join(base64.b64decode(m) for m, _ in msgs)
    back = stream.pcm16_bytes_to_f32(raw)
    assert np.allclose(back, [0.0, 16384 / 32767.0, -16384 / 32767.0, 1.0, -1.0], atol=1e-6)
    assert stream.pcm16_bytes_to_f32(b"\x01\x00\xff").tolist() == [float(np.float32(1) / np.float32(32767.0)), 0.0]   # dangling byte -> 0.0
