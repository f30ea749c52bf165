"""ORACLE (test infrastructure only): numpy restatement of the reference's stream pre-processor and REST 30 s chunker.

Follows /root/reference/src/audio/mod.rs: StreamAudioProcessor :67-155 (process_chunk :89-106, process_frame :108-141, finish :143-154),
preemphasis :260-269, normalize_audio :408-411, convert_to_mono :391-397, estimate_noise_floor :744-762, parse_audio_file_stream :158-233
(4096-sample read chunks, i16 / 32768), and the chunk loop of /root/reference/src/schedule/processors/transcribe.rs:100-142 (BUFFER_SIZE = 480 000).
Rust semantics that matter and are kept:
  * estimate_noise_floor of ONE 2048-sample frame sees two 1024-sample energies, `(2 as f32 * 0.1) as usize` == 0 of them are "noise frames",
    so it returns 0.0 / 0 = NaN; the floor is initialised once (NaN != 0.0) and stays NaN; `energy > NaN` is false and
    `(energy / NaN).max(0.1)` is 0.1 (f32::max ignores NaN): every frame is scaled by 0.1.  np.fmax / np.fmin have the same NaN rule.
  * each 2048-sample frame is denoised on its own (one STFT frame; the Hann^2 normalisation is ill-conditioned at the frame edges).
The reference holds no vectors for this (its tests need an absent ./test/a.wav): parity unpinned, pinned by construction only.
"""
from __future__ import annotations

import numpy as np

from . import denoise_oracle as dn

F32 = np.float32
FRAME = 2048
READ_CHUNK = 4096          # parse_audio_file_stream: interleaved samples per read chunk
BUFFER_SIZE = 16000 * 30   # transcribe.rs:104


def convert_to_mono(samples: np.ndarray, num_channels: int) -> np.ndarray:
    s = np.asarray(samples, F32)
    n_full = len(s) // num_channels
    out = []
    body = s[: n_full * num_channels].reshape(n_full, num_channels)
    acc = np.zeros(n_full, F32)
    for c in range(num_channels):
        acc = (acc + body[:, c]).astype(F32)
    out.append((acc / F32(num_channels)).astype(F32))
    if len(s) % num_channels:   # par_chunks: a short last chunk is still divided by num_channels
        out.append(np.array([np.sum(s[n_full * num_channels:], dtype=F32) / F32(num_channels)], F32))
    return np.concatenate(out)


def normalize_audio(chunk: np.ndarray) -> np.ndarray:
    c = np.asarray(chunk, F32)
    if len(c) == 0:
        return c
    with np.errstate(all="ignore"):
        return (c / np.max(np.abs(c))).astype(F32)


def _seq_sum_f32(v: np.ndarray) -> np.float32:
    acc = F32(0.0)
    for x in np.asarray(v, F32):
        acc = F32(acc + x)
    return acc


def estimate_noise_floor(samples: np.ndarray) -> np.float32:
    e = [F32(_seq_sum_f32(samples[i : i + 1024] ** 2) / F32(len(samples[i : i + 1024]))) for i in range(0, len(samples), 1024)]
    e.sort()
    cnt = int(F32(len(e)) * F32(0.1))
    with np.errstate(all="ignore"):
        return F32(_seq_sum_f32(np.array(e[:cnt], F32)) / F32(cnt))


def preemphasis(frame: np.ndarray, coefficient=0.97) -> np.ndarray:
    f = np.asarray(frame, F32)
    out = f.copy()
    out[1:] = (f[1:] - (F32(coefficient) * f[:-1]).astype(F32)).astype(F32)
    return out


class StreamAudioProcessor:
    def __init__(self, config: dn.DenoiseConfig | None = None, exact_sums: bool = False):
        self.config = config or dn.DenoiseConfig()
        self.buffer = np.zeros(0, F32)
        self.prev_energy = F32(0.0)
        self.noise_floor = F32(0.0)
        self.exact = exact_sums       # sequential f32 sums as Rust's iter().sum() (slow); default numpy pairwise
        self.out = []
        self.gains = []

    def process_chunk(self, chunk):
        self.buffer = np.concatenate([self.buffer, normalize_audio(chunk)])
        while len(self.buffer) >= FRAME:
            frame, self.buffer = self.buffer[:FRAME], self.buffer[FRAME:]
            if self.noise_floor == 0.0:
                self.noise_floor = estimate_noise_floor(frame)
            self.out.append(self.process_frame(frame))

    def process_frame(self, frame):
        with np.errstate(all="ignore"):
            p = preemphasis(frame)
            sq = (p * p).astype(F32)
            energy = F32((_seq_sum_f32(sq) if self.exact else np.sum(sq, dtype=F32)) / F32(len(frame)))
            threshold = F32(F32(self.noise_floor * F32(1.2)) + F32(self.prev_energy * F32(0.1)))
            gain = F32(1.0) if energy > threshold else np.fmax(F32(energy / threshold), F32(0.1))
            self.prev_energy = energy
            self.noise_floor = F32(F32(self.noise_floor * F32(0.95)) + F32(np.fmin(energy, self.noise_floor) * F32(0.05)))
            self.gains.append(float(gain))
            processed = (np.asarray(frame, F32) * F32(gain)).astype(F32)
            if self.config.enable_noise_reduction:
                processed = dn.denoise_audio(processed, self.config)[0]
            return dn.apply_noise_gate(processed, self.config.noise_gate)

    def finish(self):
        if len(self.buffer):
            frame = np.concatenate([self.buffer, np.zeros(FRAME - len(self.buffer), F32)])
            self.buffer = np.zeros(0, F32)
            self.out.append(self.process_frame(frame))


def preprocess_stream(mono16k: np.ndarray, chunk_len: int = READ_CHUNK, config=None, exact_sums=False):
    """parse_audio_file_stream for a 16 kHz stream already mixed to mono: returns (frames [n_frames, 2048], gains)."""
    p = StreamAudioProcessor(config, exact_sums)
    x = np.asarray(mono16k, F32)
    for i in range(0, len(x), chunk_len):
        p.process_chunk(x[i : i + chunk_len])
    p.finish()
    return (np.stack(p.out) if p.out else np.zeros((0, FRAME), F32)), p.gains


def rest_chunks(frames: np.ndarray):
    """transcribe.rs:100-142: append 2048-sample callbacks until >= 480 000 samples, hand that buffer over, clear; flush the remainder."""
    out, buf = [], []
    n = 0
    for f in frames:
        buf.append(f); n += len(f)
        if n >= BUFFER_SIZE:
            out.append(np.concatenate(buf)); buf, n = [], 0
    if buf:
        out.append(np.concatenate(buf))
    return out
