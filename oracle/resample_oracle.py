"""ORACLE (test infrastructure only): numpy restatement of the sinc resampler the reference's front end uses.

Reference call sites: /root/reference/src/audio/mod.rs:235-257 (create_resampler: SincFixedIn::<f32>::new(16000/from, 2.0,
{sinc_len 256, f_cutoff 0.95, Linear, oversampling 256, BlackmanHarris2}, chunk 4096, 1 channel); resample_chunk = process(&[chunk], None)).
The arithmetic lives in the crate `rubato` 0.16.0 (Cargo.lock:2405-2408), which is NOT in this container: this file restates the crate's
published algorithm (sinc table: windowed sinc of sinc_len*oversampling points, normalised to unit DC gain per phase, reversed phase order;
asynchronous fixed-input resampling: 2*sinc_len samples of history, first read position -sinc_len/2, output instants advanced by 1/ratio until
chunk - (sinc_len+1) - ceil(1/ratio), nearest two sub-phases blended linearly).  No vectors from the reference or the crate: parity unpinned.
`process` needs exactly `chunk_size` samples; anything else is rubato's "insufficient input" error, which is what the reference hits on the
last partial read chunk and on every chunk of a multi-channel file (mono chunks are 4096/channels long).
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


def blackman_harris2(npoints: int) -> np.ndarray:
    x = np.arange(npoints, dtype=F32)
    x_pi = (x * F32(2.0 * math.pi) / F32(npoints)).astype(F32)
    w = (F32(0.35875) - F32(0.48829) * np.cos(x_pi, dtype=F32) + F32(0.14128) * np.cos(F32(2.0) * x_pi, dtype=F32)
         - F32(0.01168) * np.cos(F32(3.0) * x_pi, dtype=F32)).astype(F32)
    return (w * w).astype(F32)


def _sinc(v: np.ndarray) -> np.ndarray:
    out = np.ones_like(v, dtype=F32)
    nz = v != 0
    a = (v[nz] * F32(math.pi)).astype(F32)
    out[nz] = (np.sin(a, dtype=F32) / a).astype(F32)
    return out


def make_sincs(npoints: int, factor: int, f_cutoff: float) -> np.ndarray:
    tot = npoints * factor
    w = blackman_harris2(tot)
    x = np.arange(tot, dtype=F32)
    y = (w * _sinc(((x - F32(tot // 2)) * F32(f_cutoff) / F32(factor)).astype(F32))).astype(F32)
    s = F32(0.0)
    for v in y:                      # sequential f32 sum, as the crate accumulates while it builds the table
        s = F32(s + v)
    s = F32(s / F32(factor))
    sincs = np.zeros((factor, npoints), F32)
    for n in range(factor):
        sincs[factor - n - 1, :] = (y[n::factor] / s).astype(F32)
    return sincs


def dot8(wave: np.ndarray, sinc: np.ndarray) -> np.float32:
    """8 running sums over stride-8 lanes, then acc0+...+acc7 (the scalar interpolator's unrolled loop)."""
    p = (wave.reshape(-1, 8) * sinc.reshape(-1, 8)).astype(F32)
    acc = np.zeros(8, F32)
    for row in p:
        acc = (acc + row).astype(F32)
    t = F32(0.0)
    for a in acc:
        t = F32(t + a)
    return t


class SincFixedIn:
    def __init__(self, ratio: float, chunk_size: int = 4096, sinc_len: int = 256, f_cutoff: float = 0.95, oversampling: int = 256):
        self.ratio = float(ratio)
        self.chunk = chunk_size
        self.sinc_len = 8 * int(math.ceil(sinc_len / 8.0))
        self.over = oversampling
        cutoff = F32(f_cutoff) if ratio >= 1.0 else F32(F32(f_cutoff) * F32(ratio))
        self.sincs = make_sincs(self.sinc_len, oversampling, float(cutoff))
        self.buffer = np.zeros(chunk_size + 2 * self.sinc_len, F32)
        self.last_index = -float(self.sinc_len // 2)

    def positions(self):
        """The output instants of the next chunk (relative to its first sample) and the carried-over start of the one after."""
        t_ratio = 1.0 / self.ratio
        end_idx = self.chunk - (self.sinc_len + 1) - int(math.ceil(t_ratio))
        idx, out = self.last_index, []
        while idx < float(end_idx):
            idx += t_ratio
            out.append(idx)
        return np.array(out, np.float64), idx - float(self.chunk)

    def process(self, chunk: np.ndarray) -> np.ndarray:
        chunk = np.asarray(chunk, F32)
        if len(chunk) != self.chunk:
            raise ValueError(f"Insufficient buffer size {len(chunk)} for input channel 0, expected {self.chunk}")
        L = self.sinc_len
        self.buffer[: 2 * L] = self.buffer[self.chunk : self.chunk + 2 * L]
        self.buffer[2 * L :] = chunk
        pos, self.last_index = self.positions()
        out = np.zeros(len(pos), F32)
        for k, idx in enumerate(pos):
            fl = math.floor(idx)
            sub = int(math.floor((idx - fl) * self.over))
            i0, s0 = fl, sub
            s1, i1 = sub + 1, fl
            if s1 >= self.over:
                s1 -= self.over; i1 += 1
            frac = F32(idx * self.over - math.floor(idx * self.over))
            p0 = dot8(self.buffer[i0 + 2 * L : i0 + 3 * L], self.sincs[s0])
            p1 = dot8(self.buffer[i1 + 2 * L : i1 + 3 * L], self.sincs[s1])
            out[k] = F32(p0 + F32(frac * F32(p1 - p0)))
        return out


def resample_stream(mono: np.ndarray, from_rate: int):
    """What parse_audio_file_stream feeds the pre-processor for a MONO file at `from_rate` (mod.rs:171-217): one resampled chunk per full
    4096-sample read; the first short read ends processing with an error (the remainder is lost).  Returns (chunks, tail_dropped)."""
    r = SincFixedIn(16000.0 / from_rate)
    x = np.asarray(mono, F32)
    out = []
    for i in range(0, len(x) - r.chunk + 1, r.chunk):
        out.append(r.process(x[i : i + r.chunk]))
    return out, (len(x) % r.chunk) != 0
