"""Synthetic workload definitions shared by bench.py and the tests.

Everything here is a closed formula or a counter-based generator, so every rank
and every test derives identical inputs without exchanging data.
"""
from __future__ import annotations

import math

import numpy as np

SEED_BASE = 0x5EED0000
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, first_index: int, n: int) -> np.ndarray:
    """SplitMix64 outputs first_index .. first_index+n-1 of the stream `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(first_index + 1, first_index + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def samples(seed: int, first_index: int, n: int, dtype=np.float64) -> np.ndarray:
    """Scalar samples in [-1, 1): (u >> 40) * 2^-23 - 1, exact in f32 and f64."""
    u = splitmix64(seed, first_index, n)
    return ((u >> np.uint64(40)).astype(np.float64) * 2.0 ** -23 - 1.0).astype(dtype)


def line_seed(line_index: int) -> int:
    return SEED_BASE + line_index


def fir_lowpass_taps(ntaps: int = 256, fc: float = 0.25, f32_rounded: bool = False) -> np.ndarray:
    """Hamming-windowed sinc low-pass, cutoff fc (cycles/sample), unit DC gain."""
    k = np.arange(ntaps, dtype=np.float64)
    m = k - 0.5 * (ntaps - 1)
    h = 2.0 * fc * np.sinc(2.0 * fc * m)
    if ntaps > 1:
        h *= 0.54 - 0.46 * np.cos(2.0 * np.pi * k / (ntaps - 1))
    h /= h.sum()
    if f32_rounded:
        h = h.astype(np.float32).astype(np.float64)
    return h


def biquad_rbj_lowpass(fc: float = 1000.0, fs: float = 48000.0, q: float = 1.0 / math.sqrt(2.0)) -> np.ndarray:
    """RBJ cookbook low-pass as one section {b0,b1,b2,a1,a2} (a0 normalised)."""
    w0 = 2.0 * math.pi * fc / fs
    alpha = math.sin(w0) / (2.0 * q)
    cw = math.cos(w0)
    b0 = (1.0 - cw) / 2.0
    b1 = 1.0 - cw
    b2 = (1.0 - cw) / 2.0
    a0 = 1.0 + alpha
    a1 = -2.0 * cw
    a2 = 1.0 - alpha
    return np.array([[b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0]], dtype=np.float64)


def resampler_proto(up: int = 160, down: int = 147, taps_per_phase: int = 24, beta: float = 8.6) -> np.ndarray:
    """Kaiser-windowed sinc prototype of length up*taps_per_phase, DC gain `up`."""
    n = up * taps_per_phase
    k = np.arange(n, dtype=np.float64)
    m = k - 0.5 * (n - 1)
    fc = 0.5 / max(up, down)  # cycles/sample at the up-sampled rate
    h = 2.0 * fc * np.sinc(2.0 * fc * m) * np.kaiser(n, beta)
    h *= up / h.sum()
    return h
