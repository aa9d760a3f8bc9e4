"""Counter-based synthetic data (no RNG state): element -> value by a 64-bit mix.

``value(i) = ((mix(i, seed) >> 40) * 2**-23 - 1) * scale`` -- uniform in ``[-scale, scale)``
on a 24-bit grid.  Everything before the final multiply is exact in fp32, and the multiply
is correctly rounded, so the HIP generator (``ffx_index_fill_synthetic``), the C oracle
(``ffo_synth_rows``) and this NumPy version agree bit for bit.  That lets ``bench.py`` fill
a 27 GB index on the device and still re-derive any row on the host for verification.
"""

from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 finaliser of ``idx + seed * golden`` (wrapping uint64 arithmetic)."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64(seed) * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def values(idx: np.ndarray, seed: int, scale: float = 1.0) -> np.ndarray:
    m = (mix64(idx, seed) >> np.uint64(40)).astype(np.float32)
    return (m * np.float32(2.0**-23) - np.float32(1.0)) * np.float32(scale)


def rows(seed: int, scale: float, first_row: int, n_rows: int, dim: int) -> np.ndarray:
    """Rows ``first_row .. first_row+n_rows`` of the synthetic ``(N, dim)`` matrix."""
    idx = np.arange(first_row * dim, (first_row + n_rows) * dim, dtype=np.uint64)
    return values(idx, seed, scale).reshape(n_rows, dim)


def uniform_u32(idx: np.ndarray, seed: int) -> np.ndarray:
    """32 uniform bits per counter (candidate sampling, sparse scores)."""
    return (mix64(idx, seed) >> np.uint64(32)).astype(np.uint32)
