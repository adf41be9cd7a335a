"""Counter-based normal stream used by the engine, oracle-side restatement (numpy).
TEST INFRASTRUCTURE ONLY.

The reference draws from Julia's task-local Xoshiro (`randn`), which cannot be
matched bit for bit from another language; the stream below is this project's
own definition (DESIGN.md "RNG stream") and is what the HIP kernels implement:

  Philox4x32-10 (Salmon et al., SC'11; multipliers 0xD2511F53 / 0xCD9E8D57,
  Weyl key increments 0x9E3779B9 / 0xBB67AE85)
  key     = (seed & 0xffffffff, seed >> 32)
  counter = (q & 0xffffffff, q >> 32, step, stream)      q = latent_index // 2
  stream  = MC sample index s (0-based) for the per-step draws,
            0xFFFFFFFF for the mu_0 init draw, 0xFFFFFFFE for the omega_0 init draw
  u1 = (((o1<<32 | o0) >> 11) + 1) * 2^-53   in (0, 1]
  u2 =  ((o3<<32 | o2) >> 11)      * 2^-53   in [0, 1)
  r  = sqrt(-2 ln u1);  latent 2q gets r*cos(2 pi u2), latent 2q+1 gets r*sin(2 pi u2)

Indices are positions in the reference's flat latent vector, so draws do not
depend on how barcodes are sharded over GPUs.
"""
from __future__ import annotations

import numpy as np

STREAM_INIT_MU = 0xFFFFFFFF
STREAM_INIT_OMEGA = 0xFFFFFFFE

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised over counter words (uint64 arrays holding 32-bit values)."""
    c0 = np.asarray(c0, dtype=np.uint64)
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint64), c0.shape).copy()
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint64), c0.shape).copy()
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint64), c0.shape).copy()
    c0 = c0.copy()
    k0 &= 0xFFFFFFFF
    k1 &= 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n1 = lo1
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        n3 = lo0
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def normals(seed: int, step: int, stream: int, D: int, lo: int = 0, hi: int | None = None) -> np.ndarray:
    """eps[lo:hi] of the D-long draw for (seed, step, stream)."""
    hi = D if hi is None else hi
    q = np.arange(lo // 2, (hi + 1) // 2, dtype=np.uint64)
    o0, o1, o2, o3 = philox4x32_10(q & _MASK, q >> np.uint64(32), step & 0xFFFFFFFF,
                                   stream & 0xFFFFFFFF, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = (((o1 << np.uint64(32) | o0) >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53
    u2 = ((o3 << np.uint64(32) | o2) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    r = np.sqrt(-2.0 * np.log(u1))
    out = np.empty(2 * q.shape[0])
    out[0::2] = r * np.cos(2.0 * np.pi * u2)
    out[1::2] = r * np.sin(2.0 * np.pi * u2)
    start = lo - 2 * (lo // 2)
    return out[start:start + (hi - lo)]


def pairs(seed: int, q, step, stream: int):
    """(n0, n1) of arbitrary Philox counters (q, step, stream) -- vectorised over q / step."""
    q = np.asarray(q, dtype=np.uint64)
    step = np.broadcast_to(np.asarray(step, dtype=np.uint64), q.shape)
    o0, o1, o2, o3 = philox4x32_10(q & _MASK, q >> np.uint64(32), step & _MASK, stream & 0xFFFFFFFF,
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u1 = (((o1 << np.uint64(32) | o0) >> np.uint64(11)).astype(np.float64) + 1.0) * 2.0 ** -53
    u2 = ((o3 << np.uint64(32) | o2) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    r = np.sqrt(-2.0 * np.log(u1))
    return r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)


STREAM_HIER_THETA = 0xFFFFFFF0
STREAM_HIER_UNIT = 0xFFFFFFF1


def hier_fitness(seed: int, n_samples: int, theta_idx, m_th, s_th, m_lt, s_lt, m_tt, s_tt):
    """Oracle of bb_hier_fitness (barbay.jl_amd/csrc/bb_hier.h): per unit u, draws of
    theta[theta_idx[u]] + exp(logtau_u) * theta_tilde_u from the engine's hier streams; (median, corrected std)."""
    j = np.arange(n_samples, dtype=np.uint64)
    med, sd = np.empty(len(theta_idx)), np.empty(len(theta_idx))
    for u, ith in enumerate(theta_idx):
        a, b = pairs(seed, (np.uint64(ith) << np.uint64(20)) | (j >> np.uint64(1)), j & np.uint64(1), STREAM_HIER_THETA)
        n_th = np.where(j & np.uint64(1), b, a)
        n_lt, n_tt = pairs(seed, np.full(n_samples, u, dtype=np.uint64), j, STREAM_HIER_UNIT)
        s = (m_th[ith] + s_th[ith] * n_th) + np.exp(m_lt[u] + s_lt[u] * n_lt) * (m_tt[u] + s_tt[u] * n_tt)
        med[u] = np.median(s)
        sd[u] = s.std(ddof=1)
    return med, sd
