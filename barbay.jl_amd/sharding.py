"""Barcode sharding helpers (host side): which flat-latent indices a barcode range owns, and
assembling the full variational parameter vector from per-rank copies.

Partition: rank r owns barcodes [B r / W, B (r+1) / W) -- all time points, all
replicates and every per-mutant latent of those barcodes.  The global blocks (s_pop, logsigma_pop) are
replicated and updated identically on every rank.  Genotype model: when the mutants come grouped by genotype
(geno_idx non-decreasing) the engine moves the cuts to genotype boundaries and rank r owns theta of the genotypes
[stats.geno_lo, stats.geno_hi) -- after a resident run without a communicator only the owner's copy is current;
otherwise theta is replicated (every rank reports [0, n_geno)).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def shard_range(B: int, rank: int, world_size: int) -> Tuple[int, int]:
    return B * rank // world_size, B * (rank + 1) // world_size


def owned_indices(kind: str, layout: Dict[str, Tuple[int, int]], b_lo: int, b_hi: int, n_neutral: int, n_bc: int,
                  n_time: Sequence[int], n_rep: int = 1, n_env: int = 1, geno_range: Tuple[int, int] | None = None) -> np.ndarray:
    """Flat indices (reference order) of the latents owned by barcodes [b_lo, b_hi) (+ theta of the genotypes geno_range)."""
    B = n_neutral + n_bc
    m_lo, m_hi = max(b_lo, n_neutral) - n_neutral, max(b_hi, n_neutral) - n_neutral
    idx: List[np.ndarray] = []
    lo = layout["loglambda"][0]
    for r in range(n_rep):
        T = n_time[r]
        idx.append(np.arange(lo + b_lo * T, lo + b_hi * T))
        lo += T * B
    m = np.arange(m_lo, m_hi)
    if kind in ("fitness", "multienv"):
        E = n_env if kind == "multienv" else 1
        me = np.arange(m_lo * E, m_hi * E)
        idx += [layout["s_bc"][0] + me, layout["logsigma_bc"][0] + me]
    elif kind == "genotype":
        idx += [layout[k][0] + m for k in ("theta_tilde", "logtau", "logsigma_bc")]
        if geno_range is not None:
            idx.append(layout["theta"][0] + np.arange(geno_range[0], geno_range[1]))
    elif kind == "replicate":
        idx.append(layout["theta"][0] + m)
        for r in range(n_rep):
            idx += [layout[k][0] + r * n_bc + m for k in ("theta_tilde", "logtau", "logsigma_bc")]
    else:   # multienv_replicate: theta[e, m]; tt / lt / ls [e, m, r], environment fastest
        me = np.arange(m_lo * n_env, m_hi * n_env)
        idx.append(layout["theta"][0] + me)
        for r in range(n_rep):
            idx += [layout[k][0] + r * n_bc * n_env + me for k in ("theta_tilde", "logtau", "logsigma_bc")]
    return np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)


def gather_params(per_rank: Sequence[np.ndarray], stats: Sequence[dict], kind: str, layout: Dict[str, Tuple[int, int]],
                  n_neutral: int, n_bc: int, n_time: Sequence[int], n_rep: int = 1, n_env: int = 1,
                  owned: Sequence[np.ndarray] | None = None) -> np.ndarray:
    """Full flat vector from per-rank vectors: replicated blocks from rank 0, owned entries from their rank.  `owned[r]` = rank r's
    `engine.owned()` (the library's own word, in the caller's order -- needed whenever the handle's internal order differs from the
    caller's: genotype model); None: the ranges are worked out here from the stats' shard / genotype ranges (identity order)."""
    out = np.array(per_rank[0], copy=True)
    for r, (v, st) in enumerate(zip(per_rank, stats)):
        if owned is not None:
            ix = np.asarray(owned[r])
        else:
            ix = owned_indices(kind, layout, int(st["shard_lo"]), int(st["shard_hi"]), n_neutral, n_bc, n_time, n_rep, n_env,
                               geno_range=(int(st["geno_lo"]), int(st["geno_hi"])) if kind == "genotype" else None)
        out[ix] = v[ix]
    return out
