"""Seeded synthetic barcode-count workloads of the shapes BASELINE.json names.

Common recipe (numpy ``default_rng(seed)``): n_neutral = B/50; initial frequencies
f0 ~ LogNormal(0,1) normalised; mutant fitness s_b ~ U(0, 0.8), neutrals 0; per-step noise
N(0, 0.05); f_{t+1,b} ∝ f_{t,b} exp(s_b + noise); depth n_t = 200 B reads per time point;
R_t ~ Multinomial(n_t, f_t).  Returns plain arrays in the layout `utils.data_to_arrays`
(src/utils.jl:423-432) hands to the models: counts T x B, neutrals first.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np


@dataclass
class Workload:
    kind: str
    counts: List[np.ndarray]            # per replicate, T_r x B int64
    n_neutral: int
    n_bc: int
    env_idx: Optional[np.ndarray] = None
    geno_idx: Optional[np.ndarray] = None
    truth: Dict[str, np.ndarray] = field(default_factory=dict)
    name: str = ""

    @property
    def B(self) -> int:
        return self.n_neutral + self.n_bc


def _trajectory(g, f0, s_by_step, depth):
    """counts T x B for per-step fitness s_by_step[(T-1) x B]."""
    f = f0.copy()
    rows = [g.multinomial(depth, f)]
    for s in s_by_step:
        f = f * np.exp(s + g.normal(0.0, 0.05, f.shape[0]))
        f /= f.sum()
        rows.append(g.multinomial(depth, f))
    return np.stack(rows).astype(np.int64)


def _f0(g, B):
    f = g.lognormal(0.0, 1.0, B)
    return f / f.sum()


def fitness_normal(B: int = 50_000, T: int = 8, seed: int = 42, n_neutral: Optional[int] = None) -> Workload:
    """Config C2 (headline): fitness_normal, B x T."""
    g = np.random.default_rng(seed)
    nn = max(1, B // 50) if n_neutral is None else n_neutral
    s = np.concatenate([np.zeros(nn), g.uniform(0.0, 0.8, B - nn)])
    c = _trajectory(g, _f0(g, B), np.tile(s, (T - 1, 1)), 200 * B)
    return Workload("fitness", [c], nn, B - nn, truth={"s": s[nn:]}, name=f"fitness_normal {B}x{T} seed {seed}")


def replicate_fitness_normal(B: int = 20_000, T: int = 6, R: int = 3, seed: int = 43) -> Workload:
    """Config C3: theta_b ~ U(0, 0.8), s_{b,r} = theta_b + N(0, 0.05)."""
    g = np.random.default_rng(seed)
    nn = max(1, B // 50)
    theta = g.uniform(0.0, 0.8, B - nn)
    counts = []
    for _ in range(R):
        s = np.concatenate([np.zeros(nn), theta + g.normal(0.0, 0.05, B - nn)])
        counts.append(_trajectory(g, _f0(g, B), np.tile(s, (T - 1, 1)), 200 * B))
    return Workload("replicate", counts, nn, B - nn, truth={"theta": theta},
                    name=f"replicate_fitness_normal {B}x{T}x{R} seed {seed}")


def multienv_fitness_normal(B: int = 20_000, T: int = 6, envs=(1, 1, 2, 3, 4, 1), seed: int = 44) -> Workload:
    """Config C4: per-environment fitness s_{e,b} ~ U(-0.2, 0.8); step t uses the env of t+1."""
    g = np.random.default_rng(seed)
    nn = max(1, B // 50)
    envs = list(envs)
    assert len(envs) == T
    uniq = list(dict.fromkeys(envs))
    env_idx = np.asarray([uniq.index(e) for e in envs], dtype=np.int32)
    se = g.uniform(-0.2, 0.8, (len(uniq), B - nn))
    steps = [np.concatenate([np.zeros(nn), se[env_idx[t + 1]]]) for t in range(T - 1)]
    c = _trajectory(g, _f0(g, B), steps, 200 * B)
    return Workload("multienv", [c], nn, B - nn, env_idx=env_idx, truth={"s_env": se},
                    name=f"multienv_fitness_normal {B}x{T} envs {envs} seed {seed}")


def multienv_replicate_fitness_normal(B: int = 12_000, T=(6, 5, 6), envs=((1, 1, 2, 3, 1, 2), (1, 2, 3, 1, 2), (1, 3, 2, 1, 3, 2)),
                                      seed: int = 46) -> Workload:
    """The fifth model (no BASELINE config of its own): per-environment hyper-fitness theta_{e,b} ~ U(-0.2, 0.8),
    s_{e,b,r} = theta_{e,b} + N(0, 0.05); ragged replicates, each with its own environment sequence."""
    g = np.random.default_rng(seed)
    nn = max(1, B // 50)
    T = list(T)
    flat = [e for es in envs for e in es]
    uniq = list(dict.fromkeys(flat))                       # indexin.(envs, Ref(unique(vcat(envs...))))
    theta = g.uniform(-0.2, 0.8, (len(uniq), B - nn))
    counts, env_idx = [], []
    for r, Tr in enumerate(T):
        assert len(envs[r]) == Tr
        ei = np.asarray([uniq.index(e) for e in envs[r]], dtype=np.int32)
        sr = theta + g.normal(0.0, 0.05, theta.shape)
        steps = [np.concatenate([np.zeros(nn), sr[ei[t + 1]]]) for t in range(Tr - 1)]
        counts.append(_trajectory(g, _f0(g, B), steps, 200 * B))
        env_idx.append(ei)
    return Workload("multienv_replicate", counts, nn, B - nn, env_idx=env_idx, truth={"theta": theta},
                    name=f"multienv_replicate_fitness_normal {B}x{T} seed {seed}")


def genotype_fitness_normal(B: int = 200_000, T: int = 8, G: int = 5_000, seed: int = 45) -> Workload:
    """Config C5: barcodes dealt to genotypes in contiguous blocks, s_b = theta_g + N(0, 0.05)."""
    g = np.random.default_rng(seed)
    nn = max(1, B // 50)
    nb = B - nn
    geno = (np.arange(nb) * G // nb).astype(np.int32)
    theta = g.uniform(0.0, 0.8, G)
    s = np.concatenate([np.zeros(nn), theta[geno] + g.normal(0.0, 0.05, nb)])
    c = _trajectory(g, _f0(g, B), np.tile(s, (T - 1, 1)), 200 * B)
    return Workload("genotype", [c], nn, nb, geno_idx=geno, truth={"theta": theta},
                    name=f"genotype_fitness_normal {B}x{T} G {G} seed {seed}")
