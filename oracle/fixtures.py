"""Build oracle ModelSpecs from the reference's test CSVs (copied as data under
tests/golden/).  TEST INFRASTRUCTURE ONLY.

Array construction follows src/utils.jl `_extract_R`: neutrals first then mutants
(:423-431), per-barcode rows sorted by time (:131-136), totals = row sums (:432);
barcode order = first appearance for the single-replicate paths (groupby order,
:107-110), `unique` neutrals / `sort(unique)` mutants and reps for the 3-D path
(:198, :242-244); env per time point (:576-578); genotype per mutant (:709-713).
"""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import pandas as pd

from .spec import ModelSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _first_idx(values) -> np.ndarray:
    seen: Dict[object, int] = {}
    out = []
    for v in values:
        if v not in seen:
            seen[v] = len(seen)
        out.append(seen[v])
    return np.asarray(out, dtype=np.int64)


def _matrix(df: pd.DataFrame, ids, times) -> np.ndarray:
    piv = df.pivot(index="time", columns="barcode", values="count")
    return piv.loc[times, ids].to_numpy(dtype=np.int64)


def _neutral_col(df):
    return df["neutral"].astype(str).str.lower() == "true"


def load(name: str, **priors) -> ModelSpec:
    if name == "data001_single":
        df = pd.read_csv(os.path.join(GOLDEN, name + ".csv"))
        kind = "fitness"
    elif name == "data002_hier-rep":
        df = pd.read_csv(os.path.join(GOLDEN, name + ".csv"))
        kind = "replicate"
    elif name == "data003_multienv":
        df = pd.read_csv(os.path.join(GOLDEN, name + ".csv"))
        kind = "multienv"
    elif name == "data004_multigen":
        df = pd.read_csv(os.path.join(GOLDEN, name + ".csv"))
        kind = "genotype"
    else:
        raise KeyError(name)
    neu = _neutral_col(df)
    times = sorted(df["time"].unique())
    if kind == "replicate":
        n_ids = list(pd.unique(df.loc[neu, "barcode"]))
        m_ids = sorted(pd.unique(df.loc[~neu, "barcode"]))
        reps = sorted(pd.unique(df["rep"]))
        counts = [_matrix(df[df["rep"] == r], n_ids + m_ids, times) for r in reps]
    else:
        n_ids = list(pd.unique(df.loc[neu, "barcode"]))
        m_ids = list(pd.unique(df.loc[~neu, "barcode"]))
        counts = [_matrix(df, n_ids + m_ids, times)]
    totals = [c.sum(axis=1) for c in counts]
    kw = {}
    if kind == "multienv":
        env_by_time = df.drop_duplicates("time").set_index("time").loc[times, "env"].tolist()
        kw["env_idx"] = _first_idx(env_by_time)
    if kind == "genotype":
        g = df[~neu].drop_duplicates("barcode").set_index("barcode").loc[m_ids, "genotype"].tolist()
        kw["geno_idx"] = _first_idx(g)
    return ModelSpec(kind=kind, counts=counts, totals=totals, n_neutral=len(n_ids), n_bc=len(m_ids),
                     priors=priors, **kw)


def synthetic(kind: str, B: int, T, n_rep: int = 1, n_env: int = 1, n_geno: int = 0, seed: int = 0,
              n_neutral: int | None = None, depth_per_bc: int = 200, geno_runs: bool = False, **priors) -> ModelSpec:
    """Small seeded synthetic spec for oracle-vs-engine tests (numpy default_rng).  geno_runs: the genotype model's mutants
    come grouped by genotype (non-decreasing geno_idx, runs of random length), as a host that sorts its barcodes hands them over."""
    g = np.random.default_rng(seed)
    nn = max(1, B // 5) if n_neutral is None else n_neutral
    nb = B - nn
    Ts = list(T) if isinstance(T, (list, tuple)) else [T] * n_rep
    counts = []
    theta = g.uniform(0.0, 0.8, nb)
    for r in range(len(Ts)):
        s = np.concatenate([np.zeros(nn), theta + g.normal(0, 0.05, nb)])
        f = g.lognormal(0.0, 1.0, B)
        f /= f.sum()
        rows = []
        for t in range(Ts[r]):
            rows.append(g.multinomial(depth_per_bc * B, f))
            f = f * np.exp(s + g.normal(0, 0.05, B))
            f /= f.sum()
        counts.append(np.stack(rows).astype(np.int64))
    kw = {}
    if kind == "multienv":
        e = list(range(n_env)) + list(g.integers(0, n_env, max(0, Ts[0] - n_env)))
        kw["env_idx"] = _first_idx(e[:Ts[0]])
    if kind == "multienv_replicate":
        per = []
        for r in range(len(Ts)):
            e = (list(range(n_env)) if r == 0 else []) + list(g.integers(0, n_env, Ts[r]))
            per.append(e[:Ts[r]])
        flat = _first_idx(sum(per, []))           # indexin.(envs, Ref(unique(vcat(envs...))))
        kw["env_idx"], o = [], 0
        for r in range(len(Ts)):
            kw["env_idx"].append(flat[o:o + Ts[r]])
            o += Ts[r]
    if kind == "genotype":
        gi = list(g.integers(0, max(1, n_geno), nb))
        kw["geno_idx"] = _first_idx(sorted(gi) if geno_runs else gi)
    return ModelSpec(kind=kind, counts=counts, totals=[c.sum(axis=1) for c in counts], n_neutral=nn,
                     n_bc=nb, priors=priors, **kw)
