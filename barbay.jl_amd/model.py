"""`BarBay.model.*` on the hot path: the same names, positional signature and keyword arguments as the
reference's `Turing.@model` functions (docs/src/contributing.md:9-63).  Calling one builds a plain
description of the model instance -- what `model(R, n_t, n_neutral, n_bc; kwargs...)` constructs at
src/vi.jl:172-178 -- that `vi.advi` hands to the HIP engine; no math happens here.

    fitness_normal            src/model_fitness_normal.jl:120-130
    multienv_fitness_normal   src/model_multienv_fitness_normal.jl:133-144
    genotype_fitness_normal   src/model_fitness_normal_hierarchical_genotypes.jl:151-163
    replicate_fitness_normal  src/model_fitness_normal_hierarchical_replicates.jl:145-156 (3-D array)
                              and :407-418 (Vector{Matrix}: replicates with different time points)
    multienv_replicate_fitness_normal
                              src/model_multienv_fitness_normal_hierarchical_replicates.jl:158-170 (3-D array)
                              and :449-461 (Vector{Matrix})

Prior keywords accept the reference's spellings (`logσ_pop_prior`, `logλ_prior`, `logτ_prior`, ...)
and ASCII aliases (`logsigma_pop_prior`, `loglambda_prior`, `logtau_prior`).  A prior is either the
Vector form `[mean, std]` or the Matrix form `N x 2` (model_fitness_normal.jl:125-129, 137-146).
`vi.advi` recognises model variants by substring of the function name, as the reference does
(src/vi.jl:111-118, 146, 159).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


class BarBayError(RuntimeError):
    """Python stand-in for Julia's ErrorException (`error(...)`, src/vi.jl:107, 112, 117)."""


_PRIOR_ALIASES = {
    "s_pop_prior": "s_pop_prior", "logσ_pop_prior": "logsigma_pop_prior", "logsigma_pop_prior": "logsigma_pop_prior",
    "s_bc_prior": "s_bc_prior", "logσ_bc_prior": "logsigma_bc_prior", "logsigma_bc_prior": "logsigma_bc_prior",
    "logλ_prior": "loglambda_prior", "loglambda_prior": "loglambda_prior",
    "logτ_prior": "logtau_prior", "logtau_prior": "logtau_prior",
}


def _first_appearance_index(values: Sequence) -> Tuple[np.ndarray, list]:
    """`indexin(x, unique(x))`, 0-based (model_multienv_fitness_normal.jl:151-155)."""
    uniq: Dict[object, int] = {}
    idx = []
    for v in values:
        if v not in uniq:
            uniq[v] = len(uniq)
        idx.append(uniq[v])
    return np.asarray(idx, dtype=np.int32), list(uniq)


def _prior(value, name: str) -> Tuple[np.ndarray, np.ndarray]:
    a = np.asarray(value, dtype=np.float64)
    if a.ndim == 1:
        if a.shape[0] != 2:
            raise BarBayError(f"{name}: the Vector form is [mean, std]")
        return a[:1].copy(), a[1:].copy()
    if a.ndim == 2 and a.shape[1] == 2:
        return np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1])
    raise BarBayError(f"{name}: expected [mean, std] or an N x 2 matrix")


@dataclass
class BayesModel:
    """A constructed model instance (the role of the DynamicPPL.Model at src/vi.jl:172)."""
    kind: str                       # fitness | multienv | genotype | replicate | multienv_replicate
    name: str
    counts: List[np.ndarray]        # per replicate, T_r x B
    totals: List[np.ndarray]
    n_neutral: int
    n_bc: int
    env_idx: Optional[np.ndarray] = None
    geno_idx: Optional[np.ndarray] = None
    priors: Dict[str, Tuple[np.ndarray, np.ndarray]] = field(default_factory=dict)
    ragged: bool = False

    # variable symbols of the `~` blocks in source order, as DynamicPPL names them (src/vi.jl:184-198)
    def var_symbols(self) -> List[str]:
        if self.kind in ("fitness", "multienv"):
            return ["s̲ₜ", "logσ̲ₜ", "s̲⁽ᵐ⁾", "logσ̲⁽ᵐ⁾", "logΛ̲̲"]
        return ["s̲ₜ", "logσ̲ₜ", "θ̲⁽ᵐ⁾", "θ̲̃⁽ᵐ⁾", "logτ̲⁽ᵐ⁾", "logσ̲⁽ᵐ⁾", "logΛ̲̲"]


def _split_kwargs(kwargs: dict, allowed_extra=()):
    pri = {}
    for k, v in kwargs.items():
        if k in _PRIOR_ALIASES:
            pri[_PRIOR_ALIASES[k]] = _prior(v, k)
        elif k not in allowed_extra:
            raise BarBayError(f"unknown keyword argument {k!r}")
    return pri


def _as_list(R, n_t):
    R = R if isinstance(R, (list, tuple)) else [R]
    n_t = n_t if isinstance(n_t, (list, tuple)) else [n_t]
    return [np.asarray(r, dtype=np.int64) for r in R], [np.asarray(n, dtype=np.int64) for n in n_t]


def fitness_normal(R, n_t, n_neutral: int, n_bc: int, **kwargs) -> BayesModel:
    """model_fitness_normal.jl:120-272: R is T x B (neutrals first), n_t its row sums."""
    c, t = _as_list(np.asarray(R), np.asarray(n_t))
    return BayesModel("fitness", "fitness_normal", c, t, int(n_neutral), int(n_bc), priors=_split_kwargs(kwargs))


def multienv_fitness_normal(R, n_t, n_neutral: int, n_bc: int, *, envs, **kwargs) -> BayesModel:
    """model_multienv_fitness_normal.jl:133-303: `envs` lists the environment of every time point."""
    c, t = _as_list(np.asarray(R), np.asarray(n_t))
    if len(t[0]) != len(envs):
        raise BarBayError("Number of time points must match list of of environments")   # :146-148
    idx, _ = _first_appearance_index(list(envs))
    return BayesModel("multienv", "multienv_fitness_normal", c, t, int(n_neutral), int(n_bc), env_idx=idx,
                      priors=_split_kwargs(kwargs))


def genotype_fitness_normal(R, n_t, n_neutral: int, n_bc: int, *, genotypes, **kwargs) -> BayesModel:
    """model_fitness_normal_hierarchical_genotypes.jl:151-330: `genotypes[m]` is mutant m's genotype."""
    c, t = _as_list(np.asarray(R), np.asarray(n_t))
    if int(n_bc) != len(genotypes):
        raise BarBayError("List of genotypes must match number of barcodes")             # :165-167
    idx, _ = _first_appearance_index(list(genotypes))
    return BayesModel("genotype", "genotype_fitness_normal", c, t, int(n_neutral), int(n_bc), geno_idx=idx,
                      priors=_split_kwargs(kwargs))


def replicate_fitness_normal(R, n_t, n_neutral: int, n_bc: int, **kwargs) -> BayesModel:
    """model_fitness_normal_hierarchical_replicates.jl: R is T x B x n_rep (:145-332) or a list of
    T_r x B matrices (:407-638); n_t is T x n_rep or a list of vectors.

    The ragged method pairs neutral data element (t, b) with population index (t + (T_r-1) b) div n_neutral
    (`repeat(.., inner=n_neutral)`, :599-605, against a time-fastest data vector :549).  When R is a
    list -- the call that dispatches to that method in the reference -- the engine reproduces this pairing as
    written (BB_FLAG_RAGGED_METHOD); the 3-D call evaluates the 3-D method's self-consistent pairing (:307-311)."""
    ragged = isinstance(R, (list, tuple))
    if ragged:
        c = [np.asarray(r, dtype=np.int64) for r in R]
        t = [np.asarray(n, dtype=np.int64) for n in n_t]
    else:
        R = np.asarray(R, dtype=np.int64)
        n_t = np.asarray(n_t, dtype=np.int64)
        if R.ndim != 3:
            raise BarBayError("replicate_fitness_normal expects a T x B x n_rep array or a list of matrices")
        c = [np.ascontiguousarray(R[:, :, r]) for r in range(R.shape[2])]
        t = [np.ascontiguousarray(n_t[:, r]) for r in range(R.shape[2])]
    return BayesModel("replicate", "replicate_fitness_normal", c, t, int(n_neutral), int(n_bc),
                      priors=_split_kwargs(kwargs), ragged=ragged)


def multienv_replicate_fitness_normal(R, n_t, n_neutral: int, n_bc: int, *, envs, **kwargs) -> BayesModel:
    """model_multienv_fitness_normal_hierarchical_replicates.jl: R is T x B x n_rep with one `envs` list (:158-363)
    or a list of T_r x B matrices with one env list per replicate (:449-687).  theta is n_env x n_bc (environment
    fastest), theta_tilde / logtau / logsigma_bc are n_env x n_bc x n_rep; time step t of replicate r uses the
    environment of t+1 (:339-352)."""
    ragged = isinstance(R, (list, tuple))
    if ragged:
        c = [np.asarray(r, dtype=np.int64) for r in R]
        t = [np.asarray(n, dtype=np.int64) for n in n_t]
        per = [list(e) for e in envs]
        if len(per) != len(c) or any(len(e) != len(tt) for e, tt in zip(per, t)):
            raise BarBayError("Number of time points must match list of of environments for all replicates")   # :462-464
    else:
        R = np.asarray(R, dtype=np.int64)
        n_t = np.asarray(n_t, dtype=np.int64)
        if R.ndim != 3:
            raise BarBayError("multienv_replicate_fitness_normal expects a T x B x n_rep array or a list of matrices")
        if n_t.shape[0] != len(envs):
            raise BarBayError("Number of time points must match list of of environments")                      # :172-174
        c = [np.ascontiguousarray(R[:, :, r]) for r in range(R.shape[2])]
        t = [np.ascontiguousarray(n_t[:, r]) for r in range(R.shape[2])]
        per = [list(envs)] * R.shape[2]
    flat, _ = _first_appearance_index(sum(per, []))         # indexin.(envs, Ref(unique(vcat(envs...)))) :466-472
    idx, o = [], 0
    for e in per:
        idx.append(flat[o:o + len(e)])
        o += len(e)
    return BayesModel("multienv_replicate", "multienv_replicate_fitness_normal", c, t, int(n_neutral), int(n_bc),
                      env_idx=idx, priors=_split_kwargs(kwargs), ragged=ragged)
