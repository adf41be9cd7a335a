"""Model specification + flat latent layout, oracle side.  TEST INFRASTRUCTURE ONLY.

The flat latent vector ``z`` is the concatenation of the model's ``~`` blocks in
source order; Julia arrays are column-major.  Block orders follow

* fitness_normal            /root/reference/src/model_fitness_normal.jl:137-203
* multienv_fitness_normal   /root/reference/src/model_multienv_fitness_normal.jl:162-228
* genotype_fitness_normal   /root/reference/src/model_fitness_normal_hierarchical_genotypes.jl:181-258
* replicate_fitness_normal  /root/reference/src/model_fitness_normal_hierarchical_replicates.jl:164-243 (3-D)
                            and :451-530 (ragged Vector{Matrix})
* multienv_replicate_fitness_normal
                            /root/reference/src/model_multienv_fitness_normal_hierarchical_replicates.jl:190-288 (3-D)
                            and :497-581 (ragged)
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

KINDS = ("fitness", "multienv", "genotype", "replicate", "multienv_replicate")

# reference defaults: model_fitness_normal.jl:125-129, ..._genotypes.jl:162
DEFAULT_PRIORS = {
    "s_pop_prior": (0.0, 2.0),
    "logsigma_pop_prior": (0.0, 1.0),
    "s_bc_prior": (0.0, 2.0),
    "logsigma_bc_prior": (0.0, 1.0),
    "loglambda_prior": (3.0, 3.0),
    "logtau_prior": (-2.0, 1.0),
}


@dataclass
class ModelSpec:
    """Inputs available at the boundary (src/vi.jl:172-178, src/utils.jl:48-61).

    counts[r] is the T_r x B Int64 matrix of replicate r (neutrals first,
    utils.jl:431); totals[r] its row sums.  Non-replicate models have one
    entry.  env_idx / geno_idx are 0-based first-appearance indices
    (``indexin(x, unique(x))``, model_multienv_fitness_normal.jl:151-155,
    ..._genotypes.jl:170-174).
    """

    kind: str
    counts: List[np.ndarray]
    totals: List[np.ndarray]
    n_neutral: int
    n_bc: int
    env_idx: Optional[np.ndarray] = None
    geno_idx: Optional[np.ndarray] = None
    # name -> (mean, std); each a python float (Vector form) or 1-D array (Matrix form)
    priors: Dict[str, Tuple[object, object]] = field(default_factory=dict)

    def __post_init__(self):
        assert self.kind in KINDS, self.kind
        self.counts = [np.ascontiguousarray(c, dtype=np.int64) for c in self.counts]
        self.totals = [np.ascontiguousarray(t, dtype=np.int64) for t in self.totals]
        for c, t in zip(self.counts, self.totals):
            assert c.ndim == 2 and c.shape[1] == self.n_neutral + self.n_bc
            assert t.shape == (c.shape[0],)
        if self.kind not in ("replicate", "multienv_replicate"):
            assert len(self.counts) == 1
        p = dict(DEFAULT_PRIORS)
        p.update(self.priors)
        self.priors = p
        if self.kind == "multienv":
            assert self.env_idx is not None and len(self.env_idx) == self.n_time[0]
            self.env_idx = np.asarray(self.env_idx, dtype=np.int64)
        if self.kind == "multienv_replicate":
            # env_idx: one 0-based index list per replicate (first appearance over the concatenation,
            # `indexin.(envs, Ref(unique(vcat(envs...))))`, model_multienv_..._replicates.jl:466-472)
            assert self.env_idx is not None and len(self.env_idx) == len(self.counts)
            self.env_idx = [np.asarray(e, dtype=np.int64) for e in self.env_idx]
            assert all(len(e) == t for e, t in zip(self.env_idx, self.n_time))
        if self.kind == "genotype":
            assert self.geno_idx is not None and len(self.geno_idx) == self.n_bc
            self.geno_idx = np.asarray(self.geno_idx, dtype=np.int64)

    @property
    def n_rep(self) -> int:
        return len(self.counts)

    @property
    def n_time(self) -> List[int]:
        return [c.shape[0] for c in self.counts]

    @property
    def B(self) -> int:
        return self.n_neutral + self.n_bc

    @property
    def n_env(self) -> int:
        if self.env_idx is None:
            return 1
        if self.kind == "multienv_replicate":
            return int(max(e.max() for e in self.env_idx)) + 1
        return int(self.env_idx.max()) + 1

    @property
    def n_geno(self) -> int:
        return int(self.geno_idx.max()) + 1 if self.geno_idx is not None else 0

    # ---- flat layout -----------------------------------------------------
    def blocks(self) -> List[Tuple[str, int, str]]:
        """[(block symbol, length, prior name)] in source order."""
        nt1 = sum(t - 1 for t in self.n_time)
        nl = sum(t * self.B for t in self.n_time)
        R, E, G, nb = self.n_rep, self.n_env, self.n_geno, self.n_bc
        if self.kind == "fitness":
            return [("s_pop", nt1, "s_pop_prior"), ("logsigma_pop", nt1, "logsigma_pop_prior"),
                    ("s_bc", nb, "s_bc_prior"), ("logsigma_bc", nb, "logsigma_bc_prior"),
                    ("loglambda", nl, "loglambda_prior")]
        if self.kind == "multienv":
            return [("s_pop", nt1, "s_pop_prior"), ("logsigma_pop", nt1, "logsigma_pop_prior"),
                    ("s_bc", nb * E, "s_bc_prior"), ("logsigma_bc", nb * E, "logsigma_bc_prior"),
                    ("loglambda", nl, "loglambda_prior")]
        if self.kind == "genotype":
            return [("s_pop", nt1, "s_pop_prior"), ("logsigma_pop", nt1, "logsigma_pop_prior"),
                    ("theta", G, "s_bc_prior"), ("theta_tilde", nb, "_std_normal"),
                    ("logtau", nb, "logtau_prior"), ("logsigma_bc", nb, "logsigma_bc_prior"),
                    ("loglambda", nl, "loglambda_prior")]
        if self.kind == "multienv_replicate":
            return [("s_pop", nt1, "s_pop_prior"), ("logsigma_pop", nt1, "logsigma_pop_prior"),
                    ("theta", E * nb, "s_bc_prior"), ("theta_tilde", E * nb * R, "_std_normal"),
                    ("logtau", E * nb * R, "logtau_prior"), ("logsigma_bc", E * nb * R, "logsigma_bc_prior"),
                    ("loglambda", nl, "loglambda_prior")]
        return [("s_pop", nt1, "s_pop_prior"), ("logsigma_pop", nt1, "logsigma_pop_prior"),
                ("theta", nb, "s_bc_prior"), ("theta_tilde", nb * R, "_std_normal"),
                ("logtau", nb * R, "logtau_prior"), ("logsigma_bc", nb * R, "logsigma_bc_prior"),
                ("loglambda", nl, "loglambda_prior")]

    def offsets(self) -> Dict[str, Tuple[int, int]]:
        out, o = {}, 0
        for name, n, _ in self.blocks():
            out[name] = (o, o + n)
            o += n
        return out

    @property
    def D(self) -> int:
        return sum(n for _, n, _ in self.blocks())

    def prior_arrays(self) -> Tuple[np.ndarray, np.ndarray]:
        """Per-latent prior (mean, std), length D, fp64."""
        m = np.empty(self.D)
        s = np.empty(self.D)
        for (name, n, pname), (lo, hi) in zip(self.blocks(), self.offsets().values()):
            if pname == "_std_normal":
                pm, ps = 0.0, 1.0
            else:
                pm, ps = self.priors[pname]
            pm = np.asarray(pm, dtype=np.float64)
            ps = np.asarray(ps, dtype=np.float64)
            if pm.ndim == 1:
                assert pm.shape == (n,) and ps.shape == (n,), (name, pm.shape, n)
            m[lo:hi] = pm
            s[lo:hi] = ps
        return m, s
