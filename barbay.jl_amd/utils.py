"""Host-side mirror of `BarBay.utils` either side of the hot path (pandas in place of DataFrames.jl):

    data_to_arrays   src/utils.jl:996-1033 (+ the `_extract_R` methods :409-920 and their helpers :98-382)
    advi_to_df       src/utils.jl:1409-1462 (+ helpers :1042-1343)

In the Julia integration these stay as they are; they are mirrored here so that the
Python harness exposes the same `vi.advi(data=...)` surface and its tests read like test/vi_tests.jl.
"""
from __future__ import annotations

from dataclasses import dataclass
from itertools import chain
from typing import Any, List, Optional, Sequence, Union

import numpy as np
import pandas as pd

from .model import BarBayError


@dataclass
class DataArrays:
    """src/utils.jl:48-61."""
    bc_count: Union[np.ndarray, List[np.ndarray]]     # T x B | T x B x R | list of T_r x B
    bc_total: Union[np.ndarray, List[np.ndarray]]     # T | T x R | list of T_r
    n_neutral: int
    n_bc: int
    bc_ids: list
    neutral_ids: list
    envs: Any                                          # "env1" | list | list of lists
    n_env: int
    n_rep: int
    n_time: Union[int, List[int]]
    genotypes: Any                                     # "N/A" | list
    n_geno: int


def _neutral_mask(data: pd.DataFrame, neutral_col: str) -> np.ndarray:
    col = data[neutral_col]
    if col.dtype == bool:
        return col.to_numpy()
    return col.astype(str).str.lower().isin(["true", "1"]).to_numpy()


def _unique(seq) -> list:
    return list(dict.fromkeys(seq))


def _time_rank(df: pd.DataFrame, time_col) -> np.ndarray:
    """Rank of every row's time point among the frame's sorted unique time points."""
    return pd.Categorical(df[time_col], categories=sorted(df[time_col].unique()), ordered=True).codes


def _group_matrix(df: pd.DataFrame, id_col, time_col, count_col, n_time: int, what: str, rep=None):
    """One column per barcode in groupby (first-appearance) order, rows sorted by time
    (_process_*_barcodes_single, src/utils.jl:98-173).  One stable sort over (barcode, time) instead of the
    reference's per-barcode sub-frames (those scans dominate at 10^5 barcodes)."""
    codes, ids = pd.factorize(df[id_col], sort=False)
    if len(ids) and (np.bincount(codes, minlength=len(ids)) != n_time).any():
        where = "" if rep is None else f" for replicate {rep}"
        raise BarBayError(f"Not all {what} barcodes have reported counts in all time points{where}.")
    order = np.lexsort((_time_rank(df, time_col), codes))          # stable: ties keep the frame's row order
    M = df[count_col].to_numpy(dtype=np.int64)[order].reshape(len(ids), n_time).T
    return np.ascontiguousarray(M), list(ids)


def _multi_tensor(df: pd.DataFrame, ids, reps, id_col, time_col, count_col, rep_col, n_time: int):
    """_process_*_barcodes_multi (src/utils.jl:187-266): T x n_ids x n_rep (the reference scans the frame once per
    (barcode, replicate): O(ids x reps x rows); here one sort)."""
    ci = pd.Categorical(df[id_col], categories=ids).codes.astype(np.int64)
    cr = pd.Categorical(df[rep_col], categories=reps).codes.astype(np.int64)
    cell = cr * len(ids) + ci
    if (ci < 0).any() or (cr < 0).any() or (np.bincount(cell, minlength=len(ids) * len(reps)) != n_time).any():
        raise BarBayError("Not all barcodes have reported counts in all time points of all replicates.")
    order = np.lexsort((_time_rank(df, time_col), cell))
    out = df[count_col].to_numpy(dtype=np.int64)[order].reshape(len(reps), len(ids), n_time)
    return np.ascontiguousarray(out.transpose(2, 1, 0))


def data_to_arrays(data: pd.DataFrame, *, id_col="barcode", time_col="time", count_col="count", neutral_col="neutral",
                   rep_col: Optional[str] = None, env_col: Optional[str] = None,
                   genotype_col: Optional[str] = None, group_genotypes: bool = False) -> DataArrays:
    """Tidy frame -> model inputs; neutrals first then mutants (src/utils.jl:431), totals = row sums (:432).

    group_genotypes (not in the reference): order the mutants by genotype (stable, genotypes in order of first appearance), so
    that a genotype's barcodes are consecutive.  The results are keyed by barcode id, so the order is the caller's to choose;
    with it the engine's resident launch and its genotype-aligned sharding apply to genotype_fitness_normal."""
    neu = _neutral_mask(data, neutral_col)
    timepoints = sorted(data[time_col].unique())                                   # _extract_timepoints :81-89
    if rep_col is None:                                                            # :409-448
        Rn, neutral_ids = _group_matrix(data[neu], id_col, time_col, count_col, len(timepoints), "neutral")
        Rm, bc_ids = _group_matrix(data[~neu], id_col, time_col, count_col, len(timepoints), "mutant")
        R = np.concatenate([Rn, Rm], axis=1)
        n_t: Any = R.sum(axis=1)
        n_rep, n_time = 1, len(timepoints)
    else:                                                                          # :478-534
        rep_groups = [(r, g) for r, g in data.groupby(rep_col, sort=False)]
        n_rep = len(rep_groups)
        n_rep_time = [g[time_col].nunique() for _, g in rep_groups]
        if len(set(n_rep_time)) == 1:
            dn, dm = data[neu], data[~neu]
            neutral_ids = _unique(dn[id_col].tolist())                             # unique            :198
            reps_n = _unique(dn[rep_col].tolist())
            bc_ids = sorted(_unique(dm[id_col].tolist()))                          # sort(unique(..))  :242
            reps_m = sorted(_unique(dm[rep_col].tolist()))                         #                   :244
            Rn = _multi_tensor(dn, neutral_ids, reps_n, id_col, time_col, count_col, rep_col, len(timepoints))
            Rm = _multi_tensor(dm, bc_ids, reps_m, id_col, time_col, count_col, rep_col, len(timepoints))
            R = np.concatenate([Rn, Rm], axis=1)
            n_t = R.sum(axis=1)                                                    # T x n_rep
        else:
            Rn_l, Rm_l = [], []
            neutral_ids, bc_ids = [], []
            for k, (r, g) in enumerate(rep_groups):                                # :274-381
                gneu = _neutral_mask(g, neutral_col)
                a, ids_n = _group_matrix(g[gneu], id_col, time_col, count_col, n_rep_time[k], "neutral", k + 1)
                b, ids_m = _group_matrix(g[~gneu], id_col, time_col, count_col, n_rep_time[k], "mutant", k + 1)
                if k == 0:
                    neutral_ids, bc_ids = ids_n, ids_m
                Rn_l.append(a)
                Rm_l.append(b)
            R = [np.concatenate([a, b], axis=1) for a, b in zip(Rn_l, Rm_l)]
            n_t = [m.sum(axis=1) for m in R]
        n_time = n_rep_time
    envs: Any = "env1"
    n_env = 1
    if env_col is not None:
        def env_list(df):
            u = df[[time_col, env_col]].drop_duplicates().sort_values(time_col, kind="stable")
            return u[env_col].tolist()
        if rep_col is None:                                                        # :559-596
            envs = env_list(data)
            n_env = len(set(envs))
        else:                                                                      # :622-667
            per = [env_list(g) for _, g in data.groupby(rep_col, sort=False)]
            n_env = len(set(sum(per, [])))
            envs = per[0] if all(p == per[0] for p in per) else per
    genotypes: Any = "N/A"
    n_geno = 0
    if genotype_col is not None:                                                   # :692-731
        m = dict(zip(data[id_col], data[genotype_col]))
        genotypes = [m[b] for b in bc_ids]
        n_geno = len(set(genotypes))
        if group_genotypes:
            if rep_col is not None:
                raise ValueError("group_genotypes applies to the single-replicate genotype model")
            first = {}
            for gname in genotypes:
                first.setdefault(gname, len(first))
            order = np.argsort(np.asarray([first[gname] for gname in genotypes]), kind="stable")
            nn = len(neutral_ids)
            R = np.concatenate([R[:, :nn], R[:, nn:][:, order]], axis=1)
            bc_ids = [bc_ids[i] for i in order]
            genotypes = [genotypes[i] for i in order]
    return DataArrays(R, n_t, len(neutral_ids), len(bc_ids), list(bc_ids), list(neutral_ids), envs, n_env, n_rep,
                      n_time, genotypes, n_geno)


# ---------------------------------------------------------------------------------------------------
# result formatting
# ---------------------------------------------------------------------------------------------------
VARNAME_TO_VARTYPE = {                                                            # src/utils.jl:1069-1078
    "s̲ₜ": "pop_mean_fitness", "logσ̲ₜ": "pop_std", "s̲⁽ᵐ⁾": "bc_fitness", "logσ̲⁽ᵐ⁾": "bc_std",
    "θ̲⁽ᵐ⁾": "bc_hyperfitness", "θ̲̃⁽ᵐ⁾": "bc_noncenter", "logτ̲⁽ᵐ⁾": "bc_deviations", "logΛ̲̲": "log_poisson",
}


def _ntime(output: DataArrays, r: int) -> int:
    return output.n_time if isinstance(output.n_time, int) else output.n_time[r]


def advi_to_df(data: pd.DataFrame, dist, vars: Sequence[str], *, id_col="barcode", time_col="time", count_col="count",
               neutral_col="neutral", rep_col=None, env_col=None, genotype_col=None, n_samples: int = 10_000,
               rng: Optional[np.random.Generator] = None) -> pd.DataFrame:
    """src/utils.jl:1409-1462.  `dist` exposes dist.dist.m, dist.dist.σ and dist.transform.ranges_out
    (0-based half-open (lo, hi) pairs here)."""
    output = data_to_arrays(data, id_col=id_col, time_col=time_col, count_col=count_col, neutral_col=neutral_col,
                            rep_col=rep_col, env_col=env_col, genotype_col=genotype_col)
    var_groups = [v.replace("[1]", "") for v in vars if v.endswith("[1]")]         # extract_variable_info :1042-1051
    var_range = list(dist.transform.ranges_out)
    df = pd.DataFrame({"mean": np.asarray(dist.dist.m), "std": np.asarray(dist.dist.σ)})   # :1056-1064
    df["varname"] = list(vars)
    vartype = np.empty(len(df), dtype=object)
    for (lo, hi), g in zip(var_range, var_groups):                                 # :1083-1091
        vartype[lo:hi] = VARNAME_TO_VARTYPE[g]
    df["vartype"] = vartype
    n_bc, n_neutral, n_rep, n_env = output.n_bc, output.n_neutral, output.n_rep, output.n_env
    if rep_col is not None:                                                        # add_replicate_info! :1100-1159
        rep = np.empty(len(df), dtype=object)
        for (lo, hi), g in zip(var_range, var_groups):
            if "̲ₜ" in g:
                rep[lo:hi] = "R1" if n_rep == 1 else list(chain.from_iterable([f"R{r + 1}"] * (_ntime(output, r) - 1) for r in range(n_rep)))
            elif g == "θ̲⁽ᵐ⁾":
                rep[lo:hi] = "N/A"
            elif g == "logΛ̲̲":
                rep[lo:hi] = "R1" if n_rep == 1 else list(chain.from_iterable(
                    [f"R{r + 1}"] * ((n_bc + n_neutral) * _ntime(output, r)) for r in range(n_rep)))
            else:
                rep[lo:hi] = "R1" if n_rep == 1 else list(chain.from_iterable([f"R{r + 1}"] * (n_bc * n_env) for r in range(n_rep)))
        df[rep_col] = rep
    if env_col is not None:                                                        # add_environment_info! :1164-1187
        env = np.empty(len(df), dtype=object)
        for (lo, hi), g in zip(var_range, var_groups):
            if n_env == 1:
                env[lo:hi] = "env1"
            elif "̲ₜ" in g:
                # the reference assigns `output.envs[2:end]` (:1179), which only fits one replicate; with replicates
                # the block holds (T_r - 1) entries per replicate, so the per-replicate lists are concatenated
                per = output.envs if (output.envs and isinstance(output.envs[0], (list, tuple))) else [output.envs] * max(n_rep, 1)
                env[lo:hi] = list(chain.from_iterable(e[1:] for e in per))
            elif g != "logΛ̲̲":
                # the reference fills θ̲⁽ᵐ⁾ only and leaves the other blocks #undef (:1182-1184); the
                # per-environment blocks are stored env-fastest (model_multienv_fitness_normal.jl:271-272)
                flat = sum(output.envs, []) if (output.envs and isinstance(output.envs[0], (list, tuple))) else list(output.envs)
                uniq = list(dict.fromkeys(flat))
                env[lo:hi] = uniq * ((hi - lo) // len(uniq))
        df[env_col] = env
    ids = np.empty(len(df), dtype=object)                                          # add_barcode_info! :1192-1279
    for (lo, hi), g in zip(var_range, var_groups):
        if "̲ₜ" in g:
            ids[lo:hi] = "N/A"
        elif g == "θ̲⁽ᵐ⁾" and genotype_col is None:
            ids[lo:hi] = output.bc_ids if n_env == 1 else list(chain.from_iterable([b] * n_env for b in output.bc_ids))
        elif g == "θ̲⁽ᵐ⁾":
            ids[lo:hi] = list(dict.fromkeys(output.genotypes))
        elif g == "logΛ̲̲":
            all_ids = list(output.neutral_ids) + list(output.bc_ids)
            ids[lo:hi] = list(chain.from_iterable([b] * _ntime(output, r) for r in range(n_rep) for b in all_ids))
        elif n_env == 1:
            ids[lo:hi] = list(output.bc_ids) * n_rep
        else:
            ids[lo:hi] = list(chain.from_iterable([b] * n_env for _ in range(n_rep) for b in output.bc_ids))
    df["id"] = ids
    if len(var_groups) == 7 and (n_rep > 1 or genotype_col is not None):          # :1457-1459
        df = _process_hierarchical_samples(df, output, n_samples, rep_col, env_col, genotype_col, rng,
                                           device=getattr(dist, "hier", None))
    return df


def _process_hierarchical_samples(df, output, n_samples, rep_col, env_col, genotype_col, rng, device=None):
    """process_hierarchical_samples! (src/utils.jl:1284-1343): draws n_samples Normal samples per
    parameter and appends derived `bc_fitness` rows reported as MEDIAN (under the column `mean`) and std.
    For the genotype model theta is indexed per genotype; the reference's `hcat(repeat([θ_mat], n_rep)...)`
    only lines up when every mutant has its own hyper-fitness (the replicate model), so the genotype case
    gathers theta by the mutant's genotype instead (documented deviation).
    `device` = (n_samples, median, std) computed by the engine (`bb_hier_fitness`, csrc/bb_hier.h) is used when it
    was drawn with the requested n_samples; the numpy draws below serve a `dist` that did not come from the engine."""
    rng = rng or np.random.default_rng()
    th = df[df.vartype == "bc_hyperfitness"]
    ta = df[df.vartype == "bc_deviations"]
    tt = df[df.vartype == "bc_noncenter"]
    n_unit = len(ta)
    if genotype_col is not None:
        uniq = list(dict.fromkeys(output.genotypes))
        col_of = np.asarray([uniq.index(g) for g in output.genotypes] * output.n_rep)
    else:
        col_of = np.tile(np.arange(len(th)), output.n_rep)
    med = np.empty(n_unit)
    sd = np.empty(n_unit)
    chunk = max(1, 2_000_000 // max(n_samples, 1))
    th_m, th_s = th["mean"].to_numpy(), th["std"].to_numpy()
    ta_m, ta_s = ta["mean"].to_numpy(), ta["std"].to_numpy()
    tt_m, tt_s = tt["mean"].to_numpy(), tt["std"].to_numpy()
    use_dev = device is not None and device[0] == n_samples and len(device[1]) == n_unit
    if use_dev:
        med, sd = np.asarray(device[1]), np.asarray(device[2])
    for lo in range(0, 0 if use_dev else n_unit, chunk):
        hi = min(lo + chunk, n_unit)
        c = col_of[lo:hi]
        theta = rng.normal(th_m[c], th_s[c], (n_samples, hi - lo))
        tau = np.exp(rng.normal(ta_m[lo:hi], ta_s[lo:hi], (n_samples, hi - lo)))
        ttil = rng.normal(tt_m[lo:hi], tt_s[lo:hi], (n_samples, hi - lo))
        s = theta + tau * ttil
        med[lo:hi] = np.median(s, axis=0)
        sd[lo:hi] = s.std(axis=0, ddof=1)
    extra = pd.DataFrame({"mean": med, "std": sd,
                          "varname": [v.replace("logτ", "s") for v in ta["varname"]], "vartype": "bc_fitness"})
    for c in (rep_col, env_col):
        if c is not None and c in df.columns:
            extra[c] = ta[c].to_numpy()
    extra["id"] = ta["id"].to_numpy()
    return pd.concat([df, extra], ignore_index=True)
