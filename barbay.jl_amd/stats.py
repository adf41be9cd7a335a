"""Host-side mirror of the two `BarBay.stats` functions that feed the hot path's priors:

    naive_fitness    src/stats.jl:1040-1106
    naive_prior      src/stats.jl:1175-1359

Both are one-shot array passes over the tidy frame; their outputs (`s_pop_prior`, `logσ_pop_prior`, `logλ_prior`
means) are what `docs/src/examples.md:122-140` stacks with a chosen std into the matrix-form priors that
`bb_model_desc` takes per element.  The posterior-predictive helpers of src/stats.jl are outside SURVEY.md §8.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import pandas as pd

from . import utils
from .utils import _neutral_mask


def naive_fitness(data: pd.DataFrame, *, id_col="barcode", time_col="time", count_col="count", neutral_col="neutral",
                  pseudocount: int = 1) -> pd.DataFrame:
    """Mean over time of log(f_{t+1}/f_t) of a mutant minus the neutrals' mean of the same (src/stats.jl:1040-1106).
    Frequencies use per-time totals of the pseudocounted counts (:1053-1060); the first time point of every barcode
    has no ratio (:1075-1080); rows come back in the frame's barcode order (groupby, first appearance)."""
    d = data[[id_col, time_col, count_col, neutral_col]]
    cnt = d[count_col].to_numpy(dtype=np.float64) + pseudocount
    tr = utils._time_rank(d, time_col)
    freq = cnt / np.bincount(tr, weights=cnt)[tr]
    codes, ids = pd.factorize(d[id_col], sort=False)
    order = np.lexsort((np.arange(len(d)), codes))                # rows of a barcode in frame order (the reference does not sort here)
    c, t, lf = codes[order], tr[order], np.log(freq[order])
    same = c[1:] == c[:-1]
    logf, t2, c2 = (lf[1:] - lf[:-1])[same], t[1:][same], c[1:][same]
    first = np.zeros(len(ids), dtype=np.int64)
    first[codes[::-1]] = np.arange(len(d))[::-1]
    neu = _neutral_mask(d, neutral_col)[first]                    # first(d[:, neutral_col]) per barcode (:1082)
    n2 = neu[c2]
    nt = int(tr.max()) + 1
    st = np.bincount(t2[n2], weights=logf[n2], minlength=nt) / np.maximum(np.bincount(t2[n2], minlength=nt), 1)
    norm = logf - st[t2]
    keep = ~n2
    fit = np.bincount(c2[keep], weights=norm[keep], minlength=len(ids)) / np.maximum(np.bincount(c2[keep], minlength=len(ids)), 1)
    mut = ~neu
    return pd.DataFrame({id_col: np.asarray(ids)[mut], "fitness": fit[mut]})


def _finite_mean_std(x: np.ndarray):
    """Row-wise mean and corrected std over the finite entries (`x[.!isinf.(x)]`, src/stats.jl:1264-1266, :1300-1302)."""
    ok = ~np.isinf(x)
    n = ok.sum(axis=1)
    xs = np.where(ok, x, 0.0)
    mean = xs.sum(axis=1) / n
    var = (np.where(ok, x - mean[:, None], 0.0) ** 2).sum(axis=1) / (n - 1)
    return mean, np.sqrt(var)


def naive_prior(data: pd.DataFrame, *, id_col="barcode", time_col="time", count_col="count", neutral_col="neutral",
                rep_col: Optional[str] = None, pseudocount: int = 1) -> Dict[str, np.ndarray]:
    """Empirical prior means from the neutral lineages (src/stats.jl:1175-1359):
        s_pop_prior    = -mean_b log(f_{t+1,b}/f_{t,b})   per time step (and replicate, replicate-major)
        logσ_pop_prior = -std_b  log(f_{t+1,b}/f_{t,b})   (sic: minus the std, :1338)
        logλ_prior     = log(counts + pseudocount), time-fastest per barcode (and replicate-major)  (:1347-1352)
    Unlike the reference (:1187) the caller's frame is left untouched."""
    d = data.copy()
    d[count_col] = d[count_col] + pseudocount
    arr = utils.data_to_arrays(d, id_col=id_col, time_col=time_col, count_col=count_col, neutral_col=neutral_col, rep_col=rep_col)
    if isinstance(arr.bc_count, list):                                             # uneven replicates (:1237-1258)
        mats, tots = arr.bc_count, arr.bc_total
    elif arr.bc_count.ndim == 3:                                                   # T x B x R (:1211-1236)
        mats = [arr.bc_count[:, :, r] for r in range(arr.bc_count.shape[2])]
        tots = [arr.bc_total[:, r] for r in range(arr.bc_count.shape[2])]
    else:
        mats, tots = [arr.bc_count], [arr.bc_total]
    s_pop, ls_pop, loglam = [], [], []
    for R, n in zip(mats, tots):
        f = R[:, :arr.n_neutral] / n[:, None]
        with np.errstate(divide="ignore", invalid="ignore"):
            lr = np.log(f[1:] / f[:-1])
        m, sd = _finite_mean_std(lr)
        s_pop.append(-m)
        ls_pop.append(-sd)
        with np.errstate(divide="ignore"):
            loglam.append(np.log(R.astype(np.float64)).T.reshape(-1))              # column-major `[:]`
    return {"s_pop_prior": np.concatenate(s_pop), "logσ_pop_prior": np.concatenate(ls_pop),
            "logλ_prior": np.concatenate(loglam)}
