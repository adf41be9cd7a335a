"""`BarBay.mcmc.mcmc_sample` on the device log-density (src/mcmc.jl:86-160).

The reference samples the Turing model with NUTS (`Turing.NUTS(0.65)`, `Turing.sample(model, sampler, ensemble,
n_steps, n_walkers)`) and saves the chain.  Here the sampler is a host-side NUTS (Hoffman & Gelman 2014, algorithm 6:
slice variant, dual-averaging step size, diagonal metric) whose every leapfrog asks the engine for
log p(data, z) and its gradient (`bb_logdensity_grad`: the same fused kernels as the ADVI step, draw pinned to z).
The metric and the start point come from a short ADVI run on the same handle (q's sigma^2 and mean): the
variational fit costs a few thousand device steps and spares NUTS its longest warm-up phase.

The reference's entry point is stale (it indexes the `data_to_arrays` result as a Dict and passes `rm_T0` / `verbose`
kwargs that function no longer has, src/mcmc.jl:120-129, 143-146); the argument list is kept, `rm_T0` is applied here.
Output: `<outputname>.npz` with `ids`, `var_names`, `chain` (n_walkers x n_steps x D), `logp` (the reference
writes `ids` and an MCMCChains object to `<outputname>.jld2`).
"""
from __future__ import annotations

import logging
import os
from typing import Callable, Dict, Optional

import numpy as np

from . import utils
from . import vi as _vi
from .model import BarBayError, BayesModel

log = logging.getLogger("barbay")


def _leapfrog(f, z, r, g, eps, minv):
    r = r + 0.5 * eps * g
    z = z + eps * minv * r
    lp, g = f(z)
    r = r + 0.5 * eps * g
    return z, r, lp, g


def _energy(lp, r, minv):
    h = lp - 0.5 * float(np.dot(r, minv * r))
    return h if np.isfinite(h) else -np.inf


def _find_step(f, z, lp, g, minv, rng):
    """Heuristic initial step size (Hoffman & Gelman, algorithm 4)."""
    eps = 1.0
    r = rng.standard_normal(z.shape[0]) / np.sqrt(minv)
    h0 = _energy(lp, r, minv)
    _, r1, lp1, _ = _leapfrog(f, z, r, g, eps, minv)
    a = 1.0 if _energy(lp1, r1, minv) - h0 > np.log(0.5) else -1.0
    for _ in range(60):
        _, r1, lp1, _ = _leapfrog(f, z, r, g, eps, minv)
        if a * (_energy(lp1, r1, minv) - h0) <= -a * np.log(2.0):
            break
        eps *= 2.0 ** a
    return eps


def _build_tree(f, z, r, g, logu, v, j, eps, h0, minv, rng):
    if j == 0:
        z1, r1, lp1, g1 = _leapfrog(f, z, r, g, v * eps, minv)
        h1 = _energy(lp1, r1, minv)
        n1 = int(logu <= h1)
        s1 = logu < 1000.0 + h1
        alpha = min(1.0, float(np.exp(min(0.0, h1 - h0)))) if np.isfinite(h1) else 0.0
        return z1, r1, g1, z1, r1, g1, z1, lp1, g1, n1, s1, alpha, 1
    zm, rm, gm, zp, rp, gp, z1, lp1, g1, n1, s1, a1, na1 = _build_tree(f, z, r, g, logu, v, j - 1, eps, h0, minv, rng)
    if s1:
        if v < 0:
            zm, rm, gm, _, _, _, z2, lp2, g2, n2, s2, a2, na2 = _build_tree(f, zm, rm, gm, logu, v, j - 1, eps, h0, minv, rng)
        else:
            _, _, _, zp, rp, gp, z2, lp2, g2, n2, s2, a2, na2 = _build_tree(f, zp, rp, gp, logu, v, j - 1, eps, h0, minv, rng)
        if n2 > 0 and rng.random() < n2 / max(n1 + n2, 1):
            z1, lp1, g1 = z2, lp2, g2
        dz = zp - zm
        s1 = s2 and float(np.dot(dz, minv * rm)) >= 0.0 and float(np.dot(dz, minv * rp)) >= 0.0
        n1 += n2
        a1 += a2
        na1 += na2
    return zm, rm, gm, zp, rp, gp, z1, lp1, g1, n1, s1, a1, na1


def nuts(f: Callable, z0: np.ndarray, n_steps: int, n_adapt: int, *, target_accept: float = 0.65,
         minv: Optional[np.ndarray] = None, rng: Optional[np.random.Generator] = None, max_depth: int = 10):
    """One NUTS chain on `f(z) -> (logp, grad)`.  Returns (chain[n_steps, D], logp[n_steps], info); the n_adapt
    warm-up draws (step-size dual averaging towards `target_accept`) are not part of the returned chain, as with
    `Turing.NUTS` (`discard_adapt = true`)."""
    rng = rng or np.random.default_rng()
    z = np.array(z0, dtype=np.float64)
    D = z.shape[0]
    minv = np.ones(D) if minv is None else np.asarray(minv, dtype=np.float64)
    lp, g = f(z)
    if not np.isfinite(lp):
        raise BarBayError("log density is not finite at the initial point")
    eps = _find_step(f, z, lp, g, minv, rng)
    mu, eps_bar, h_bar, gamma, t0, kappa = np.log(10.0 * eps), 1.0, 0.0, 0.05, 10.0, 0.75
    chain = np.empty((n_steps, D))
    lps = np.empty(n_steps)
    depths, n_grad = [], 0
    for m in range(1, n_adapt + n_steps + 1):
        r0 = rng.standard_normal(D) / np.sqrt(minv)
        h0 = _energy(lp, r0, minv)
        logu = h0 + np.log(rng.random())
        zm = zp = z
        rm = rp = r0
        gm = gp = g
        j, n, s = 0, 1, True
        alpha = n_alpha = 0
        while s and j < max_depth:
            v = -1 if rng.random() < 0.5 else 1
            if v < 0:
                zm, rm, gm, _, _, _, z1, lp1, g1, n1, s1, alpha, n_alpha = _build_tree(f, zm, rm, gm, logu, v, j, eps, h0, minv, rng)
            else:
                _, _, _, zp, rp, gp, z1, lp1, g1, n1, s1, alpha, n_alpha = _build_tree(f, zp, rp, gp, logu, v, j, eps, h0, minv, rng)
            if s1 and rng.random() < min(1.0, n1 / n):
                z, lp, g = z1, lp1, g1
            n += n1
            dz = zp - zm
            s = s1 and float(np.dot(dz, minv * rm)) >= 0.0 and float(np.dot(dz, minv * rp)) >= 0.0
            j += 1
            n_grad += n_alpha
        if m <= n_adapt:
            h_bar = (1.0 - 1.0 / (m + t0)) * h_bar + (target_accept - alpha / max(n_alpha, 1)) / (m + t0)
            eps = float(np.exp(mu - np.sqrt(m) / gamma * h_bar))
            w = m ** -kappa
            eps_bar = float(np.exp(w * np.log(eps) + (1.0 - w) * np.log(eps_bar)))
            if m == n_adapt:
                eps = eps_bar
        else:
            chain[m - n_adapt - 1] = z
            lps[m - n_adapt - 1] = lp
            depths.append(j)
    return chain, lps, {"step_size": eps, "mean_tree_depth": float(np.mean(depths)) if depths else 0.0, "n_grad": n_grad}


def mcmc_sample(*, data, n_walkers: int, n_steps: int, outputname: Optional[str], model: Callable,
                model_kwargs: Optional[Dict] = None, id_col="barcode", time_col="time", count_col="count",
                neutral_col="neutral", rep_col: Optional[str] = None, env_col: Optional[str] = None,
                genotype_col: Optional[str] = None, rm_T0: bool = False, target_accept: float = 0.65,
                n_adapt: Optional[int] = None, advi_steps: int = 3000, verbose: bool = True, seed: int = 0, device: int = 0,
                engine_kwargs: Optional[Dict] = None):
    """src/mcmc.jl:86-160.  `sampler = Turing.NUTS(0.65)` becomes `target_accept`; `ensemble` is serial (one device).
    `advi_steps` > 0 preconditions NUTS with a mean-field fit on the same handle (0: unit metric, prior-mean start)."""
    fname = None if outputname is None else f"{outputname}.npz"
    if fname is not None and os.path.isfile(fname):                                # :104-106
        raise BarBayError(f"{fname} was already processed")
    mname = getattr(model, "__name__", str(model))
    if "replicate" in mname and rep_col is None:                                   # :109-111
        raise BarBayError("Hierarchical models for experimental replicates require argument `:rep_col`")
    if "multienv" in mname and env_col is None:
        raise BarBayError("Models with multiple environments require argument `:env_col`")
    if verbose:
        log.info("Pre-processing data...")                                         # :115
    if rm_T0:                                                                      # documented kwarg of the reference
        data = data[data[time_col] != sorted(data[time_col].unique())[0]]
    arrays = utils.data_to_arrays(data, id_col=id_col, time_col=time_col, count_col=count_col, neutral_col=neutral_col,
                                  rep_col=rep_col, env_col=env_col, genotype_col=genotype_col)
    model_kwargs = dict(model_kwargs or {})
    if "multienv" in mname:
        model_kwargs = {"envs": arrays.envs, **model_kwargs}
    if "genotype" in mname:
        model_kwargs = {"genotypes": arrays.genotypes, **model_kwargs}
    bayes_model = model(arrays.bc_count, arrays.bc_total, arrays.n_neutral, arrays.n_bc, **model_kwargs)   # :138-144
    if not isinstance(bayes_model, BayesModel):
        raise BarBayError("model must be one of barbay model constructors (BarBay.model.*)")
    n_adapt = min(1000, n_steps // 2) if n_adapt is None else n_adapt              # Turing.NUTS default n_adapts
    if verbose:
        log.info("Sampling posterior...")                                          # :131-133
    rng = np.random.default_rng(seed)
    with _vi.make_engine(bayes_model, _vi.ADVI(1, max(advi_steps, 1)), _vi.TruncatedADAGrad(), seed, device,
                         **(engine_kwargs or {})) as e:
        ranges = [(lo, hi) for _, lo, hi in e.layout()]
        if advi_steps > 0:
            e.run(advi_steps)
            mean, sigma = e.posterior()
            minv = sigma ** 2
        else:
            mean, _ = e.posterior()
            mean, minv = np.zeros_like(mean), np.ones_like(mean)
        chains, lps, infos = [], [], []
        for w in range(n_walkers):
            z0 = mean + np.sqrt(minv) * rng.standard_normal(mean.shape[0]) * (0.1 if advi_steps > 0 else 0.0)
            c, lp, info = nuts(e.logdensity_grad, z0, n_steps, n_adapt, target_accept=target_accept, minv=minv, rng=rng)
            chains.append(c)
            lps.append(lp)
            infos.append(info)
            if verbose:
                log.info("walker %d: step size %.3g, mean tree depth %.2f, %d gradients", w + 1, info["step_size"],
                         info["mean_tree_depth"], info["n_grad"])
    var_names = []
    for sym, (lo, hi) in zip(bayes_model.var_symbols(), ranges):
        var_names += [f"{sym}[{x}]" for x in range(1, hi - lo + 1)]
    out = {"ids": np.asarray(arrays.bc_ids, dtype=object), "var_names": np.asarray(var_names, dtype=object),
           "chain": np.stack(chains), "logp": np.stack(lps), "step_size": np.asarray([i["step_size"] for i in infos])}
    if fname is None:
        return out
    if verbose:
        log.info("Saving %s chain...", fname)                                      # :155-157
    np.savez(fname, **out)
    return None
