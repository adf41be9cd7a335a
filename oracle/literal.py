"""Literal fp64 transcription of the reference log-joints.  TEST INFRASTRUCTURE ONLY.

Each ``logjoint_*`` follows its reference model body statement by statement
(Poisson + Multinomial + diagonal MvNormal exactly as written; no
independent-Poisson identity, no moment decomposition), on torch fp64 tensors so
that gradients come from autograd and share no algebra with the HIP kernels.

Distribution log-densities are the Distributions.jl 0.25 formulas (third-party,
not under /root/reference; restated from the published definitions and
cross-checked against scipy.stats in tests/test_oracle_literal.py):

* Poisson      logpdf(x; lam)     = xlogy(x, lam) - lam - lgamma(x + 1)
* Multinomial  logpdf(x; n, p)    = lgamma(n+1) - sum lgamma(x_i+1) + sum xlogy(x_i, p_i);
                                    -Inf when sum(x) != n
* MvNormal(mu, Diagonal(v))       = sum -0.5*log(2 pi v_i) - 0.5 (x_i-mu_i)^2 / v_i

PARITY UNPINNED (see oracle/__init__.py).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np
import torch

from .spec import ModelSpec

LOG2PI = math.log(2.0 * math.pi)
F64 = torch.float64


# ---- distribution pieces ---------------------------------------------------
def mvnormal_diag_logpdf(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor) -> torch.Tensor:
    return torch.sum(-0.5 * (LOG2PI + torch.log(var)) - 0.5 * (x - mean) ** 2 / var)


def poisson_logpdf(x: torch.Tensor, lam: torch.Tensor) -> torch.Tensor:
    return torch.xlogy(x, lam) - lam - torch.lgamma(x + 1.0)


def multinomial_logpdf(x: torch.Tensor, n: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    if float(x.sum()) != float(n):
        return torch.tensor(-math.inf, dtype=F64)
    return torch.lgamma(n + 1.0) - torch.sum(torch.lgamma(x + 1.0)) + torch.sum(torch.xlogy(x, p))


def _prior(z: torch.Tensor, mean, std) -> torch.Tensor:
    """`x ~ MvNormal(mean, Diagonal(std.^2))` for Vector- or Matrix-form priors
    (model_fitness_normal.jl:137-146 and siblings)."""
    n = z.shape[0]
    m = torch.as_tensor(np.broadcast_to(np.asarray(mean, dtype=np.float64), (n,)).copy())
    s = torch.as_tensor(np.broadcast_to(np.asarray(std, dtype=np.float64), (n,)).copy())
    return mvnormal_diag_logpdf(z, m, s ** 2)


def _jl_reshape(v: torch.Tensor, *shape: int) -> torch.Tensor:
    """Julia column-major reshape of a vector."""
    return v.reshape(*reversed(shape)).permute(*reversed(range(len(shape))))


def _jl_vec(a: torch.Tensor) -> torch.Tensor:
    """Julia vec() of a column-major array."""
    return a.permute(*reversed(range(a.dim()))).reshape(-1)


def _obs_terms(Lam: torch.Tensor, R: torch.Tensor, n_t: torch.Tensor) -> torch.Tensor:
    """n_t ~ arraydist(Poisson.(row sums)) and sum_t logpdf(Multinomial(n_t, F_t), R_t)
    (model_fitness_normal.jl:224-226, 239-244).  Lam, R are T x B."""
    F = Lam / Lam.sum(dim=1, keepdim=True)
    lp = poisson_logpdf(n_t, Lam.sum(dim=1)).sum()
    for t in range(Lam.shape[0]):
        lp = lp + multinomial_logpdf(R[t], n_t[t], F[t])
    return lp


# ---- models ------------------------------------------------------------------
def logjoint_fitness(z: torch.Tensor, sp: ModelSpec) -> torch.Tensor:
    """model_fitness_normal.jl:132-271."""
    off, pr = sp.offsets(), sp.priors
    T, B, nn, nb = sp.n_time[0], sp.B, sp.n_neutral, sp.n_bc
    s_t = z[slice(*off["s_pop"])]
    lsig_t = z[slice(*off["logsigma_pop"])]
    s_m = z[slice(*off["s_bc"])]
    lsig_m = z[slice(*off["logsigma_bc"])]
    logL = z[slice(*off["loglambda"])]
    lp = _prior(s_t, *pr["s_pop_prior"])                       # :137-146
    lp = lp + _prior(lsig_t, *pr["logsigma_pop_prior"])        # :149-159
    lp = lp + _prior(s_m, *pr["s_bc_prior"])                   # :164-173
    lp = lp + _prior(lsig_m, *pr["logsigma_bc_prior"])         # :177-187
    lp = lp + _prior(logL, *pr["loglambda_prior"])             # :192-203
    Lam = _jl_reshape(torch.exp(logL), T, B)                   # :209
    F = Lam / Lam.sum(dim=1, keepdim=True)                     # :212
    logG = torch.log(F[1:, :] / F[:-1, :])                     # :215
    logG_n = _jl_vec(logG[:, :nn])                             # :218
    logG_m = _jl_vec(logG[:, nn:nn + nb])                      # :219
    R = torch.as_tensor(sp.counts[0], dtype=F64)
    n_t = torch.as_tensor(sp.totals[0], dtype=F64)
    lp = lp + _obs_terms(Lam, R, n_t)                          # :224-244
    lp = lp + mvnormal_diag_logpdf(                            # :251-257
        logG_n, (-s_t).repeat(nn), (torch.exp(lsig_t) ** 2).repeat(nn))
    lp = lp + mvnormal_diag_logpdf(                            # :262-270
        logG_m,
        s_m.repeat_interleave(T - 1) - s_t.repeat(nb),
        (torch.exp(lsig_m) ** 2).repeat_interleave(T - 1))
    return lp


def logjoint_multienv(z: torch.Tensor, sp: ModelSpec) -> torch.Tensor:
    """model_multienv_fitness_normal.jl:146-302."""
    off, pr = sp.offsets(), sp.priors
    T, B, nn, nb, E = sp.n_time[0], sp.B, sp.n_neutral, sp.n_bc, sp.n_env
    env_idx = torch.as_tensor(sp.env_idx)
    s_t = z[slice(*off["s_pop"])]
    lsig_t = z[slice(*off["logsigma_pop"])]
    s_m = z[slice(*off["s_bc"])]
    lsig_m = z[slice(*off["logsigma_bc"])]
    logL = z[slice(*off["loglambda"])]
    lp = _prior(s_t, *pr["s_pop_prior"])
    lp = lp + _prior(lsig_t, *pr["logsigma_pop_prior"])
    lp = lp + _prior(s_m, *pr["s_bc_prior"])
    lp = lp + _prior(lsig_m, *pr["logsigma_bc_prior"])
    lp = lp + _prior(logL, *pr["loglambda_prior"])
    Lam = _jl_reshape(torch.exp(logL), T, B)                   # :234
    F = Lam / Lam.sum(dim=1, keepdim=True)                     # :237
    logG = torch.log(F[1:, :] / F[:-1, :])                     # :240
    logG_n = _jl_vec(logG[:, :nn])
    logG_m = _jl_vec(logG[:, nn:nn + nb])
    R = torch.as_tensor(sp.counts[0], dtype=F64)
    n_t = torch.as_tensor(sp.totals[0], dtype=F64)
    lp = lp + _obs_terms(Lam, R, n_t)                          # :248-268
    s_m2 = _jl_reshape(s_m, E, nb)                             # :271  n_env x n_bc
    lsig_m2 = _jl_reshape(lsig_m, E, nb)                       # :272
    lp = lp + mvnormal_diag_logpdf(                            # :279-285
        logG_n, (-s_t).repeat(nn), (torch.exp(lsig_t) ** 2).repeat(nn))
    lp = lp + mvnormal_diag_logpdf(                            # :293-301
        logG_m,
        _jl_vec(s_m2[env_idx[1:], :]) - s_t.repeat(nb),
        _jl_vec(torch.exp(lsig_m2[env_idx[1:], :]) ** 2))
    return lp


def logjoint_genotype(z: torch.Tensor, sp: ModelSpec) -> torch.Tensor:
    """model_fitness_normal_hierarchical_genotypes.jl:165-329."""
    off, pr = sp.offsets(), sp.priors
    T, B, nn, nb = sp.n_time[0], sp.B, sp.n_neutral, sp.n_bc
    geno_idx = torch.as_tensor(sp.geno_idx)
    s_t = z[slice(*off["s_pop"])]
    lsig_t = z[slice(*off["logsigma_pop"])]
    theta = z[slice(*off["theta"])]
    theta_t = z[slice(*off["theta_tilde"])]
    ltau = z[slice(*off["logtau"])]
    lsig_m = z[slice(*off["logsigma_bc"])]
    logL = z[slice(*off["loglambda"])]
    lp = _prior(s_t, *pr["s_pop_prior"])                       # :182-191
    lp = lp + _prior(lsig_t, *pr["logsigma_pop_prior"])        # :194-204
    lp = lp + _prior(theta, *pr["s_bc_prior"])                 # :209-218
    lp = lp + _prior(theta_t, 0.0, 1.0)                        # :221
    lp = lp + _prior(ltau, *pr["logtau_prior"])                # :224-227
    s_m = theta[geno_idx] + torch.exp(ltau) * theta_t          # :230
    lp = lp + _prior(lsig_m, *pr["logsigma_bc_prior"])         # :233-243
    lp = lp + _prior(logL, *pr["loglambda_prior"])             # :247-258
    Lam = _jl_reshape(torch.exp(logL), T, B)                   # :264
    F = Lam / Lam.sum(dim=1, keepdim=True)                     # :267
    logG = torch.log(F[1:, :] / F[:-1, :])                     # :270
    logG_n = _jl_vec(logG[:, :nn])
    logG_m = _jl_vec(logG[:, nn:nn + nb])
    R = torch.as_tensor(sp.counts[0], dtype=F64)
    n_t = torch.as_tensor(sp.totals[0], dtype=F64)
    lp = lp + _obs_terms(Lam, R, n_t)                          # :278-298
    lp = lp + mvnormal_diag_logpdf(                            # :305-311
        logG_n, (-s_t).repeat(nn), (torch.exp(lsig_t) ** 2).repeat(nn))
    lp = lp + mvnormal_diag_logpdf(                            # :316-328
        logG_m,
        s_m.repeat_interleave(T - 1) - s_t.repeat(nb),
        (torch.exp(lsig_m) ** 2).repeat_interleave(T - 1))
    return lp


def logjoint_replicate(z: torch.Tensor, sp: ModelSpec, ragged_quirk: bool = False) -> torch.Tensor:
    """model_fitness_normal_hierarchical_replicates.jl:158-331 (3-D method) and
    :420-637 (ragged method).  With equal T_r the two methods define the same
    density; they are transcribed as one per-replicate loop in the ragged
    method's shape (rep_ranges / time_ranges, :426-447).

    ragged_quirk=True reproduces the ragged method's neutral-term ordering
    exactly as written (`repeat(s_t[range], inner=n_neutral)`, :599-605, against
    a time-fastest data vector :549) -- SURVEY.md quirk Q1.  False uses the
    ordering of the 3-D method (:307-311), which is self-consistent.
    """
    off, pr = sp.offsets(), sp.priors
    B, nn, nb, Rn = sp.B, sp.n_neutral, sp.n_bc, sp.n_rep
    Ts = sp.n_time
    s_t = z[slice(*off["s_pop"])]
    lsig_t = z[slice(*off["logsigma_pop"])]
    theta = z[slice(*off["theta"])]
    theta_t = z[slice(*off["theta_tilde"])]
    ltau = z[slice(*off["logtau"])]
    lsig_m = z[slice(*off["logsigma_bc"])]
    logL = z[slice(*off["loglambda"])]
    lp = _prior(s_t, *pr["s_pop_prior"])                       # :165-174 / :452-461
    lp = lp + _prior(lsig_t, *pr["logsigma_pop_prior"])        # :177-187 / :464-473
    lp = lp + _prior(theta, *pr["s_bc_prior"])                 # :192-201 / :478-487
    lp = lp + _prior(theta_t, 0.0, 1.0)                        # :205-207 / :491-493
    lp = lp + _prior(ltau, *pr["logtau_prior"])                # :210-213 / :496-499
    s_m = theta.repeat(Rn) + torch.exp(ltau) * theta_t         # :216 / :502
    lp = lp + _prior(lsig_m, *pr["logsigma_bc_prior"])         # :219-228 / :505-515
    lp = lp + _prior(logL, *pr["loglambda_prior"])             # :231-243 / :518-530
    s_m2 = _jl_reshape(s_m, nb, Rn)                            # :553
    lsig_m2 = _jl_reshape(lsig_m, nb, Rn)                      # :554
    expL = torch.exp(logL)
    ro, to = 0, 0
    for r in range(Rn):
        T = Ts[r]
        Lam = _jl_reshape(expL[ro:ro + T * B], T, B)           # :249 / :536-539
        F = Lam / Lam.sum(dim=1, keepdim=True)                 # :252 / :542
        logG = torch.log(F[1:, :] / F[:-1, :])                 # :255 / :545
        logG_n = _jl_vec(logG[:, :nn])                         # :259 / :549
        logG_m = _jl_vec(logG[:, nn:nn + nb])                  # :260 / :550
        Rr = torch.as_tensor(sp.counts[r], dtype=F64)
        n_t = torch.as_tensor(sp.totals[r], dtype=F64)
        lp = lp + _obs_terms(Lam, Rr, n_t)                     # :265-291 / :559-586
        st_r = s_t[to:to + T - 1]
        sg_r = lsig_t[to:to + T - 1]
        if ragged_quirk:
            mean_n = -(st_r.repeat_interleave(nn))             # :599
            var_n = (torch.exp(sg_r) ** 2).repeat_interleave(nn)   # :601-607
        else:
            mean_n = (-st_r).repeat(nn)                        # :307
            var_n = (torch.exp(sg_r) ** 2).repeat(nn)          # :310-312
        lp = lp + mvnormal_diag_logpdf(logG_n, mean_n, var_n)  # :304-315 / :596-610
        lp = lp + mvnormal_diag_logpdf(                        # :320-330 / :615-634
            logG_m,
            s_m2[:, r].repeat_interleave(T - 1) - st_r.repeat(nb),
            (torch.exp(lsig_m2[:, r]) ** 2).repeat_interleave(T - 1))
        ro += T * B
        to += T - 1
    return lp


def logjoint_multienv_replicate(z: torch.Tensor, sp: ModelSpec) -> torch.Tensor:
    """model_multienv_fitness_normal_hierarchical_replicates.jl:172-362 (3-D method) and :462-686 (ragged method),
    transcribed as one per-replicate loop in the ragged method's shape (rep_ranges / time_ranges :474-491); with
    equal T_r and a common env list the two methods define the same density (the ragged method's neutral term uses
    the outer `repeat(s_t[range], n_neutral)`, :653-667, i.e. no ordering quirk here)."""
    off, pr = sp.offsets(), sp.priors
    B, nn, nb, Rn, E = sp.B, sp.n_neutral, sp.n_bc, sp.n_rep, sp.n_env
    Ts = sp.n_time
    s_t = z[slice(*off["s_pop"])]
    lsig_t = z[slice(*off["logsigma_pop"])]
    theta = z[slice(*off["theta"])]
    theta_t = z[slice(*off["theta_tilde"])]
    ltau = z[slice(*off["logtau"])]
    lsig_m = z[slice(*off["logsigma_bc"])]
    logL = z[slice(*off["loglambda"])]
    lp = _prior(s_t, *pr["s_pop_prior"])                       # :191-200 / :498-507
    lp = lp + _prior(lsig_t, *pr["logsigma_pop_prior"])        # :203-213 / :510-520
    lp = lp + _prior(theta, *pr["s_bc_prior"])                 # :218-227 / :525-534
    lp = lp + _prior(theta_t, 0.0, 1.0)                        # :230-232 / :537-539
    lp = lp + _prior(ltau, *pr["logtau_prior"])                # :235-238 / :542-545
    s_m = theta.repeat(Rn) + torch.exp(ltau) * theta_t         # :241 / :548
    lp = lp + _prior(lsig_m, *pr["logsigma_bc_prior"])         # :244-254 / :551-561
    lp = lp + _prior(logL, *pr["loglambda_prior"])             # :258-270 / :565-577
    s_m3 = _jl_reshape(s_m, E, nb, Rn)                         # :311 / :606   n_env x n_bc x n_rep
    lsig_m3 = _jl_reshape(lsig_m, E, nb, Rn)                   # :312 / :607
    expL = torch.exp(logL)
    ro, to = 0, 0
    for r in range(Rn):
        T = Ts[r]
        env_idx = torch.as_tensor(sp.env_idx[r])
        Lam = _jl_reshape(expL[ro:ro + T * B], T, B)           # :276 / :584-587
        F = Lam / Lam.sum(dim=1, keepdim=True)                 # :279 / :590
        logG = torch.log(F[1:, :] / F[:-1, :])                 # :282 / :593
        logG_n = _jl_vec(logG[:, :nn])
        logG_m = _jl_vec(logG[:, nn:nn + nb])
        Rr = torch.as_tensor(sp.counts[r], dtype=F64)
        n_t = torch.as_tensor(sp.totals[r], dtype=F64)
        lp = lp + _obs_terms(Lam, Rr, n_t)                     # :291-306 / :612-637
        st_r = s_t[to:to + T - 1]
        sg_r = lsig_t[to:to + T - 1]
        lp = lp + mvnormal_diag_logpdf(logG_n, (-st_r).repeat(nn), (torch.exp(sg_r) ** 2).repeat(nn))   # :322-334 / :647-668
        lp = lp + mvnormal_diag_logpdf(                        # :339-352 / :670-682
            logG_m,
            _jl_vec(s_m3[env_idx[1:], :, r]) - st_r.repeat(nb),
            _jl_vec(torch.exp(lsig_m3[env_idx[1:], :, r])) ** 2)
        ro += T * B
        to += T - 1
    return lp


_LOGJOINT = {
    "fitness": logjoint_fitness,
    "multienv": logjoint_multienv,
    "genotype": logjoint_genotype,
    "replicate": logjoint_replicate,
    "multienv_replicate": logjoint_multienv_replicate,
}


def logjoint(z, sp: ModelSpec, **kw) -> torch.Tensor:
    z = torch.as_tensor(z, dtype=F64)
    return _LOGJOINT[sp.kind](z, sp, **kw)


def logjoint_and_grad(z: np.ndarray, sp: ModelSpec, **kw) -> Tuple[float, np.ndarray]:
    zt = torch.tensor(np.asarray(z, dtype=np.float64), requires_grad=True)
    lp = _LOGJOINT[sp.kind](zt, sp, **kw)
    (g,) = torch.autograd.grad(lp, zt)
    return float(lp.detach()), g.numpy()


# ---- ELBO (AdvancedVI 0.2 `ELBO` functor + Turing.meanfield family) [third-party, restated]
def elbo_and_grad(mu: np.ndarray, omega: np.ndarray, eps: np.ndarray, sp: ModelSpec,
                  **kw) -> Tuple[float, np.ndarray, np.ndarray]:
    """ELBO = (1/S) sum_s logjoint(mu + softplus(omega)*eps_s) + H(q),
    H = D (1 + log 2pi)/2 + sum log softplus(omega); all bijectors are identity
    (every latent is an unconstrained MvNormal variable), so logabsdetjac = 0.
    eps is S x D.  Returns (ELBO, dELBO/dmu, dELBO/domega)."""
    eps = np.atleast_2d(np.asarray(eps, dtype=np.float64))
    S, D = eps.shape
    mu_t = torch.tensor(np.asarray(mu, dtype=np.float64), requires_grad=True)
    om_t = torch.tensor(np.asarray(omega, dtype=np.float64), requires_grad=True)
    sigma = torch.logaddexp(om_t, torch.zeros_like(om_t))   # exact softplus (StatsFuns.log1pexp)
    acc = torch.zeros((), dtype=F64)
    for s in range(S):
        zz = mu_t + sigma * torch.as_tensor(eps[s])
        acc = acc + _LOGJOINT[sp.kind](zz, sp, **kw) / S
    elbo = acc + 0.5 * D * (1.0 + LOG2PI) + torch.sum(torch.log(sigma))
    gmu, gom = torch.autograd.grad(elbo, (mu_t, om_t))
    return float(elbo.detach()), gmu.numpy(), gom.numpy()
