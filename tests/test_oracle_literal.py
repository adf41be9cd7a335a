"""Pins the oracle: Philox known-answer vectors (Random123), distribution formulas against
scipy.stats, the independent-Poisson identity of SURVEY.md 7.2, committed golden vectors."""
import os

import numpy as np
import pytest
import scipy.stats as st
import torch

from oracle import advi, fixtures, literal, rng

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, want in kat:
        o = rng.philox4x32_10(np.array([c[0]]), c[1], c[2], c[3], k[0], k[1])
        assert tuple(int(x[0]) for x in o) == want


def test_normals_slicing_and_moments():
    e = rng.normals(5, 3, 1, 200001)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1) < 0.01
    assert st.kstest(e, "norm").pvalue > 1e-3
    np.testing.assert_array_equal(rng.normals(5, 3, 1, 200001, 7, 1002), e[7:1002])
    np.testing.assert_array_equal(rng.normals(5, 3, 1, 200001, 8, 11), e[8:11])
    assert not np.allclose(rng.normals(5, 4, 1, 100), e[:100])


def test_distribution_formulas_match_scipy():
    g = np.random.default_rng(0)
    x = g.integers(0, 5000, 12)
    lam = g.uniform(0.5, 4000, 12)
    ours = literal.poisson_logpdf(torch.tensor(x, dtype=torch.float64), torch.tensor(lam)).numpy()
    np.testing.assert_allclose(ours, st.poisson.logpmf(x, lam), rtol=1e-12)
    p = g.dirichlet(np.ones(9))
    n = 12345
    xs = g.multinomial(n, p)
    ours = float(literal.multinomial_logpdf(torch.tensor(xs, dtype=torch.float64), torch.tensor(float(n), dtype=torch.float64), torch.tensor(p)))
    assert abs(ours - st.multinomial.logpmf(xs, n, p)) < 1e-9 * abs(ours)
    assert float(literal.multinomial_logpdf(torch.tensor(xs, dtype=torch.float64), torch.tensor(float(n + 1), dtype=torch.float64), torch.tensor(p))) == -np.inf
    y, m, s = g.normal(size=7), g.normal(size=7), g.uniform(0.1, 2, 7)
    ours = float(literal.mvnormal_diag_logpdf(torch.tensor(y), torch.tensor(m), torch.tensor(s ** 2)))
    assert abs(ours - st.norm.logpdf(y, m, s).sum()) < 1e-12 * abs(ours)


def _scipy_logjoint_fitness(z, sp):
    """Independent evaluation of model_fitness_normal.jl:132-271 with scipy.stats only."""
    off, pr = sp.offsets(), sp.priors
    T, B, nn, nb = sp.n_time[0], sp.B, sp.n_neutral, sp.n_bc
    s_t, ls_t = z[slice(*off["s_pop"])], z[slice(*off["logsigma_pop"])]
    s_m, ls_m = z[slice(*off["s_bc"])], z[slice(*off["logsigma_bc"])]
    logL = z[slice(*off["loglambda"])]
    lp = st.norm.logpdf(s_t, *pr["s_pop_prior"]).sum() + st.norm.logpdf(ls_t, *pr["logsigma_pop_prior"]).sum()
    lp += st.norm.logpdf(s_m, *pr["s_bc_prior"]).sum() + st.norm.logpdf(ls_m, *pr["logsigma_bc_prior"]).sum()
    lp += st.norm.logpdf(logL, *pr["loglambda_prior"]).sum()
    Lam = np.exp(logL).reshape(B, T).T
    F = Lam / Lam.sum(axis=1, keepdims=True)
    lG = np.log(F[1:] / F[:-1])
    lp += st.poisson.logpmf(sp.totals[0], Lam.sum(axis=1)).sum()
    for t in range(T):
        lp += st.multinomial.logpmf(sp.counts[0][t], sp.totals[0][t], F[t])
    for b in range(nn):
        lp += st.norm.logpdf(lG[:, b], -s_t, np.exp(ls_t)).sum()
    for m in range(nb):
        lp += st.norm.logpdf(lG[:, nn + m], s_m[m] - s_t, np.exp(ls_m[m])).sum()
    return lp


def test_literal_fitness_matches_scipy_evaluation():
    sp = fixtures.load("data001_single")
    mu, _ = advi.meanfield_init(1, sp.D)
    z = mu * 0.3
    z[slice(*sp.offsets()["loglambda"])] += np.log(sp.counts[0].T.reshape(-1) + 1.0)
    lp, _ = literal.logjoint_and_grad(z, sp)
    assert abs(lp - _scipy_logjoint_fitness(z, sp)) < 1e-10 * abs(lp)


def test_independent_poisson_identity():
    """Poisson(n_t | sum lam) * Multinomial(R_t | n_t, F_t) == prod_b Poisson(R_tb | lam_tb)
    when n_t = sum_b R_tb (docs/src/math.md:405-407)."""
    sp = fixtures.load("data001_single")
    g = np.random.default_rng(3)
    Lam = torch.tensor(np.exp(g.normal(8, 1, sp.counts[0].shape)))
    R = torch.tensor(sp.counts[0], dtype=torch.float64)
    n = torch.tensor(sp.totals[0], dtype=torch.float64)
    lit = float(literal._obs_terms(Lam, R, n))
    fused = float(literal.poisson_logpdf(R, Lam).sum())
    assert abs(lit - fused) < 1e-12 * abs(lit)


@pytest.mark.parametrize("name", ["data001_single", "data002_hier-rep", "data003_multienv", "data004_multigen"])
def test_golden_vectors(name):
    gold = np.load(os.path.join(GOLD, f"golden_{name}.npz"))
    sp = fixtures.load(name)
    assert sp.D == int(gold["D"])
    lp, g = literal.logjoint_and_grad(gold["z"], sp)
    assert abs(lp - float(gold["logjoint"])) <= 1e-12 * abs(lp)
    np.testing.assert_allclose(g, gold["grad_z"], rtol=1e-10, atol=1e-10 * np.abs(g).max())
    el, gm, go = literal.elbo_and_grad(gold["mu"], gold["omega"], gold["eps"], sp)
    assert abs(el - float(gold["elbo"])) <= 1e-12 * abs(el)
    np.testing.assert_allclose(gm, gold["grad_mu"], rtol=1e-10, atol=1e-10 * np.abs(gm).max())
    np.testing.assert_allclose(go, gold["grad_omega"], rtol=1e-10, atol=1e-10 * np.abs(go).max())


def test_ragged_quirk_differs_only_in_neutral_term():
    """the ragged method's `repeat(.., inner=n_neutral)` ordering changes the density
    unless n_neutral == 1 or T_r == 2."""
    sp = fixtures.load("data002_hier-rep")
    mu, _ = advi.meanfield_init(2, sp.D)
    a = float(literal.logjoint(mu, sp))
    b = float(literal.logjoint(mu, sp, ragged_quirk=True))
    assert a != b
    sp1 = fixtures.synthetic("replicate", B=12, T=[3, 4], n_rep=2, n_neutral=1, seed=0)
    mu1, _ = advi.meanfield_init(2, sp1.D)
    assert abs(float(literal.logjoint(mu1, sp1)) - float(literal.logjoint(mu1, sp1, ragged_quirk=True))) < 1e-6


def test_optimisers_restated():
    o = advi.TruncatedADAGrad(eta=0.1, tau=40.0, n=3)
    o.init(2)
    d = np.array([3.0, -4.0])
    out = [o.apply(d * (k + 1)) for k in range(5)]
    # window of 3: step 4 sums squares of steps 2,3,4
    np.testing.assert_allclose(out[3], 4 * d * 0.1 / (40 + np.sqrt((4 + 9 + 16) * d ** 2)))
    o2 = advi.DecayedADAGrad()
    o2.init(2)
    r = o2.apply(d)
    np.testing.assert_allclose(r, d * 0.1 / (np.sqrt(0.9e-8 + d ** 2) + 1e-8))
