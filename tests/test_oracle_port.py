"""The oracle's C port (oracle/c/bb_port.c, the timed CPU baseline) against the literal oracle."""
import numpy as np
import pytest

import _cases as c
from oracle import advi, fixtures, literal, port, rng


@pytest.mark.parametrize("name", ["data001_single", "data002_hier-rep", "data003_multienv", "data004_multigen"])
@pytest.mark.parametrize("nthreads", [1, 3])
def test_port_elbo_grad_fixtures(name, nthreads):
    sp = fixtures.load(name)
    p = port.Port(sp)
    mu, om = advi.meanfield_init(3, sp.D)
    mu, om = mu * 0.2 + 3, om * 0.5 - 2
    eps = np.stack([rng.normals(3, 0, s, sp.D) for s in range(2)])
    el, gm, go = literal.elbo_and_grad(mu, om, eps, sp)
    el2, gm2, go2 = p.elbo_grad(mu, om, eps, nthreads=nthreads)
    assert abs(el - el2) <= 1e-12 * abs(el)
    assert np.abs(gm - gm2).max() <= 1e-12 * np.abs(gm).max()
    assert np.abs(go - go2).max() <= 1e-12 * np.abs(go).max()


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "genotype", "replicate_ragged", "multienv_replicate"])
@pytest.mark.parametrize("opt", ["TruncatedADAGrad", "DecayedADAGrad"])
def test_port_trajectory(name, opt):
    sp = c.synth(name, seed=6)
    p = port.Port(sp)
    mu0, om0 = advi.meanfield_init(9, sp.D)
    f = lambda m, o, e: literal.elbo_and_grad(m, o, e, sp)
    o_ref = advi.TruncatedADAGrad(n=4) if opt == "TruncatedADAGrad" else advi.DecayedADAGrad()
    m2, o2, tr = advi.run_advi(sp, f, mu0, om0, 9, 2, o_ref, 9)
    m3, o3, tr3, _ = p.run(mu0, om0, 9, S=2, optimizer=opt, window=4, window_exact=True, seed=9, nthreads=2)
    assert np.abs(m2 - m3).max() < 1e-10 and np.abs(o2 - o3).max() < 1e-10
    assert np.abs(tr - tr3).max() <= 1e-12 * np.abs(tr).max()


def test_port_normals_match_numpy_stream():
    out = np.empty(1001)
    port.lib().port_normals(77, 5, 2, 1001, out.ctypes.data_as(port._dp))
    assert np.abs(out - rng.normals(77, 5, 2, 1001)).max() < 1e-13


@pytest.mark.parametrize("name", ["multienv_replicate", "multienv_replicate_3d"])
def test_port_elbo_grad_multienv_replicate(name):
    """The fifth model (no reference fixture exercises it): the port against the literal transcription on synthetic data."""
    sp = c.synth(name, seed=3)
    p = port.Port(sp)
    mu, om = advi.meanfield_init(3, sp.D)
    mu, om = mu * 0.2 + 3, om * 0.5 - 2
    eps = np.stack([rng.normals(3, 0, s, sp.D) for s in range(2)])
    el, gm, go = literal.elbo_and_grad(mu, om, eps, sp)
    el2, gm2, go2 = p.elbo_grad(mu, om, eps, nthreads=2)
    assert abs(el - el2) <= 1e-12 * abs(el)
    assert np.abs(gm - gm2).max() <= 1e-12 * np.abs(gm).max() and np.abs(go - go2).max() <= 1e-12 * np.abs(go).max()
