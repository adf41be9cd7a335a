"""Regenerates tests/golden/golden_*.npz from the literal oracle (oracle/literal.py) on the
reference's own test CSVs (copied next to this file as data).  The reference itself cannot run in
this image (no julia), so these vectors pin the ORACLE against regressions; they are not outputs of
the reference.  Usage: python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import advi, fixtures, literal, rng  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
for i, name in enumerate(["data001_single", "data002_hier-rep", "data003_multienv", "data004_multigen"]):
    sp = fixtures.load(name)
    seed = 100 + i
    mu, om = advi.meanfield_init(seed, sp.D)
    lo, hi = sp.offsets()["loglambda"]
    mu[lo:hi] = np.log(np.concatenate([c.T.reshape(-1) for c in sp.counts]) + 1.0) + 0.1 * mu[lo:hi]
    om = 0.5 * om - 2.0
    eps = np.stack([rng.normals(seed, 0, s, sp.D) for s in range(2)])
    z = mu + advi.softplus(om) * eps[0]
    lp, gz = literal.logjoint_and_grad(z, sp)
    el, gm, go = literal.elbo_and_grad(mu, om, eps, sp)
    np.savez_compressed(os.path.join(HERE, f"golden_{name}.npz"), D=sp.D, z=z, logjoint=lp, grad_z=gz, mu=mu,
                        omega=om, eps=eps, elbo=el, grad_mu=gm, grad_omega=go)
    print(name, sp.D, lp, el)
