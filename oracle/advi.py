"""Oracle ADVI loop: AdvancedVI 0.2 `optimize!` + optimisers, restated.  TEST INFRASTRUCTURE ONLY.

Third-party algorithm (AdvancedVI 0.2.x, Turing 0.36; not under /root/reference, no
Manifest.toml pins the patch version).  Anchored on the reference's call site
`q = Turing.vi(bayes_model, advi; optimizer=opt)` (src/vi.jl:201) and option types
(src/vi.jl:98-99).  Restated from the published sources:

  optimize!:  for i in 1:max_iters:  D = grad(-ELBO)(theta);  D = apply!(opt, theta, D);  theta -= D
  theta = [mu; omega],  sigma = softplus(omega)            (Turing.meanfield / update)
  TruncatedADAGrad(eta=0.1, tau=40, n=100):
      g2[mod(i-1, n)+1] = D^2;  s = sum(g2) (slot order);  D *= eta / (tau + sqrt(s))
  DecayedADAGrad(eta=0.1, pre=1.0, post=0.9):
      acc (init 1e-8) = post*acc + pre*D^2;  D *= eta / (sqrt(acc) + 1e-8)

PARITY UNPINNED (see oracle/__init__.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional, Tuple

import numpy as np

from . import rng
from .spec import ModelSpec


@dataclass
class TruncatedADAGrad:
    eta: float = 0.1
    tau: float = 40.0
    n: int = 100

    def init(self, nparams: int):
        self.g2 = np.zeros((self.n, nparams))
        self.i = 1

    def apply(self, delta: np.ndarray) -> np.ndarray:
        idx = (self.i - 1) % self.n
        self.g2[idx] = delta ** 2
        s = np.zeros_like(delta)
        for j in range(self.n):          # sum(g2): slot order
            s = s + self.g2[j]
        self.i += 1
        return delta * (self.eta / (self.tau + np.sqrt(s)))


@dataclass
class DecayedADAGrad:
    eta: float = 0.1
    pre: float = 1.0
    post: float = 0.9

    def init(self, nparams: int):
        self.acc = np.full(nparams, 1e-8)

    def apply(self, delta: np.ndarray) -> np.ndarray:
        self.acc = self.post * self.acc + self.pre * delta ** 2
        return delta * (self.eta / (np.sqrt(self.acc) + 1e-8))


def meanfield_init(seed: int, D: int) -> Tuple[np.ndarray, np.ndarray]:
    """Turing.meanfield: mu0 = randn(D), sigma0 = softplus.(randn(D)) => omega0 = randn(D);
    drawn here from the engine's init streams (oracle/rng.py)."""
    mu0 = rng.normals(seed, 0, rng.STREAM_INIT_MU, D)
    om0 = rng.normals(seed, 0, rng.STREAM_INIT_OMEGA, D)
    return mu0, om0


def softplus(x: np.ndarray) -> np.ndarray:
    return np.maximum(x, 0.0) + np.log1p(np.exp(-np.abs(x)))


def run_advi(sp: ModelSpec, elbo_grad: Callable, mu0: np.ndarray, om0: np.ndarray, n_steps: int,
             samples_per_step: int = 1, opt=None, seed: int = 0,
             eps_fn: Optional[Callable[[int], np.ndarray]] = None, first_step: int = 0):
    """elbo_grad(mu, omega, eps[S,D]) -> (elbo, dELBO/dmu, dELBO/domega).
    eps for step i comes from eps_fn(i) if given, else from the Philox stream
    (seed, step=i, stream=s).  Returns (mu, omega, elbo_trace)."""
    D = mu0.shape[0]
    opt = opt or TruncatedADAGrad()
    opt.init(2 * D)
    theta = np.concatenate([mu0, om0]).astype(np.float64)
    trace = []
    for i in range(first_step, first_step + n_steps):
        if eps_fn is not None:
            eps = eps_fn(i)
        else:
            eps = np.stack([rng.normals(seed, i, s, D) for s in range(samples_per_step)])
        elbo, gmu, gom = elbo_grad(theta[:D], theta[D:], eps)
        delta = -np.concatenate([gmu, gom])          # gradient of -ELBO
        theta = theta - opt.apply(delta)
        trace.append(elbo)
    return theta[:D].copy(), theta[D:].copy(), np.array(trace)
