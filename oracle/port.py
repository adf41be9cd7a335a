"""ctypes wrapper of the oracle's C port (oracle/c/bb_port.c).  TEST INFRASTRUCTURE ONLY.

`time_workload` is the `cpu_baseline` leg of bench.py: the port timed on the GPU box's host cores
on a bounded number of steps of the same workload."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time
from typing import Optional

import numpy as np

from .spec import ModelSpec

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "c", "bb_port.c")
LIB = os.path.join(_HERE, "c", "libbb_port.so")
MAXR = 16
_KIND = {"fitness": 0, "multienv": 1, "genotype": 2, "replicate": 3, "multienv_replicate": 4}
_dp = C.POINTER(C.c_double)


class port_model(C.Structure):
    _fields_ = [("kind", C.c_int), ("R", C.c_int), ("E", C.c_int), ("G", C.c_int),
                ("nn", C.c_int64), ("nb", C.c_int64), ("D", C.c_int64), ("T", C.c_int * MAXR),
                ("counts", C.POINTER(C.c_int64)), ("env_idx", C.POINTER(C.c_int32)), ("geno_idx", C.POINTER(C.c_int32)),
                ("pmean", _dp), ("pstd", _dp),
                ("o_spop", C.c_int64), ("o_lspop", C.c_int64), ("o_s", C.c_int64), ("o_tt", C.c_int64),
                ("o_lt", C.c_int64), ("o_ls", C.c_int64), ("o_l", C.c_int64)]


_lib: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["gcc", "-std=gnu11", "-O2", "-fopenmp", "-fPIC", "-shared", SRC, "-o", LIB, "-lm"], check=True)
    return LIB


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.port_logjoint_grad.restype = C.c_double
        L.port_logjoint_grad.argtypes = [C.POINTER(port_model), _dp, _dp, C.c_int]
        L.port_elbo_grad.restype = C.c_double
        L.port_elbo_grad.argtypes = [C.POINTER(port_model), _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, C.c_int]
        L.port_normals.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, _dp]
        L.port_run.argtypes = [C.POINTER(port_model), _dp, _dp, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_double,
                               C.c_double, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64, _dp, _dp, C.c_int]
        _lib = L
    return _lib


def _p(a, ty=_dp):
    return a.ctypes.data_as(ty)


class Port:
    def __init__(self, sp: ModelSpec):
        self.sp = sp
        self.D = sp.D
        off = sp.offsets()
        m = port_model()
        m.kind, m.R, m.E, m.G = _KIND[sp.kind], sp.n_rep, sp.n_env, sp.n_geno
        m.nn, m.nb, m.D = sp.n_neutral, sp.n_bc, sp.D
        for r, t in enumerate(sp.n_time):
            m.T[r] = t
        self._counts = np.concatenate([np.ascontiguousarray(c.T).reshape(-1) for c in sp.counts]).astype(np.int64)
        m.counts = _p(self._counts, C.POINTER(C.c_int64))
        if sp.env_idx is not None:
            env = np.concatenate(sp.env_idx) if sp.kind == "multienv_replicate" else sp.env_idx    # replicate-major [sum T_r]
            self._env = np.ascontiguousarray(env, dtype=np.int32)
            m.env_idx = _p(self._env, C.POINTER(C.c_int32))
        if sp.geno_idx is not None:
            self._geno = np.ascontiguousarray(sp.geno_idx, dtype=np.int32)
            m.geno_idx = _p(self._geno, C.POINTER(C.c_int32))
        self._pm, self._ps = sp.prior_arrays()
        m.pmean, m.pstd = _p(self._pm), _p(self._ps)
        m.o_spop, m.o_lspop = off["s_pop"][0], off["logsigma_pop"][0]
        m.o_s = off["s_bc"][0] if "s_bc" in off else off["theta"][0]
        m.o_tt = off.get("theta_tilde", (0, 0))[0]
        m.o_lt = off.get("logtau", (0, 0))[0]
        m.o_ls, m.o_l = off["logsigma_bc"][0], off["loglambda"][0]
        self.m = m

    def logjoint_grad(self, z, nthreads: int = 1):
        z = np.ascontiguousarray(z, dtype=np.float64)
        g = np.empty(self.D)
        lp = lib().port_logjoint_grad(C.byref(self.m), _p(z), _p(g), nthreads)
        return lp, g

    def elbo_grad(self, mu, omega, eps, nthreads: int = 1):
        mu, omega = np.ascontiguousarray(mu, dtype=np.float64), np.ascontiguousarray(omega, dtype=np.float64)
        eps = np.ascontiguousarray(np.atleast_2d(eps), dtype=np.float64)
        gm, go, work = np.empty(self.D), np.empty(self.D), np.empty(2 * self.D)
        # gmu/gom must be contiguous for port_run; here separate arrays are fine
        el = lib().port_elbo_grad(C.byref(self.m), _p(mu), _p(omega), _p(eps), eps.shape[0], _p(gm), _p(go), _p(work), nthreads)
        return el, gm, go

    def run(self, mu, omega, n_steps, S=1, optimizer="TruncatedADAGrad", eta=0.1, tau=40.0, window=100,
            window_exact=False, pre=1.0, post=0.9, seed=0, first_step=0, state=None, nthreads=1):
        mu = np.array(mu, dtype=np.float64)
        omega = np.array(omega, dtype=np.float64)
        opt = 0 if optimizer == "TruncatedADAGrad" else 1
        if state is None:
            state = np.zeros((window + 2) * 2 * self.D) if opt == 0 else np.full(2 * self.D, 1e-8)   # window, running sums, their low-order parts
        trace = np.empty(n_steps)
        rc = lib().port_run(C.byref(self.m), _p(mu), _p(omega), first_step, n_steps, S, opt, eta, tau, window,
                            int(window_exact), pre, post, seed, _p(state), _p(trace), nthreads)
        assert rc == 0
        return mu, omega, trace, state


def spec_from_workload(wl) -> ModelSpec:
    return ModelSpec(kind=wl.kind, counts=wl.counts, totals=[c.sum(axis=1) for c in wl.counts],
                     n_neutral=wl.n_neutral, n_bc=wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx)


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, capped by a cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def time_workload(wl, ncores: int, seconds_budget: float = 20.0) -> dict:
    """steps/s of the port on `wl`: running-window TruncatedADAGrad(0.1, 40, 100), S = 1, same Philox
    stream as the engine; one thread and all `ncores` threads, each on a bounded number of steps."""
    from . import advi
    sp = spec_from_workload(wl)
    p = Port(sp)
    mu0, om0 = advi.meanfield_init(42, sp.D)
    out = {}
    ncores = max(1, min(ncores, usable_cores()))
    for label, nt in (("1", 1), ("all", ncores)):
        p.run(mu0, om0, 2, seed=42, nthreads=nt)          # touch pages, spin up the team
        n, t0 = 0, time.perf_counter()
        mu, om, state = mu0, om0, None
        chunk = 4
        while True:
            mu, om, _, state = p.run(mu, om, chunk, seed=42, first_step=n, state=state, nthreads=nt)
            n += chunk
            el = time.perf_counter() - t0
            if el > seconds_budget / 2 or n >= 400:
                break
        out[label] = (n / el, n)
    best = "all" if out["all"][0] >= out["1"][0] else "1"
    return {"value": round(out[best][0], 3), "unit": "steps/s", "cores": ncores if best == "all" else 1, "kind": "port",
            "sample": f"oracle/c/bb_port.c (fused analytic gradient, OpenMP), {out[best][1]} steps of the same workload; "
                      f"1 thread: {out['1'][0]:.3f} steps/s over {out['1'][1]} steps",
            "single_thread_value": round(out["1"][0], 3)}
