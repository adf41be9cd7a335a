#!/usr/bin/env python3
"""Diagnostic: where the wall time of `bench.py --steps 20 --warmup 5`'s timed region goes (median of REP repeats, us)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import barbay_jl_amd as bb
from barbay_jl_amd import synth, _capi

REP = int(os.environ.get("REP", 15))
N = int(os.environ.get("N", 20))
wl = synth.fitness_normal(50000, 8, 42)
_lib = _capi.load_library(os.environ["LIB"]) if os.environ.get("LIB") else None
e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, _lib=_lib)
e.run(5)
pc = time.perf_counter
rows = []
for rep in range(REP):
    torch.cuda.synchronize()
    t0 = pc(); torch.cuda.synchronize(); t_sync_idle = pc() - t0
    t0 = pc(); e.run(N); t_run = pc() - t0
    t1 = pc(); torch.cuda.synchronize(); t_sync_after = pc() - t1
    k = e.stats()["last_run_ms"] * 1e3
    t0 = pc(); e._lib.bb_run(e._h, N); t_raw = pc() - t0
    k2 = e.stats()["last_run_ms"] * 1e3
    t0 = pc(); e._lib.bb_run(e._h, 0); t_zero = pc() - t0
    rows.append((t_sync_idle * 1e6, t_run * 1e6, t_sync_after * 1e6, k, t_raw * 1e6, k2, t_zero * 1e6))
r = np.median(np.array(rows), axis=0)
print(f"N = {N}: torch.cuda.synchronize() idle {r[0]:.1f} | Engine.run wall {r[1]:.1f} (HIP events {r[3]:.1f}) | synchronize after it {r[2]:.1f} | "
      f"bench region = {r[1] + r[2]:.1f} -> {N / (r[1] + r[2]) * 1e6:.0f} steps/s | raw ctypes bb_run wall {r[4]:.1f} (events {r[5]:.1f}) | bb_run(0) wall {r[6]:.1f}")
print("first repeats (cold):", " ".join(f"{x[1]:.0f}" for x in rows[:5]))
