#!/usr/bin/env python3
"""Diagnostic: steps/s of C2 with several MC samples per step and with ELBO recording -- k_res's MS instances (launch_mode 2)
against the two-kernel step (launch_mode 1)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb
from barbay_jl_amd import synth
wl = synth.fitness_normal(50000, 8, 42)
for S, ev in ((1, 0), (2, 0), (4, 0), (1, 1), (1, 10), (2, 10)):
    row = []
    for mode in (2, 1):
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, samples_per_step=S, elbo_every=ev, launch_mode=mode)
        e.run(200)
        n = 2000 if mode == 2 else 400
        e.run(n)
        st = e.stats()
        row.append(f"mode {mode}: k{st['resident_kernel']} {n / st['last_run_ms'] * 1e3:9.1f} steps/s")
        e.close()
    print(f"S = {S}, elbo_every = {ev:2d}: " + " | ".join(row), flush=True)
