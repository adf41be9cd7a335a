#!/usr/bin/env python3
"""Diagnostic: steps/s of C2 with several MC samples per step and with ELBO recording -- k_res's MS instances (launch_mode 2)
against the two-kernel step (launch_mode 1).  WL=C5: the same on BASELINE config 5 on one GPU (k_stream's MS instances, round 4)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb
from barbay_jl_amd import synth
c5 = os.environ.get("WL") == "C5"
wl = synth.genotype_fitness_normal(200_000, 8, 5_000, 45) if c5 else synth.fitness_normal(50000, 8, 42)
for S, ev in (((1, 0), (2, 0), (1, 1), (1, 10)) if c5 else ((1, 0), (2, 0), (4, 0), (1, 1), (1, 10), (2, 10))):
    row = []
    for mode in (2, 1):
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, geno_idx=wl.geno_idx, seed=42, samples_per_step=S, elbo_every=ev, launch_mode=mode)
        e.run(60 if c5 else 200)
        n = (300 if c5 else 2000) if mode == 2 else (100 if c5 else 400)
        e.run(n)
        st = e.stats()
        row.append(f"mode {mode}: {e.kernel_name() if mode == 2 else 'two kernels'} {n / st['last_run_ms'] * 1e3:9.1f} steps/s")
        e.close()
    print(f"S = {S}, elbo_every = {ev:2d}: " + " | ".join(row), flush=True)
