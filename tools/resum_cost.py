#!/usr/bin/env python3
"""Diagnostic: what the periodic exact re-add of the TruncatedADAGrad window costs on C2 (steps/s for several resum_every)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import synth  # noqa: E402

wl = synth.fitness_normal(50_000, 8, 42)
for re_ in (100, 200, 400, 1000, 100000):
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, resum_every=re_)
    e.run(2000)
    e.run(8000)
    ms = e.stats()["last_run_ms"]
    print(f"resum_every {re_:6d}: {8000 / ms * 1e3:9.1f} steps/s ({ms / 8:.3f} us/step)", flush=True)
    e.close()
