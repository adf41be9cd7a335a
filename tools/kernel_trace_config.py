#!/usr/bin/env python3
"""One launch of STEPS steps of a BASELINE config for `rocprofv3 --kernel-trace --stats`:  CFG=C3 python tools/kernel_trace_config.py
(prints the HIP-event time of the same launch; profiles/r02*/kernel_stats_C*.csv)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import synth  # noqa: E402

cfg = os.environ.get("CFG", "C3")
wl = {"C2": lambda: synth.fitness_normal(50_000, 8, 42), "C3": lambda: synth.replicate_fitness_normal(20_000, 6, 3, 43),
      "C4": lambda: synth.multienv_fitness_normal(20_000, 6, (1, 1, 2, 3, 4, 1), 44),
      "C5rank": lambda: synth.genotype_fitness_normal(25_000, 8, 625, 45),
      # the same shard shape with loglambda at an EVEN flat index, as in the real 8-rank run (global offset 593 014): the plain instance
      "C5rank_plain": lambda: synth.genotype_fitness_normal(25_000, 8, 626, 45), "C5": lambda: synth.genotype_fitness_normal(200_000, 8, 5_000, 45)}[cfg]()
steps = int(os.environ.get("STEPS", 4000))
e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=1)
e.run(200)
e.run(steps)
st = e.stats()
print(f"{cfg}: {e.kernel_name()} {st['n_blocks']} x {st['block_threads']}: {steps} steps, HIP events {st['last_run_ms']:.3f} ms = {st['last_run_ms'] / steps * 1e3:.3f} us per step, algorithmic {st['bytes_per_step']} B per step -> "
      f"{st['bytes_per_step'] * steps / st['last_run_ms'] / 1e6:.0f} GB/s", flush=True)
