#!/usr/bin/env python3
"""steps/s of every BASELINE.json config on ONE GPU (configs 4 and 5 are specified sharded over 4 / 8 GPUs; this
is their single-GPU rate).  Prints one JSON line per config; not the driver's bench (that is bench.py on C2)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import synth  # noqa: E402

CFG = {
    "C1 fitness_normal 15x5 (reference fixture shape)": lambda: synth.fitness_normal(15, 5, 1, n_neutral=5),
    "C2 fitness_normal 50000x8": lambda: synth.fitness_normal(50_000, 8, 42),
    "C3 replicate_fitness_normal 20000x6x3": lambda: synth.replicate_fitness_normal(20_000, 6, 3, 43),
    "C4 multienv_fitness_normal 20000x6 E=4": lambda: synth.multienv_fitness_normal(20_000, 6, (1, 1, 2, 3, 4, 1), 44),
    "C5 genotype_fitness_normal 200000x8 G=5000": lambda: synth.genotype_fitness_normal(200_000, 8, 5_000, 45),
    "C5 as ONE of its 8 ranks: genotype_fitness_normal 25000x8 G=625 (200000/8 barcodes, 5000/8 genotypes)":
        lambda: synth.genotype_fitness_normal(25_000, 8, 625, 45),
    "(no BASELINE config) replicate_fitness_normal 80000x6x3, beyond the register file": lambda: synth.replicate_fitness_normal(80_000, 6, 3, 43),
    "C5 at about the largest size whose state fits one GPU's registers: genotype_fitness_normal 50000x8 G=1250": lambda: synth.genotype_fitness_normal(50_000, 8, 1_250, 45),
    "(no BASELINE config) multienv_replicate_fitness_normal 12000x(6,5,6) E=3": lambda: synth.multienv_replicate_fitness_normal(),
}
steps = int(os.environ.get("STEPS", 4000))


def st0_fast(wl):
    return wl.n_bc < 70_000          # (full C5 runs the two-kernel step, ~130 us each: fewer steps keep the script short)


for name, mk in CFG.items():
    if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
        continue
    wl = mk()
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=1)
    e.run(1200 if st0_fast(wl) else 300)          # past the ten early windows of the default re-add schedule (one exact re-add per window there)
    n = steps if st0_fast(wl) else max(200, steps // 4)
    t0 = time.perf_counter()
    e.run(n)
    dt = time.perf_counter() - t0
    st = e.stats()
    print(json.dumps({"config": name, "steps_per_s": round(n / dt, 1), "us_per_step": round(dt / n * 1e6, 2),
                      "n_latents": st["n_latents"], "bytes_per_step": st["bytes_per_step"],
                      "frac_hbm_peak": round(st["bytes_per_step"] * n / dt / 8e12, 4),
                      "resident_launch_pairs": st["persistent_pairs"], "resident_kernel": {0: "none (two kernels per step)", 1: "k_persist", 2: "k_res", 3: "k_stream"}[st["resident_kernel"]],
                      "kernel_instance": e.kernel_name(), "workgroups": st["n_blocks"], "threads": st["block_threads"]}),
          flush=True)
    e.close()
