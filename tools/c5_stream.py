#!/usr/bin/env python3
"""Diagnostic: BASELINE config 5 (genotype_fitness_normal 200 000 x 8, 5 000 genotypes) on ONE GPU -- k_stream (bb_stream.h) against the
two-kernel step (BB_NO_STREAM=1): us per step, algorithmic GB/s, and equality of the two after 61 steps.   LIB=path: an A/B build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import barbay_jl_amd as bb
from barbay_jl_amd import synth, _capi
lib = _capi.load_library(os.environ["LIB"]) if os.environ.get("LIB") else None
B, G = int(os.environ.get("B", 200000)), int(os.environ.get("G", 5000))
wl = synth.genotype_fitness_normal(B, 8, G, 45) if os.environ.get("WL", "genotype") == "genotype" else synth.fitness_normal(B, 8, 42)
outs = {}
for nostream in ("0", "1"):
    os.environ["BB_NO_STREAM"] = nostream
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, geno_idx=wl.geno_idx, seed=7, _lib=lib)
    e.run(61)
    outs[nostream] = e.get_params()
    st = e.stats()
    n = 300
    e.run(n)
    st = e.stats()
    us = st["last_run_ms"] * 1e3 / n
    print(f"BB_NO_STREAM={nostream}: resident_kernel {st['resident_kernel']} pairs/thread {st['persistent_pairs']} blocks {st['n_blocks']} x {st['block_threads']} lds {st['lds_bytes']}  "
          f"{us:8.2f} us/step  {1e6 / us:9.1f} steps/s  algorithmic {st['bytes_per_step'] / us / 1e3:7.1f} GB/s = {st['bytes_per_step'] / us / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    e.close()
d = max(np.abs(outs["0"][0] - outs["1"][0]).max(), np.abs(outs["0"][1] - outs["1"][1]).max())
print(f"max |stream - two-kernel| after 61 steps: {d:.3e}; finite: {bool(np.isfinite(outs['0'][0]).all())}")
