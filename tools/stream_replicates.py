#!/usr/bin/env python3
"""Diagnostic: a replicate_fitness_normal problem beyond the register file (default 80 000 barcodes x 6 time points x 3 replicates) on ONE GPU --
k_stream (bb_stream.h) against the two-kernel step (BB_NO_STREAM=1): us per step, and equality of the two after 41 steps.
WL=multienv_replicate / B= / T= / R= select other shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import barbay_jl_amd as bb
from barbay_jl_amd import synth
B, T, R = int(os.environ.get("B", 80000)), int(os.environ.get("T", 6)), int(os.environ.get("R", 3))
wl = synth.replicate_fitness_normal(B, T, R, 43)
outs, us = {}, {}
for nostream in ("0", "1"):
    os.environ["BB_NO_STREAM"] = nostream
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=7)
    e.run(41)
    outs[nostream] = e.get_params()
    n = 300
    e.run(n)
    st = e.stats()
    us[nostream] = st["last_run_ms"] * 1e3 / n
    print(f"BB_NO_STREAM={nostream}: {e.kernel_name()} resident_kernel {st['resident_kernel']} pairs/thread {st['persistent_pairs']} blocks {st['n_blocks']} x {st['block_threads']} lds {st['lds_bytes']}  "
          f"{us[nostream]:8.2f} us/step  {1e6 / us[nostream]:9.1f} steps/s  algorithmic {st['bytes_per_step'] / us[nostream] / 1e3:7.1f} GB/s = {st['bytes_per_step'] / us[nostream] / 1e3 / 8000:.3f} of 8 TB/s", flush=True)
    e.close()
d = max(np.abs(outs["0"][0] - outs["1"][0]).max(), np.abs(outs["0"][1] - outs["1"][1]).max())
print(f"replicate_fitness_normal {B} x {T} x {R}: k_stream / two-kernel time = {us['0'] / us['1']:.3f}; max |stream - two-kernel| after 41 steps: {d:.3e}; finite: {bool(np.isfinite(outs['0'][0]).all())}")
