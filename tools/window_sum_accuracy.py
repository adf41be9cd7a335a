#!/usr/bin/env python3
"""How far the compensated running window sum (engine: bb_opt_apply; oracle port: the same arithmetic, bit-equal in the trajectory
tests) is from the reference's `sum(g2)` over the window after a LONG run -- measured on the accumulator itself, not on the
trajectory: single-sample ADVI amplifies any rounding-level difference between two runs to the iterates' wander within a few
thousand steps, so mu(running) - mu(exact) at 10 000 steps says nothing about the sum's accuracy.
   python tools/window_sum_accuracy.py [steps] [barcodes]      (CPU only: the C port)"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from barbay_jl_amd import synth  # noqa: E402
from oracle import advi, port  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000
W = 100
wl = synth.fitness_normal(B, 8, 42)
sp = port.spec_from_workload(wl)
m0, o0 = advi.meanfield_init(42, sp.D)
p = port.Port(sp)
out = {"workload": wl.name, "window": W, "checkpoints": []}
state = None
mu, om = m0, o0
done = 0
for upto in sorted({100, 1_000, steps // 2, steps}):
    mu, om, _, state = p.run(mu, om, upto - done, seed=42, first_step=done, state=state, window=W, nthreads=port.usable_cores())
    done = upto
    D2 = 2 * sp.D
    hist = state[:W * D2].reshape(W, D2)
    acc, lo = state[W * D2:(W + 1) * D2], state[(W + 1) * D2:(W + 2) * D2]
    exact = np.array([math.fsum(hist[:, j]) for j in range(D2)])          # correctly rounded sum of the window's squares
    naive = hist.sum(axis=0)                                              # the reference's own left-to-right sum(g2) rounds too
    rel = np.abs(acc - exact) / np.maximum(exact, 1e-300)
    rel_pair = np.abs((acc + lo) - exact) / np.maximum(exact, 1e-300)
    rel_ref = np.abs(naive - exact) / np.maximum(exact, 1e-300)
    out["checkpoints"].append({"steps": upto, "max_rel_err_running_sum": float(rel.max()), "max_rel_err_running_pair": float(rel_pair.max()),
                               "max_rel_err_of_a_plain_sum_over_the_window": float(rel_ref.max()),
                               "max_rel_change_of_the_step_size": float((np.abs(1 / (40 + np.sqrt(acc)) - 1 / (40 + np.sqrt(exact))) * (40 + np.sqrt(exact))).max())})
print(json.dumps(out, indent=1))
