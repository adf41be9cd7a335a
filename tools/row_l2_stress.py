#!/usr/bin/env python3
"""Long runs with the same-XCD row stores on and off (BB_TUNE_ROW_L2): the parameters after N steps must be the same bits -- an entry that
reached the leader stale or torn even once in N x tiles hand-offs would show.   python tools/row_l2_stress.py [steps]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import barbay_jl_amd as bb
from barbay_jl_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
for wl, n in ((synth.fitness_normal(50000, 8, 42), N), (synth.replicate_fitness_normal(), N // 2), (synth.multienv_fitness_normal(), N // 2),
              (synth.genotype_fitness_normal(), N // 8)):
    out = []
    for sw in ("1", "0"):
        os.environ["BB_TUNE_ROW_L2"] = sw
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42)
        t0 = time.time()
        try:
            e.run(n)
        except bb._capi.BarBayNonFinite:
            pass                      # (a divergent trajectory diverges the same way twice)
        st = e.stats()
        mu, om = e.get_params()
        out.append((hashlib.sha256(mu.tobytes() + om.tobytes()).hexdigest()[:16], st["rows_same_xcd"], st["n_blocks"], time.time() - t0, e.kernel_name()))
        e.close()
    print(f"{wl.name}: {n} steps, {out[0][4]}: on: {out[0][1]} of {out[0][2]} tiles through L2, sha {out[0][0]} ({out[0][3]:.1f} s) | off: {out[1][1]} tiles, sha {out[1][0]} "
          f"({out[1][3]:.1f} s) -> {'SAME BITS' if out[0][0] == out[1][0] else 'DIFFERENT'}", flush=True)
    assert out[0][0] == out[1][0]
