#!/usr/bin/env python3
"""Diagnostic A/B: steps/s of C2 (or WL=...) for several launch geometries / kernels in one process.
   python tools/ab_geom.py 1024 512 256 old1024      (BB_TUNE_NTHR values; a leading `old` keeps k_persist)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import synth  # noqa: E402

WL = os.environ.get("WL", "fitness_normal")
wl = synth.fitness_normal(int(os.environ.get("B", 50000)), int(os.environ.get("T", 8)), 42) if WL == "fitness_normal" else getattr(synth, WL)()
steps = int(os.environ.get("STEPS", 4000))
for rep in range(2):
    for spec in sys.argv[1:] or ["1024"]:
        old = spec.startswith("old")
        nthr = spec[3:] if old else spec
        os.environ["BB_NO_RES"] = "1" if old else "0"
        os.environ["BB_TUNE_NTHR"] = nthr
        try:
            e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42)
        except bb.BarBayHipError as err:
            print(f"{spec:10s} not possible: {err}")
            continue
        e.run(1000)
        e.run(steps)
        st = e.stats()
        ms = st["last_run_ms"]
        m, s = e.posterior()
        print(f"{spec:10s} kernel {st['resident_kernel']} pairs {st['persistent_pairs']} threads {st['block_threads']} lds {st['lds_bytes']:6d}  "
              f"{steps / ms * 1e3:10.1f} steps/s  ({ms / steps * 1e3:.3f} us/step)  checksum {float(m.sum()):.12g}", flush=True)
        e.close()
