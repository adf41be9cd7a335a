#!/usr/bin/env python3
"""Diagnostic A/B: steps/s of the C2 (or B=, T=) workload for every library variant given on the command line.
Variants are built with  python tools/ab_persist.py --build NAME -DFLAG=..  into barbay.jl_amd/lib/ab/NAME.so"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

AB = os.path.join(g.PKG, "lib", "ab")
if len(sys.argv) > 2 and sys.argv[1] == "--build":
    os.makedirs(AB, exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", *g.HIP_FLAGS,
                    *sys.argv[3:], g.SRC, "-o", os.path.join(AB, sys.argv[2] + ".so"), "-ldl"], check=True)
    sys.exit(0)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import _capi, synth  # noqa: E402

WL = os.environ.get("WL", "fitness_normal")          # or replicate_fitness_normal / multienv_fitness_normal (their C3 / C4 sizes)
wl = synth.fitness_normal(int(os.environ.get("B", 50000)), int(os.environ.get("T", 8)), 42) if WL == "fitness_normal" else getattr(synth, WL)()
names = sys.argv[1:] or sorted(f[:-3] for f in os.listdir(AB) if f.endswith(".so"))
for rep in range(2):
    for name in names:
        path = g.LIB if name == "default" else os.path.join(AB, name + ".so")
        lib = _capi.load_library(path)
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, _lib=lib)
        e.run(2000)
        e.run(8000)
        ms = e.stats()["last_run_ms"]
        m, s = e.posterior()
        print(f"{name:24s} {8000 / ms * 1e3:10.1f} steps/s   ({ms / 8:.3f} us/step)  checksum {float(m.sum()):.12g}", flush=True)
        e.close()
