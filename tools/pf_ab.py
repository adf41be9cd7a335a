#!/usr/bin/env python3
"""Diagnostic: steps/s of one workload under each BB_TUNE_PF choice (window slot fetch point), same library, same box.
   python tools/pf_ab.py LIB WL [PF ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import _capi, synth  # noqa: E402
import __graft_entry__ as g  # noqa: E402

lib = _capi.load_library(os.path.join(g.PKG, "lib", "ab", sys.argv[1] + ".so"))
wl = getattr(synth, sys.argv[2])()
for rep in range(2):
    for pf in sys.argv[3:]:
        if pf == "-":
            os.environ.pop("BB_TUNE_PF", None)
        else:
            os.environ["BB_TUNE_PF"] = pf
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, _lib=lib)
        e.run(1000)
        e.run(4000)
        st = e.stats()
        print(f"pf={pf} k{st['resident_kernel']} P{st['persistent_pairs']} x{st['block_threads']} {4000 / st['last_run_ms'] * 1e3:9.1f} steps/s", flush=True)
        e.close()
