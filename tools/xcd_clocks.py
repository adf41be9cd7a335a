#!/usr/bin/env python3
"""Diagnostic: do the XCDs run alike?  Per XCD (tile index modulo 8, the dispatcher's round-robin): the tiles' S / G pass lengths in shader
cycles (s_memtime), the shader clock from the two clocks stamped at the same two points (s_memtime against the 100 MHz s_memrealtime, publish ->
totals seen), and when the tiles publish / see the totals (100 MHz wall clock, relative to the first tile).  Needs lib/ab/base_st.so.
   WL=replicate_fitness_normal python tools/xcd_clocks.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import barbay_jl_amd as bb
from barbay_jl_amd import _capi, synth
WL = os.environ.get("WL", "fitness_normal")
wl = synth.fitness_normal(50000, 8, 42) if WL == "fitness_normal" else getattr(synth, WL)()
lib = _capi.load_library(os.path.join(ROOT, "barbay.jl_amd", "lib", "ab", "base_st.so"))
for trial in range(3):
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, _lib=lib)
    e.run(200 + 37 * trial)
    e.run(21)
    s = e.stamps().astype(np.int64)
    n = e.stats()["n_blocks"]
    s = s[:n]
    x = np.arange(n) % 8
    ord_ = np.arange(n) >= 8                      # (leaders hold smaller tiles)
    print(f"{wl.name}, trial {trial}: {n} tiles; per XCD (ordinary tiles): S cycles | G cycles | shader MHz (publish -> totals seen) | publish, totals seen (ns after the first tile, medians)")
    for k in range(8):
        m = (x == k) & ord_
        mhz = (s[m, 1] - s[m, 24]) / np.maximum(s[m, 30] - s[m, 29], 1) * 100.0
        print(f"   XCD {k}: {m.sum():3d} tiles  S {np.median(s[m, 21] - s[m, 20]):6.0f}  G {np.median(s[m, 28] - s[m, 26]):6.0f}  {np.median(mhz):7.1f} MHz   "
              f"publish {np.median(s[m, 29] - s[:, 29].min()) * 10:6.0f}  totals seen {np.median(s[m, 30] - s[:, 30].min()) * 10:6.0f}")
    e.close()
