#!/usr/bin/env python3
"""Diagnostic: the replicate model on one GPU at sizes around C3's, 1024 threads x 1 pair slot against 512 threads x 2 / 3 slots
(BB_TUNE_NTHR) -- what a pair-slot layout that fits C3 into 1024 threads would be worth.
   python tools/c3_geometry.py [B ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb
from barbay_jl_amd import synth

for B in [int(x) for x in sys.argv[1:]] or [16000, 17000, 18000, 20000]:
    wl = synth.replicate_fitness_normal(B, 6, 3, 43)
    for nthr, stream in ((1024, 0), (512, 0), (1024, 1)):
        os.environ["BB_TUNE_NTHR"] = str(nthr)
        os.environ["BB_TUNE_STREAM"] = str(stream)          # (1: the streaming-resident launch on a shape the register file holds)
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42)
        e.run(1000)
        e.run(4000)
        st = e.stats()
        us = st["last_run_ms"] / 4
        D = e.D
        by = 96 * D + 4 * sum(c.size for c in wl.counts)
        print(f"B {B:6d} nthr {nthr:4d}: {e.kernel_name():44s} pairs/thread {st['persistent_pairs']} tiles {st['n_blocks']} {us:7.3f} us/step  {by / us / 1e3:7.1f} GB/s = {by / us / 1e3 / 8000:.3f}   ns per 1000 latents {us * 1e6 / D:.2f}", flush=True)
        e.close()
