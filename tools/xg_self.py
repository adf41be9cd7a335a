#!/usr/bin/env python3
"""Diagnostic: ONE rank's cross-GPU resident launch on a whole GPU -- world_size 1 with the inbox protocol switched on (the inbox is
the rank's own memory, it holds 8 group rows instead of 8 W): what a rank of a W-GPU run executes per step, minus the remote rows.
   python tools/xg_self.py [barcodes] [steps]        (C2's shape; 6250 = one rank of eight)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 6250
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
lib = None
if os.environ.get("LIB"):
    from barbay_jl_amd import _capi
    lib = _capi.load_library(os.environ["LIB"])
wl = synth.fitness_normal(B, 8, 42)
os.environ["BB_P2P_SELF"] = "1"
for xg in (False, True):
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, _lib=lib)
    if xg:
        e.p2p_import([e.p2p_export()])
        assert e.p2p_selftest() and e.p2p_enable(True)
    e.run(1000)
    e.run(steps)
    st = e.stats()
    m, s = e.posterior()
    print(f"{B} barcodes, {'inbox protocol (one rank of a sharded run)' if xg else 'one-GPU protocol'}: kernel {st['resident_kernel']} P{st['persistent_pairs']} x{st['block_threads']} "
          f"{st['n_blocks']} tiles, {steps / st['last_run_ms'] * 1e3:.1f} steps/s ({st['last_run_ms'] / steps * 1e3:.2f} us/step), mean[0] {m[0]:.12f}", flush=True)
    e.close()
