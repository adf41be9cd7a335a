#!/usr/bin/env python3
"""Diagnostic: the driver's bench form (`--steps 20 --warmup 5`: ONE 20-step launch, the second launch of the process) against the same
launch repeated -- kernel time (the run's HIP events) and wall time of each launch in order, and (LIB = a -DBB_HOST_TIMES build:
python tools/xp.py build ht -DBB_HOST_TIMES) the host's share of bb_run split into enqueue / wait.
   WARM=5 N=20 python tools/launch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import barbay_jl_amd as bb
from barbay_jl_amd import synth, _capi
wl = synth.fitness_normal(50000, 8, 42)
_lib = _capi.load_library(os.environ["LIB"]) if os.environ.get("LIB") else None
W, N = int(os.environ.get("WARM", 5)), int(os.environ.get("N", 20))
SYNC = os.environ.get("TORCH_SYNC", "1") == "1"
for trial in range(int(os.environ.get("TRIALS", 3))):
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, _lib=_lib)
    if os.environ.get("SLEEP"):
        time.sleep(float(os.environ["SLEEP"]))
    e.run(W)
    if os.environ.get("REINIT"):          # the same launches from the initial state again: data or clocks?
        e.init_meanfield()
        e.run(int(os.environ["REINIT"]))
    out = []
    for rep in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(N)
        t1 = time.perf_counter()
        if SYNC:
            torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append((e.stats()["last_run_ms"] * 1e3, (t1 - t0) * 1e6, (t2 - t1) * 1e6))
    print(f"trial {trial} (warm-up {W}, then {N}-step launches in order): " + "  ".join(f"[kernel {k:6.1f} run {w:6.1f} +sync {s:4.1f}]" for k, w, s in out), flush=True)
    e.close()
