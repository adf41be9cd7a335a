#!/usr/bin/env python3
"""Diagnostic: fixed cost of a bb_run (kernel prologue / epilogue + host) -- kernel time (HIP events) and wall time for n steps.
(Tried on the host side and dropped, gpurun_out r02e: polling hipEventQuery instead of hipStreamSynchronize -- no change; the
run's timing events attached to the launch with hipExtLaunchKernel -- event time -3 us, wall time +8 us.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import barbay_jl_amd as bb
from barbay_jl_amd import synth
wl = synth.fitness_normal(50000, 8, 42)
from barbay_jl_amd import _capi
_lib = _capi.load_library(os.environ["LIB"]) if os.environ.get("LIB") else None        # (an A/B build: tools/xp.py build NAME)
e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, _lib=_lib)
e.run(1200)
for n in (1, 2, 5, 10, 20, 40, 80):
    ks, ws = [], []
    for rep in range(7):
        t0 = time.perf_counter(); e.run(n); ws.append((time.perf_counter() - t0) * 1e6); ks.append(e.stats()["last_run_ms"] * 1e3)
    print(f"n {n:3d}: kernel {np.median(ks):8.1f} us  wall {np.median(ws):8.1f} us   per step {np.median(ks) / n:7.2f} / {np.median(ws) / n:7.2f}")

# where a launch's fixed time goes: wall-clock stamps (100 MHz) of the -DBB_STAMPS build (python tools/xp.py build base)
st_lib = os.path.join(ROOT, "barbay.jl_amd", "lib", "ab", "base_st.so")
if os.path.exists(st_lib) and not os.environ.get("NO_STAMPS"):
    e.close()
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, _lib=_capi.load_library(st_lib))
    e.run(1200)
    for n in (1, 20):
        e.run(n)
        s = e.stamps().astype(np.int64)
        t0 = s[:, 2].min()
        f = lambda col: f"{(np.median(s[:, col]) - t0) / 100:7.2f} (last tile {(s[:, col].max() - t0) / 100:7.2f})"
        print(f"n {n:3d} stamped launch, us after the first tile's entry: all tiles entered {(s[:, 2].max() - t0) / 100:.2f}; segment table built {f(7)}; met {f(8)}; LDS tables {f(9)}; descriptors + loads issued {f(10)}; prologue done {f(3)}; "
              f"first normals drawn {f(4)}; loop done {f(5)}; epilogue done {f(6)}; kernel time {e.stats()['last_run_ms'] * 1e3:.1f}")
