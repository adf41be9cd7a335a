#!/usr/bin/env python3
"""Diagnostic: where a step's cycles go.  Builds barbay.jl_amd/lib/libbarbay_hip_stamps.so with
-DBB_STAMPS (s_memtime at every pass boundary), runs the C2 workload a few steps and prints the
median per-pass cycle shares over workgroups.  Read SHARES, not lengths (the stamps perturb)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

out = os.path.join(g.PKG, "lib", "libbarbay_hip_stamps.so")
stale = not os.path.exists(out) or any(os.path.getmtime(os.path.join(os.path.dirname(g.SRC), f)) > os.path.getmtime(out)
                                        for f in os.listdir(os.path.dirname(g.SRC)))
if "--build-only" in sys.argv or stale:
    subprocess.run(["/opt/rocm/bin/hipcc", *g.HIP_FLAGS, "-DBB_STAMPS", g.SRC, "-o", out, "-ldl"], check=True)
    if "--build-only" in sys.argv:
        sys.exit(0)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import _capi, synth  # noqa: E402

lib = _capi.load_library(out)
WL = os.environ.get("WL", "fitness_normal")          # or replicate_fitness_normal / multienv_fitness_normal (C3 / C4 sizes)
wl = synth.fitness_normal(int(os.environ.get("B", 50000)), int(os.environ.get("T", 8)), 42) if WL == "fitness_normal" else getattr(synth, WL)()
mode = int(os.environ.get("MODE", 1))
e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, steps_per_graph=-1, launch_mode=mode, optimizer=os.environ.get("OPT", "TruncatedADAGrad"), _lib=lib)
e.run(21)
st = e.stamps().astype(np.int64)
names = {1: "S draw", 2: "E tables", 3: "M accumulate", 4: "M reduce", 5: "row_sum", 6: "write partials",
         9: "F sum rows", 10: "F finish", 11: "stage z", 12: "E tables", 13: "R residuals", 14: "U unit sums",
         15: "G gather+update", 16: "tail"}
print(e.stats())
names.update({21: "S draw", 22: "E tables", 23: "M moments", 24: "publish", 25: "GRID BARRIER", 26: "F totals+finish", 27: "R/U", 28: "G gather+update"})
for lo, hi, title in (((0, 6, "k_sample"), (8, 16, "k_update")) if mode == 1 else ((20, 28, "k_persist step"),)):
    tot = np.median(st[:, hi] - st[:, lo])
    print(f"{title}: median block span {tot:.0f} cycles (s_memtime ticks)")
    prev = lo
    for i in range(lo + 1, hi + 1):
        if i not in names or not st[:, i].any():
            continue
        d = np.median(st[:, i] - st[:, prev])
        dd = st[:, i] - st[:, prev]
        print(f"   {names[i]:16s} {d:9.0f}  {100 * d / tot:5.1f}%   min {dd.min():7d}  p90 {int(np.percentile(dd, 90)):7d}  max {dd.max():7d}")
        prev = i
    if mode == 2:
        for a, b, nm in ((27, 29, "G: enter -> pair_of"), (29, 30, "G: glik + prior"), (30, 31, "G: optimiser math"), (31, 19, "G: window stores drained"), (19, 28, "G: wait for other waves")):
            print(f"      {nm:28s} {np.median(st[:, b] - st[:, a]):9.0f}")
    if mode == 2:
        lead = np.arange(st.shape[0]) < 8
        # (k_res / k_stream with tagged rows: no drain, no meet, no "members seen": 24 = own row out, 18 = leader's group row out, 1 = group rows seen)
        print(f"      leaders: own row out -> group row out {np.median(st[lead, 18] - st[lead, 24]):.0f}  then until group rows seen {np.median(st[lead, 1] - st[lead, 18]):.0f}")
        print(f"      others : own row out -> group rows seen {np.median(st[~lead, 1] - st[~lead, 24]):.0f} (min {(st[~lead, 1] - st[~lead, 24]).min()})   group rows seen -> totals in LDS {np.median(st[:, 25] - st[:, 1]):.0f}")
        arr = st[:, 24] - st[:, 24].min(); rel = st[:, 25] - st[:, 24].min()
        print(f"      arrival spread over tiles: median {np.median(arr):.0f} p90 {np.percentile(arr, 90):.0f} max {arr.max()}   release after first arrival: median {np.median(rel):.0f} max {rel.max()}")
        late = np.argsort(arr)[-5:]
        print("      latest tiles", late.tolist(), "their draw spans", (st[late, 21] - st[late, 20]).tolist(), "moments", (st[late, 23] - st[late, 22]).tolist())
    print(f"   first block start -> last block end: {st[:, hi].max() - st[:, lo].min()} ticks")
