#!/usr/bin/env python3
"""Wall time of the whole `vi.advi` call (tidy frame in, result frame out) on a C2-shaped frame, with the engine's share."""
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import synth  # noqa: E402

wl = synth.fitness_normal(50_000, 8, 42)
c = wl.counts[0]
T, B = c.shape
ids = np.array([f"bc{i:06d}" for i in range(B)])
df = pd.DataFrame({"barcode": np.repeat(ids, T), "time": np.tile(np.arange(T), B), "count": c.T.reshape(-1),
                   "neutral": np.repeat(np.arange(B) < wl.n_neutral, T)})
for it in range(2):
    t0 = time.perf_counter()
    out = bb.vi.advi(data=df, model=bb.model.fitness_normal, advi=bb.vi.ADVI(1, 10_000), verbose=False)
    print(f"vi.advi pass {it}: {time.perf_counter() - t0:.2f} s for 10 000 iterations, {len(out)} rows", flush=True)
t0 = time.perf_counter()
arr = bb.utils.data_to_arrays(df)
t1 = time.perf_counter()
bm = bb.model.fitness_normal(arr.bc_count, arr.bc_total, arr.n_neutral, arr.n_bc)
q = bb.vi.vi(bm, bb.vi.ADVI(1, 10_000))
t2 = time.perf_counter()
print(f"data_to_arrays {t1 - t0:.2f} s, engine create + 10 000 steps + posterior {t2 - t1:.2f} s")
