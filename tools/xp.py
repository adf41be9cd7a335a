#!/usr/bin/env python3
"""Experiment driver for kernel variants (diagnostic).
   python tools/xp.py build NAME [-DFLAG ...]     here: lib/ab/NAME.so and NAME_st.so (-DBB_STAMPS), only the C2 / C4 instances (-DBB_FAST_BUILD)
   python tools/xp.py run NAME [NAME ...]         on the GPU box: steps/s of C2 (WL=, B=, T= as tools/ab_geom.py) and the stamp shares"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

AB = os.path.join(g.PKG, "lib", "ab")
if sys.argv[1] == "build":
    os.makedirs(AB, exist_ok=True)
    name, flags = sys.argv[2], sys.argv[3:]
    SRC = os.environ.get("XP_SRC", g.SRC)          # (XP_SRC=/tmp/old/bb_engine.hip: an A/B build of another checkout's sources)
    ps = [subprocess.Popen(["/opt/rocm/bin/hipcc", *g.HIP_FLAGS, "-DBB_FAST_BUILD", *flags, *extra, "-I", os.path.join(ROOT, "include"), SRC, "-o", os.path.join(AB, name + suf + ".so"), "-ldl"])
          for suf, extra in (("", []), ("_st", ["-DBB_STAMPS"]))]
    sys.exit(max(p.wait() for p in ps))

import barbay_jl_amd as bb  # noqa: E402
from barbay_jl_amd import _capi, synth  # noqa: E402

WL = os.environ.get("WL", "fitness_normal")
if WL == "fitness_normal":
    wl = synth.fitness_normal(int(os.environ.get("B", 50000)), int(os.environ.get("T", 8)), 42)
elif WL == "genotype_fitness_normal" and os.environ.get("B"):          # (B=25000 G=626: the shape one rank of C5's 8-GPU run holds, plain instance)
    wl = synth.genotype_fitness_normal(int(os.environ["B"]), int(os.environ.get("T", 8)), int(os.environ.get("G", 626)), 45)
else:
    wl = getattr(synth, WL)()
names = sys.argv[2:]
PH = [(20, 21, "S"), (21, 23, "M"), (23, 24, "pub"), (24, 25, "X"), (25, 26, "F"), (26, 28, "G")]
# k_stream since round 4 (one pass over the state per step: the next sample is formed inside the G passes): a step runs M -> G-U
PH_STREAM = [(22, 23, "M"), (23, 24, "pub"), (24, 25, "X"), (25, 26, "F"), (26, 27, "G-L"), (27, 28, "G-U")]


def engine(lib):
    return bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, _lib=lib)


for rep in range(2):
    for name in names:
        e = engine(_capi.load_library(os.path.join(AB, name + ".so")))
        e.run(1000)
        e.run(4000)
        st = e.stats()
        ms = st["last_run_ms"]
        line = f"{name:14s} k{st['resident_kernel']} P{st['persistent_pairs']} x{st['block_threads']:4d} {4000 / ms * 1e3:9.1f} steps/s {ms / 4:7.3f} us"
        e.close()
        if rep == 1:
            e = engine(_capi.load_library(os.path.join(AB, name + "_st.so")))
            e.run(21)
            s = e.stamps().astype(np.int64)
            fused = st["resident_kernel"] == 3 and s[:, 22].any() and s[:, 27].any()
            tot = np.median(s[:, 28] - s[:, 22 if fused else 20])
            line += f" | span {tot:6.0f}: " + " ".join(f"{nm} {np.median(s[:, b] - s[:, a]):5.0f}" for a, b, nm in (PH_STREAM if fused else PH) if s[:, b].any() and s[:, a].any())
            lead = np.arange(s.shape[0]) < 8
            # (tagged rows: a leader polls its members' rows and sums them in one pass -- there is no "members seen" stamp any more; 18 = its
            #  group row is out, 1 = the group rows have been seen, 24 = the tile's own row is out)
            line += (f"\n     leaders: own row out -> group row out (poll members' rows + sum + store) {np.median(s[lead, 18] - s[lead, 24]):.0f}, "
                     f"then until the group rows are seen {np.median(s[lead, 1] - s[lead, 18]):.0f}; others: own row out -> group rows seen {np.median(s[~lead, 1] - s[~lead, 24]):.0f}; "
                     f"all: group rows seen -> totals in LDS {np.median(s[:, 25] - s[:, 1]):.0f}")
            # wall-clock stamps (100 MHz): publish and totals-seen over all tiles, in shader cycles at ~2.4 GHz
            for ev, nm in ((29, "publish"), (30, "totals seen")):
                tt = (s[:, ev] - s[:, ev].min()) * 24
                order = np.argsort(tt)
                line += (f"\n     {nm:12s} (wall clock, cycles after the first tile): median {np.median(tt):.0f} p90 {np.percentile(tt, 90):.0f} max {tt.max()}; "
                         f"last tiles {order[-5:].tolist()}; leaders {tt[:8].tolist()}")
            line += f"\n     publish(last tile) -> totals seen(first / median / last tile): {(s[:, 30].min() - s[:, 29].max()) * 24} / {(np.median(s[:, 30]) - s[:, 29].max()) * 24:.0f} / {(s[:, 30].max() - s[:, 29].max()) * 24}"
            if os.environ.get("WAVES"):
                w = e.stamps(per_wave=True).astype(np.int64)
                for b in [int(x) for x in os.environ["WAVES"].split(",")]:
                    t0 = s[b, 26]
                    line += (f"\n     tile {b} (relative to F-done of the last step; S of that step came before):"
                             + f"\n       S-start " + " ".join(f"{v:6d}" for v in (w[b, 1] - t0))
                             + f"\n       S-end   " + " ".join(f"{v:6d}" for v in (w[b, 0] - t0))
                             + f"\n       S len   " + " ".join(f"{v:6d}" for v in (w[b, 0] - w[b, 1]))
                             + f"\n       G-start " + " ".join(f"{v:6d}" for v in (w[b, 2] - t0))
                             + f"\n       G-end   " + " ".join(f"{v:6d}" for v in (w[b, 3] - t0))
                             + f"\n       G len   " + " ".join(f"{v:6d}" for v in (w[b, 3] - w[b, 2])))
            e.close()
        print(line, flush=True)
