import sys, time
sys.path.insert(0, "/root/repo")
import barbay_jl_amd as bb
from barbay_jl_amd import synth
wl = synth.genotype_fitness_normal(200_000, 8, 5_000, 45)
e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, geno_idx=wl.geno_idx, seed=1)
e.run(100); t=time.perf_counter(); e.run(500); print("C5", 500/(time.perf_counter()-t), e.stats())
