import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import barbay_jl_amd as bb
from barbay_jl_amd import synth
for name, wl in (("fitness 50000x7", synth.fitness_normal(50_000, 7, 42)), ("fitness 50000x5", synth.fitness_normal(50_000, 5, 42)),
                 ("fifth model 12000x(6,5,6) E=3", synth.multienv_replicate_fitness_normal()), ("replicate 20000x5x3", synth.replicate_fitness_normal(20_000, 5, 3, 43))):
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=1)
    e.run(1200); e.run(3000)
    st = e.stats()
    print(f"{name}: kernel {st['resident_kernel']} P{st['persistent_pairs']} x{st['block_threads']}: {3000 / st['last_run_ms'] * 1e3:.1f} steps/s {st['last_run_ms'] / 3:.3f} us", flush=True)
    e.close()
