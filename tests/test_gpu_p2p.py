"""GPU: the sharded resident launch (bb_p2p_*) rehearsed with TWO PROCESSES ON ONE GPU: each rank drives its barcode shard
with one resident launch, the group rows cross through IPC-mapped fine-grained inboxes (what xGMI peers would map), gloo
carries the handles and the votes.  The result must equal the unsharded run.  (A one-GPU box cannot show the xGMI hop
itself; the protocol, the handles, the inbox layout and both kernels' co-residency are what this covers.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, steps, out_dir, ms=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import barbay_jl_amd as bb
    import _cases as c
    from conftest import make_engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sp = c.synth(case, seed=4)
    ekw = dict(samples_per_step=2, elbo_every=1) if ms else {}
    e = make_engine(sp, None, seed=5, window=4, resum_every=1, rank=rank, world_size=world, device=0, **ekw)
    on = bb.dist.setup_p2p(e)
    st0 = e.stats()
    for _ in range(2):            # the second pass restarts from the initial state with the inboxes still holding the first's words
        e.init_meanfield()
        e.run(3)
        e.run(steps - 3)
    mean, sigma = bb.dist.gather_posterior(e, sp.kind, sp.n_neutral, sp.n_bc, sp.n_time, sp.n_rep, sp.n_env)
    st = e.stats()
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mean=mean, sigma=sigma, on=on, pairs=st["persistent_pairs"],
                 launches=st["launches_last_run"], blocks=st0["n_blocks"], name=e.kernel_name(),
                 trace=e.elbo_trace(0, steps) if ms else np.zeros(0))
    dist.barrier()
    e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["fitness_multi_tile", "replicate_ragged", "multienv", "fitness_T6", "multienv_T8", "genotype_runs"])
def test_two_processes_one_gpu(hip_lib, tmp_path, monkeypatch, case):
    """(the first three run k_persist's cross-GPU form, the others k_res's; genotype_runs: shards cut at genotype boundaries, every
    rank owns its genotypes' theta -- no communicator here, so the gather takes theta from its owners)"""
    import _cases as c
    from conftest import make_engine
    monkeypatch.setenv("BB_TUNE_NB", "24" if case == "genotype_runs" else "16")   # >= 8 tiles per rank; both ranks' small grids fit the one GPU together
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    steps = 9
    sp = c.synth(case, seed=4)
    with make_engine(sp, hip_lib, seed=5, window=4, resum_every=1, launch_mode=1) as e1:
        e1.run(steps)
        m1, s1 = e1.posterior()
    mp.spawn(_worker, args=(2, _free_port(), case, steps, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")
    assert bool(got["on"]) and int(got["pairs"]) == 1 and int(got["launches"]) == 1     # the resident launch really ran
    assert np.abs(got["mean"] - m1).max() < 1e-9 and np.abs(got["sigma"] - s1).max() < 1e-9


@pytest.mark.parametrize("case", ["fitness_T6", "genotype_runs"])
def test_two_processes_one_gpu_several_samples_and_elbo_trace(hip_lib, tmp_path, monkeypatch, case):
    """`Turing.ADVI(2, ...)` (src/vi.jl:98) with the ELBO recorded every step on a SHARDED run: the resident launch's MS + cross-GPU
    instance (k_res<.., true, 0, .., true>; round 3 dropped such runs to two kernels + RCCL) -- posterior and trace equal the unsharded run's."""
    import _cases as c
    from conftest import make_engine
    monkeypatch.setenv("BB_TUNE_NB", "24" if case == "genotype_runs" else "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    steps = 9
    sp = c.synth(case, seed=4)
    with make_engine(sp, hip_lib, seed=5, window=4, resum_every=1, launch_mode=1, samples_per_step=2, elbo_every=1) as e1:
        e1.run(steps)
        m1, s1 = e1.posterior()
        t1 = e1.elbo_trace(0, steps)
    mp.spawn(_worker, args=(2, _free_port(), case, steps, str(tmp_path), True), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")
    assert bool(got["on"]) and int(got["pairs"]) == 1 and int(got["launches"]) == 1
    assert str(got["name"]).startswith("k_res<") and str(got["name"]).endswith(",true>") and ",true,0," in str(got["name"]), str(got["name"])
    assert np.abs(got["mean"] - m1).max() < 1e-9 and np.abs(got["sigma"] - s1).max() < 1e-9
    assert np.abs(got["trace"] - t1).max() <= 1e-10 * np.abs(t1).max()
