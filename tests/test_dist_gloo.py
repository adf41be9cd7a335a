"""world_size-2 run of the sharded path on CPU: torch.distributed (gloo) carries the one exchange per MC
sample (K moment rows) between two ranks, each driving its barcode shard through the C ABI's split-phase
calls on the host emulation of the block programs; the result must equal the unsharded run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, emu_path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch.distributed as dist
    import barbay_jl_amd as bb
    from barbay_jl_amd import _capi
    import _cases as c
    from conftest import make_engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _capi._declare(ctypes.CDLL(emu_path))
    sp = c.synth(case, seed=4)
    e = make_engine(sp, lib, seed=5, samples_per_step=2, window=4, rank=rank, world_size=world)
    bb.dist.run_external(e, 5)
    mean, sigma = bb.dist.gather_posterior(e, sp.kind, sp.n_neutral, sp.n_bc, sp.n_time, sp.n_rep, sp.n_env)
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mean=mean, sigma=sigma)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["fitness_multi_tile", "replicate_ragged"])
def test_two_ranks_match_one(emu_lib, tmp_path, case):
    import __graft_entry__ as g
    import _cases as c
    from conftest import make_engine
    sp = c.synth(case, seed=4)
    with make_engine(sp, emu_lib, seed=5, samples_per_step=2, window=4) as e1:
        e1.run(5)
        m1, s1 = e1.posterior()
    mp.spawn(_worker, args=(2, _free_port(), case, g.EMU, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")
    assert np.abs(got["mean"] - m1).max() < 1e-10 and np.abs(got["sigma"] - s1).max() < 1e-10


def _worker_one_fails(rank, world, port, emu_path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch.distributed as dist
    import barbay_jl_amd as bb
    from barbay_jl_amd import _capi
    import _cases as c
    from conftest import make_engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _capi._declare(ctypes.CDLL(emu_path))
    sp = c.synth("fitness_multi_tile", seed=4)
    e = make_engine(sp, lib, seed=5, window=4)       # (one unsharded engine per rank: only the agreement is under test)
    if rank == 1:
        mu, om = e.get_params()
        mu[3] = np.nan                                # this rank's run ends in BarBayNonFinite, rank 0's is clean
        e.set_params(mu, om)
    try:
        bb.dist.run(e, 3)
        said = "clean"
    except bb.BarBayNonFinite as err:
        said = f"own: {err}"
    except bb.BarBayHipError as err:
        said = f"told: {err}"
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(said)
    dist.barrier()                                    # both ranks are still in step: the next collective completes
    dist.destroy_process_group()


def test_a_failed_rank_is_reported_on_every_rank(emu_lib, tmp_path):
    """dist.run: rank 1 diverges, rank 0 does not -- both raise, and both reach the next collective (a rank-local
    BarBayNonFinite must not leave the clean ranks alone inside gather_posterior's all-gather)."""
    import __graft_entry__ as g
    mp.spawn(_worker_one_fails, args=(2, _free_port(), g.EMU, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (tmp_path / "rank0.txt").read_text(), (tmp_path / "rank1.txt").read_text()
    assert r0.startswith("told: rank 1 failed its run") and "NonFinite" in r0, r0
    assert r1.startswith("own:"), r1


def _worker_scattered_genotypes(rank, world, port, emu_path, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes
    import torch.distributed as dist
    import barbay_jl_amd as bb
    from barbay_jl_amd import _capi
    import _cases as c
    from conftest import make_engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BB_TUNE_NB"] = "24"
    os.environ["BB_TUNE_NTHR"] = "128"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _capi._declare(ctypes.CDLL(emu_path))
    sp = c.synth("genotype", seed=4)
    # the emulation steps the ranks of a resident sharded run in lock step inside ONE process: every process builds all ranks' handles,
    # runs them, and then gathers with ITS OWN rank's handle through torch.distributed -- gather_posterior is what is under test
    es = [make_engine(sp, lib, seed=5, window=4, resum_every=1, rank=r, world_size=world) for r in range(world)]
    handles = [e.p2p_export() for e in es]
    for e in es:
        e.p2p_import(handles)
    assert all(e.p2p_selftest() for e in es) and all(e.p2p_enable(True) for e in es)
    arr = (ctypes.c_void_p * world)(*[e._h for e in es])
    lib.bb_emu_run_group.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64]
    assert lib.bb_emu_run_group(arr, world, 7) == 0, lib.bb_last_error()
    mine = es[rank]
    assert not (mine.permutation() == np.arange(mine.D)).all()
    mean, sigma = bb.dist.gather_posterior(mine, sp.kind, sp.n_neutral, sp.n_bc, sp.n_time, sp.n_rep, sp.n_env)
    np.savez(os.path.join(out_dir, f"sharded{rank}.npz"), mean=mean, sigma=sigma)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_posterior_on_scattered_genotypes(emu_lib, tmp_path, monkeypatch):
    """bb_create regroups the genotype model's mutants on sharded handles too; shard_lo / shard_hi / geno_lo /
    geno_hi then describe the INTERNAL order while posterior() presents the caller's.  gather_posterior maps ownership through
    engine.permutation(): the 2-rank gather on data in the reference's order of appearance equals the single-rank result."""
    import __graft_entry__ as g
    import _cases as c
    from conftest import make_engine
    sp = c.synth("genotype", seed=4)
    monkeypatch.setenv("BB_TUNE_NB", "24")
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    with make_engine(sp, emu_lib, seed=5, window=4, resum_every=1, launch_mode=1) as e1:
        e1.run(7)
        m1, s1 = e1.posterior()
    mp.spawn(_worker_scattered_genotypes, args=(2, _free_port(), g.EMU, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"sharded{r}.npz")
        assert np.abs(got["mean"] - m1).max() < 1e-10 and np.abs(got["sigma"] - s1).max() < 1e-10
