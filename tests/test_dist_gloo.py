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
