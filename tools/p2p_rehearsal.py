#!/usr/bin/env python3
"""Diagnostic: the sharded resident launch (bb_p2p_*) at full C2 size with W processes ON ONE GPU (gloo carries
handles and votes; a one-GPU box has no xGMI peer).  Every rank gets 256 / W tiles so that all ranks' grids are resident
together.  Prints steps/s and the deviation from the unsharded run.   python tools/p2p_rehearsal.py [W] [steps] [native]"""
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, steps, out):
    import torch.distributed as dist
    import barbay_jl_amd as bb
    from barbay_jl_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = synth.fitness_normal(50_000, 8, 42)
    e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, device=0, rank=rank, world_size=world)
    on = bb.dist.setup_p2p(e)
    if rank == 0:
        print("resident multi-rank launch:", on, e.stats(), flush=True)
    if not on:
        return
    e.run(200)
    dist.barrier()
    t0 = time.perf_counter()
    e.run(steps)
    dist.barrier()
    dt = time.perf_counter() - t0
    mean, sigma = bb.dist.gather_posterior(e, wl.kind, wl.n_neutral, wl.n_bc, [8], 1, 1)
    if rank == 0:
        print(f"{world} ranks on one GPU: {steps / dt:.1f} steps/s ({dt / steps * 1e6:.2f} us/step)", flush=True)
        np.savez(out, mean=mean, sigma=sigma)
    dist.barrier()
    e.close()


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    native = len(sys.argv) > 3 and sys.argv[3] == "native"      # the geometry a real W-GPU run picks per rank (small tiles, many per CU here)
    if not native:
        os.environ["BB_TUNE_NB"] = str(-(-(50_000 // W) // (256 // W)))
        os.environ["BB_TUNE_NTHR"] = "1024"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = "/tmp/p2p_rehearsal.npz"
    mp.spawn(worker, args=(W, port, steps, out), nprocs=W, join=True)
    if os.path.exists(out):
        import barbay_jl_amd as bb
        from barbay_jl_amd import synth
        os.environ.pop("BB_TUNE_NB", None)
        os.environ.pop("BB_TUNE_NTHR", None)
        wl = synth.fitness_normal(50_000, 8, 42)
        e = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42)
        e.run(200 + steps)
        m, s = e.posterior()
        got = np.load(out)
        print("max |mean - unsharded|", float(np.abs(got["mean"] - m).max()), " max |sigma - unsharded|", float(np.abs(got["sigma"] - s).max()))
