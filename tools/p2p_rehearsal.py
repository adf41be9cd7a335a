#!/usr/bin/env python3
"""Diagnostic: the sharded resident launch (bb_p2p_*) with W processes ON ONE GPU (gloo carries handles and votes; a one-GPU
box has no xGMI peer, the "remote" inboxes are local memory).  All W grids have to be resident together on the one GPU, so every
rank runs 256 / W tiles.  Two sizings:

  WL=c2 (default)      the full C2 problem over W x (256 / W) tiles
  WL=c4 | c5 | c2r     the TILE SHAPE of one rank of the real W-GPU run (BASELINE configs 4 / 5 on 4 / 8 GPUs, C2 on W GPUs): the
                       rank's own geometry is read from a one-rank engine of the shard's size, and the rehearsed problem is
                       W x (256 / W) tiles of that shape -- so the kernel instance is the one the real rank would run

SHAPE_W=8 takes the tile shape of one rank of 8 while W (<= 4: the box allows 6 processes on its card) processes rehearse it.
Prints the kernel instance, steps/s, and the deviation from the unsharded run.   python tools/p2p_rehearsal.py [W] [steps]"""
import os
import socket
import sys
import time

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WL = os.environ.get("WL", "c2")
KERNEL = {0: "none (two kernels per step)", 1: "k_persist", 2: "k_res", 3: "k_stream"}


def full_workload(B=None):
    from barbay_jl_amd import synth
    if WL in ("c2", "c2r"):
        return synth.fitness_normal(B or 50_000, 8, 42)
    if WL == "c4":
        return synth.multienv_fitness_normal(B or 20_000, 6, (1, 1, 2, 3, 4, 1), 44)
    if WL == "c5":
        B = B or 200_000
        return synth.genotype_fitness_normal(B, 8, max(2, B // 40), 45)
    raise SystemExit(f"WL={WL}?")


def engine(bb, wl, **kw):
    return bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, **kw)


def plan(world, q):
    """(in a child of its own: the parent never touches the GPU before it spawns)  -> barcodes of the rehearsed problem, NB, threads"""
    import barbay_jl_amd as bb
    full = full_workload()
    B = full.n_neutral + full.n_bc
    if WL == "c2":
        q.put((B, -(-(B // world) // (256 // world)), 1024, "full C2"))
        return
    real = int(os.environ.get("SHAPE_W", world))       # ranks of the real run whose tile shape is rehearsed (a one-GPU box allows 6 processes on its card: W <= 4 here)
    shard = full_workload(B // real)
    e = engine(bb, shard)
    st = e.stats()
    e.close()
    nb = -(-(B // real) // st["n_blocks"])
    q.put((nb * (256 // world) * world, nb, st["block_threads"],
           f"one rank of {real}: {B // real} barcodes, {KERNEL[st['resident_kernel']]} P{st['persistent_pairs']} x{st['block_threads']}, {st['n_blocks']} tiles of {nb} barcodes"))


def worker(rank, world, port, steps, out, B):
    import torch.distributed as dist
    import barbay_jl_amd as bb
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = full_workload(B)
    e = engine(bb, wl, device=0, rank=rank, world_size=world)
    on = bb.dist.setup_p2p(e)
    st = e.stats()
    if rank == 0:
        print(f"resident multi-rank launch: {on}; rank 0 runs {KERNEL[st['resident_kernel']]}<cross-GPU exchange> P{st['persistent_pairs']} x{st['block_threads']}, "
              f"{st['n_blocks']} tiles, {wl.n_neutral + wl.n_bc} barcodes over {world} ranks", flush=True)
    if not on:
        return
    e.run(200)
    dist.barrier()
    t0 = time.perf_counter()
    e.run(steps)
    dist.barrier()
    dt = time.perf_counter() - t0
    mean, sigma = bb.dist.gather_posterior(e, wl.kind, wl.n_neutral, wl.n_bc, [c.shape[0] for c in wl.counts], len(wl.counts),
                                           1 if wl.env_idx is None else int(np.max(wl.env_idx)) + 1)
    if rank == 0:
        print(f"{world} ranks on one GPU: {steps / dt:.1f} steps/s ({dt / steps * 1e6:.2f} us/step)", flush=True)
        np.savez(out, mean=mean, sigma=sigma)
    dist.barrier()
    e.close()


if __name__ == "__main__":
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=plan, args=(W, q))
    p.start()
    B, nb, nthr, what = q.get()
    p.join()
    print(f"WL={WL} W={W}: {what}; rehearsed problem {B} barcodes", flush=True)
    os.environ["BB_TUNE_NB"] = str(nb)
    os.environ["BB_TUNE_NTHR"] = str(nthr)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = "/tmp/p2p_rehearsal.npz"
    if os.path.exists(out):
        os.remove(out)
    mp.spawn(worker, args=(W, port, steps, out, B), nprocs=W, join=True)
    if os.path.exists(out):
        import barbay_jl_amd as bb
        os.environ.pop("BB_TUNE_NB", None)
        os.environ.pop("BB_TUNE_NTHR", None)
        e = engine(bb, full_workload(B))
        e.run(200 + steps)
        m, s = e.posterior()
        got = np.load(out)
        print("max |mean - unsharded|", float(np.abs(got["mean"] - m).max()), " max |sigma - unsharded|", float(np.abs(got["sigma"] - s).max()))
