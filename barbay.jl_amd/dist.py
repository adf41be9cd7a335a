"""One-process-per-GPU driving of a sharded engine with `torch.distributed` as the plumbing.

* `init_rccl(engine)`       -- in-library RCCL: rank 0 makes the id, it is broadcast, every rank joins;
                               `engine.run(n)` then issues one ncclAllReduce of K doubles per MC sample.
* `setup_p2p(engine)`       -- the resident multi-GPU launch: IPC handles of the ranks' inboxes all-gathered, mapped,
                               probed, switched on only if EVERY rank can (else the RCCL path stays); returns the verdict.
* `run(engine, n)`          -- `engine.run(n)` on every rank, then ONE agreement on the outcome: an error any rank saw (a timed-out
                               exchange and the non-finite flag are rank-local) is raised on EVERY rank, so that nobody walks
                               into the next collective (`gather_posterior`) alone.
* `run_external(engine, n)` -- the same step with the reduction done by `torch.distributed.all_reduce`
                               on the host buffer (any backend; this is what the gloo tests drive).
* `gather_posterior(...)`   -- full (mean, sigma) on every rank from the per-rank shards.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np



def init_rccl(engine) -> None:
    import torch.distributed as dist
    ids = [engine.make_comm_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    engine.comm_init(ids[0])


def setup_p2p(engine) -> bool:
    """bb_p2p_export / import / selftest / enable with the two votes in between (include/barbay_hip.h).  Any backend:
    only Python objects travel.  True: every rank's `engine.run` now is ONE resident launch per call, exchanging over
    peer-mapped memory (xGMI between GPUs); False: nothing changed."""
    import torch.distributed as dist
    from ._capi import BarBayHipError
    world = dist.get_world_size()

    def vote(ok: bool) -> bool:
        votes = [None] * world
        dist.all_gather_object(votes, bool(ok))
        return all(votes)

    try:
        handle = engine.p2p_export()
    except BarBayHipError:
        handle = None
    handles = [None] * world
    dist.all_gather_object(handles, handle)
    if any(h is None for h in handles):
        return False
    ok = True
    try:
        engine.p2p_import(handles)
    except BarBayHipError:
        ok = False
    if not vote(ok):
        return False
    try:
        ok = engine.p2p_selftest()
    except BarBayHipError:
        ok = False
    if not vote(ok):
        return False
    try:
        ok = engine.p2p_enable(True)
    except BarBayHipError:
        ok = False
    # the two resident kernels speak different inbox protocols (k_res: tagged entries, k_persist: rows + ready words): every rank the same one
    kinds = [None] * world
    dist.all_gather_object(kinds, int(engine.stats()["resident_kernel"]) if ok else -1)
    if not vote(ok and all(k == kinds[0] for k in kinds)):
        try:
            engine.p2p_enable(False)
        except BarBayHipError:
            pass
        return False
    return True


def run(engine, n_steps: int) -> None:
    import torch.distributed as dist
    from ._capi import BarBayHipError
    err = None
    try:
        engine.run(n_steps)
    except BarBayHipError as e:          # (the library itself has taken its own collectives before it reported: bb_run)
        err = e
    said = [None] * dist.get_world_size()
    dist.all_gather_object(said, None if err is None else f"{type(err).__name__}: {err}")
    if err is not None:
        raise err
    for r, msg in enumerate(said):
        if msg is not None:
            raise BarBayHipError(f"rank {r} failed its run ({msg}); this rank's own run was clean")


def run_external(engine, n_steps: int) -> None:
    import torch
    import torch.distributed as dist
    for _ in range(n_steps * engine.samples_per_step):
        t = torch.from_numpy(engine.step_moments())
        if dist.get_backend() == "nccl":
            g = t.cuda()
            dist.all_reduce(g)
            t = g.cpu()
        else:
            dist.all_reduce(t)
        engine.step_apply(t.numpy())


def gather_posterior(engine, kind: str, n_neutral: int, n_bc: int, n_time: Sequence[int], n_rep: int = 1,
                     n_env: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    import torch.distributed as dist
    mean, sigma = engine.posterior()
    # the library names the entries this rank owns, in the CALLER's order (bb_get_owned): shard and genotype ranges live in the handle's
    # internal order (mutants regrouped by genotype, loglambda in front of theta where that keeps its pairs aligned)
    ix = engine.owned()
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (ix, mean[ix], sigma[ix]))
    for i, m, s in parts:
        mean[i] = m
        sigma[i] = s
    return mean, sigma
