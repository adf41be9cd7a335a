#!/usr/bin/env python3
"""ELBO-gradient steps/s of the MI355X-native ADVI engine on BASELINE.json's headline workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one ADVI iteration (S = 1 reparameterised sample of the ELBO gradient + one
TruncatedADAGrad update of all 2D variational parameters) of `fitness_normal` on the synthetic
50 000 barcodes x 8 time points workload (config C2, seed 42), inputs resident in HBM.  With N > 1
the barcodes shard over the ranks (one process per GPU) and every step carries one RCCL all-reduce
of the K moment rows, so the job is strong-scaled: `value` is the step rate of the whole job.

Prints ONE JSON line on rank 0 with the driver's contract plus `roofline` (dominant kernel, HIP
events on the engine's stream) and `cpu_baseline` (the oracle's C port on the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(wl, seconds_budget: float = 20.0):
    """Time the oracle's fused C port (oracle/c/bb_port.c) on the same workload, same options."""
    from oracle import port
    return port.time_workload(wl, port.usable_cores(), seconds_budget)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: N fresh child processes, one per GPU, each re-running this file with
    the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  The parent touches no GPU and imports no torch;
    it relays rank 0's JSON line and returns the worst exit status.  If a rank dies the others are ended (by PID)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    import threading
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.read().splitlines()), daemon=True)
    reader.start()
    rc = 0
    try:
        live = list(procs)
        while live:                                   # a rank that dies must not leave the others waiting at a barrier
            time.sleep(0.2)
            for p in list(live):
                if p.poll() is not None:
                    live.remove(p)
                    if p.returncode != 0 and rc == 0:
                        rc = p.returncode
                        print(f"[bench] rank {procs.index(p)} exited with status {p.returncode}; ending the other ranks", file=sys.stderr)
                        for q in live:
                            q.kill()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    reader.join(timeout=10)
    line = None
    for ln in lines:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is None:
        return rc or 1
    print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--no-p2p", action="store_true", help="N > 1: keep the two-kernel + ncclAllReduce step")
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--barcodes", type=int, default=50_000)
    ap.add_argument("--timepoints", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=200, help="eagerly launched steps timed per kernel for `roofline`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process only spawns and waits (it never imports torch nor touches a GPU)
        raise SystemExit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU fallback)"
    # BB_BENCH_REHEARSAL=1 (one-GPU boxes): all ranks on device 0, gloo instead of RCCL (which refuses two ranks on one
    # device), no in-library communicator -- the N > 1 branches of this file and the resident multi-rank launch still run
    rehearsal = world > 1 and os.environ.get("BB_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
        # all ranks' resident launches must be co-resident on the ONE device: 256 / world tiles of 1024 threads per rank
        # (one 1024-thread workgroup fills a CU's register file; the engine sizes tiles for a whole GPU per rank otherwise)
        per_rank = -(-args.barcodes // world)
        os.environ.setdefault("BB_TUNE_NB", str(-(-per_rank // max(256 // world, 1))))
        os.environ.setdefault("BB_TUNE_NTHR", "1024")
    torch.cuda.set_device(local_rank)
    tdev = "cpu" if rehearsal else "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    import barbay_jl_amd as bb
    from barbay_jl_amd import synth

    wl = synth.fitness_normal(args.barcodes, args.timepoints, seed=42)
    eng = bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, seed=42, device=local_rank, rank=rank, world_size=world)
    if world > 1:
        if not rehearsal:
            ids = [eng.make_comm_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            eng.comm_init(ids[0])
        # the resident multi-GPU launch (inboxes mapped over xGMI), only if every rank can; else the RCCL step stays
        exchange = "p2p" if (not args.no_p2p and bb.dist.setup_p2p(eng)) else "rccl"
    else:
        exchange = "none"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()              # (the barrier is itself GPU work)

    # Timed region at N > 1: all ranks leave ONE barrier together (fence), every rank then times its own K steps
    # up to its own device being idle, and the job's time is the MAX over the ranks -- the moment the last rank has finished, which is
    # what a closing barrier would see LESS the barrier's own cost (gloo: ~100 us of TCP, RCCL: a kernel launch and a ring; a 20-step
    # region is ~300 us).  The MAX is formed after the clocks have stopped; every rank's own time is printed (`rank_ms`).

    if exchange == "p2p":
        # first resident launches under a vote: a rank that times out must not leave the others behind on another path
        ok = True
        try:
            eng.run(args.warmup)
        except bb.BarBayHipError as err:
            ok = False
            print(f"[rank {rank}] resident multi-GPU launch failed, falling back to the RCCL step: {err}", file=sys.stderr, flush=True)
        votes = [None] * world
        dist.all_gather_object(votes, ok)
        if not all(votes):
            eng.p2p_enable(False)
            eng.init_meanfield()                  # ranks may have stopped at different steps: start over, identically
            exchange = "rccl (p2p fell back)"
            eng.run(args.warmup)
    else:
        eng.run(args.warmup)
    fence()
    t0 = time.perf_counter()
    try:
        eng.run(args.steps)      # returns after the engine's stream has drained
    except bb._capi.BarBayNonFinite as err:       # the steps were taken; reported below as posterior_finite = false
        print(f"[rank {rank}] {err}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()                      # (bb_run has drained the engine's stream already; this closes the bracket on torch's side)
    dt = time.perf_counter() - t0
    rank_ms = [round(dt * 1e3, 4)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        every = [None] * world
        dist.all_gather_object(every, dt)
        rank_ms = [round(x * 1e3, 4) for x in every]
        dt = float(t.item())
    st = eng.stats()
    mu, sigma = eng.posterior()
    finite = bool(np.isfinite(mu).all() and np.isfinite(sigma).all())
    replicas_equal = None
    if world > 1:
        # every rank updates its copy of the global latents from the totals it assembled: any lost or stale moment row
        # on any rank shows up as a difference between the copies
        lay = {n: (lo, hi) for n, lo, hi in eng.layout()}
        glo, ghi = lay["s_pop"][0], lay["logsigma_pop"][1]
        copies = [None] * world
        dist.all_gather_object(copies, (mu[glo:ghi].tobytes(), sigma[glo:ghi].tobytes()))
        replicas_equal = all(c == copies[0] for c in copies)

    roofline = None
    if st["persistent_pairs"] > 0:      # (N > 1: rank 0's launch and rank 0's shard of the bytes -- a per-GPU figure)
        # resident launch: the timed region IS the kernel (HIP events on the engine's stream bracket its launches)
        launches = max(int(st["launches_last_run"]), 1)
        steps_per_launch = args.steps / launches
        launch_s = st["last_run_ms"] * 1e-3 / launches
        ach = st["bytes_per_step"] * steps_per_launch / launch_s / 1e9
        kname = {1: "k_persist", 2: "k_res", 3: "k_stream"}.get(int(st.get("resident_kernel", 1)), "k_persist")
        roofline = {"bound": "hbm", "kernel": kname, "kernel_instance": eng.kernel_name(),      # (the library names the template instance it launches: bb_kernel_name)
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": int(st["bytes_per_step"] * steps_per_launch),
                    "algorithmic_bytes_per_step": int(st["bytes_per_step"]), "steps_per_launch": steps_per_launch,
                    "avg_launch_us": round(launch_s * 1e6, 2), "workgroups": int(st["n_blocks"]), "threads": int(st["block_threads"]),
                    # `achieved` / `frac` price the ALGORITHMIC bytes of a step (every latent's state read and written
                    # once) as the bench contract asks; the state of a resident launch lives in registers, so what actually crosses the
                    # memory side is `traffic` (counter passes) and its share of the peak is `hbm_side_frac`
                    "frac_basis": "algorithmic bytes per step / launch time (not counter bytes: see traffic, hbm_side_frac)"}
    elif world == 1:
        eng.run_profiled(args.profile_steps)
        sp = eng.stats()
        upd_s = sp["avg_update_ms"] * 1e-3
        ach = sp["bytes_update"] / upd_s / 1e9
        roofline = {"bound": "hbm", "kernel": "k_update", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "algorithmic_bytes_per_launch": int(sp["bytes_update"]),
                    "avg_launch_us": round(sp["avg_update_ms"] * 1e3, 2),
                    "other_kernels": {"k_sample": {"avg_launch_us": round(sp["avg_sample_ms"] * 1e3, 2),
                                                   "algorithmic_bytes_per_launch": int(sp["bytes_sample"])}}}
    if roofline is not None and world > 1:
        roofline["scope"] = "rank 0's GPU and its shard of the bytes"
    if roofline is not None and world == 1:
        # `traffic` is NOT measured in this run (PMC counters need rocprofv3 around the process): it is REPLAYED from the last
        # committed counter passes of the same kernel, and only while the device sources are still the ones profiled
        tr = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
        why = "profiles/hbm_traffic_latest.json missing"
        if os.path.exists(tr):
            try:
                from tools.hbm_traffic import csrc_sha
                t = json.load(open(tr))
                inst = str(t.get("kernel_instance", ""))
                want_inst = roofline.get("kernel_instance", roofline["kernel"])
                if t.get("kernel") != roofline["kernel"] or (roofline.get("kernel_instance") and want_inst.replace(" ", "") not in inst.replace(" ", "")):
                    why = f"counter passes are of {inst or t.get('kernel')}, this run's kernel is {want_inst}"
                elif t.get("csrc_sha") != csrc_sha():
                    why = "device sources changed since the counter passes were taken (csrc hash differs)"
                else:
                    per_step = float(t["hbm_bytes_per_step"])
                    spl = roofline.get("steps_per_launch", 1)
                    roofline["traffic"] = int(per_step * spl)
                    roofline["traffic_source"] = (f"replayed, not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                                  f"'{t.get('source')}' of {inst}, {t.get('steps_in_launch')} steps per launch, scaled per step")
                    roofline["traffic_steps_in_launch"] = t.get("steps_in_launch")
                    roofline["traffic_note"] = t.get("note")
                    # counter bytes over the measured launch time: what actually crossed the memory side, against the peak
                    roofline["hbm_side_frac"] = round(per_step * spl / (roofline["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                    why = None
            except Exception as err:      # a broken profile file must not take the bench line with it
                why = f"unreadable: {err}"
        if why:
            roofline["traffic_source"] = f"none ({why})"

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl)

    if rank == 0:
        out = {
            "metric": "ELBO-grad steps/sec, 50k barcodes x 8 timepoints",
            "value": round(args.steps / dt, 2),
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "rank_ms": rank_ms,                       # every rank's own time over the K steps; value uses their MAX
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": wl.name, "model": "fitness_normal", "barcodes": wl.B, "timepoints": args.timepoints,
                       "n_latents": int(st["n_latents"]), "samples_per_step": 1,
                       # which path ran on rank 0 (0: two kernels per step, 1: k_persist, 2: k_res, 3: k_stream), its instance and grid
                       "resident_kernel": int(st.get("resident_kernel", 0)),
                       "kernel_instance": eng.kernel_name(),
                       "tiles_per_rank": int(st["n_blocks"]), "threads_per_tile": int(st["block_threads"]),
                       "tiles_storing_their_row_through_their_leaders_l2": int(st["rows_same_xcd"]),
                       "optimizer": "TruncatedADAGrad(0.1, 40, 100)", "sharding": f"barcodes/{world}",
                       "collective": "none" if world == 1 else (
                           f"resident launch per rank; 8 group rows of {int(st['n_moments'])} + 2(T-1) f64 pushed into every rank's inbox over xGMI per step"
                           if exchange == "p2p" else f"1 ncclAllReduce of {int(st['n_moments'])} f64 per step ({exchange})")},
            "posterior_finite": finite,
            "replicated_latents_identical_on_all_ranks": replicas_equal,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
