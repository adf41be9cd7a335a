"""GPU, >= 2 physical devices (skipped on the one-GPU boxes): the sharded run over REAL peers, both exchange paths --
(a) two kernels + ncclAllReduce of the K moment rows (RCCL over xGMI), (b) the resident launch with the group rows pushed
into peer-mapped inboxes -- each against the unsharded run, with the replicated global latents bit-identical on all ranks;
and `python bench.py --gpus N` started as a plain process (it spawns its own ranks)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_gpus():
    import torch
    return torch.cuda.device_count()       # (counting devices does not initialise the GPU)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, case, steps, p2p, out_dir, ms=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import barbay_jl_amd as bb
    import _cases as c
    from conftest import make_engine
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    sp = c.synth(case, seed=4)
    ekw = dict(samples_per_step=2, elbo_every=1) if ms else {}
    e = make_engine(sp, None, seed=5, window=4, resum_every=1, rank=rank, world_size=world, device=rank, **ekw)
    bb.dist.init_rccl(e)
    on = bb.dist.setup_p2p(e) if p2p else False
    for _ in range(2):                      # the second pass restarts with the inboxes still holding the first's words
        e.init_meanfield()
        e.run(3)
        e.run(steps - 3)
    mean, sigma = bb.dist.gather_posterior(e, sp.kind, sp.n_neutral, sp.n_bc, sp.n_time, sp.n_rep, sp.n_env)
    lay = {n: (lo, hi) for n, lo, hi in e.layout()}
    glo, ghi = lay["s_pop"][0], lay["logsigma_pop"][1]
    m_own, s_own = e.posterior()
    copies = [None] * world
    dist.all_gather_object(copies, (m_own[glo:ghi].tobytes(), s_own[glo:ghi].tobytes()))
    st = e.stats()
    if rank == 0:
        np.savez(os.path.join(out_dir, "sharded.npz"), mean=mean, sigma=sigma, on=on, pairs=st["persistent_pairs"],
                 identical=all(cp == copies[0] for cp in copies), name=e.kernel_name(),
                 trace=e.elbo_trace(0, steps) if ms else np.zeros(0))
    dist.barrier()
    e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("p2p", [False, True], ids=["rccl", "xgmi_inbox"])
@pytest.mark.parametrize("case", ["fitness_multi_tile", "replicate_ragged", "genotype_runs"])
def test_sharded_over_real_peers(hip_lib, tmp_path, monkeypatch, case, p2p):
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    import torch.multiprocessing as mp
    import _cases as c
    from conftest import make_engine
    geno = case == "genotype_runs"
    world = min(_n_gpus(), 2 if geno else 4)
    monkeypatch.setenv("BB_TUNE_NB", "24" if geno else "8")   # >= 8 tiles per rank (the resident launch's minimum); a tile holds whole genotypes
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    steps = 50
    sp = c.synth(case, seed=4)
    with make_engine(sp, hip_lib, seed=5, window=4, resum_every=1, launch_mode=1) as e1:
        e1.run(steps)
        m1, s1 = e1.posterior()
    mp.spawn(_worker, args=(world, _free_port(), case, steps, p2p, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sharded.npz")
    if p2p:
        assert bool(got["on"]) and int(got["pairs"]) >= 1          # the resident launch over peer-mapped inboxes really ran
    assert bool(got["identical"])                                   # replicated global latents: bit-identical on all ranks
    assert np.abs(got["mean"] - m1).max() < 1e-8 and np.abs(got["sigma"] - s1).max() < 1e-8


@pytest.mark.parametrize("case", ["fitness_T6", "genotype"])
def test_sharded_several_samples_and_elbo_trace_over_real_peers(hip_lib, tmp_path, monkeypatch, case):
    """round 4: `Turing.ADVI(2, ...)` with the ELBO recorded every step stays in the resident launch on a sharded run (MS + cross-GPU
    instances); `genotype`: geno_idx in order of appearance -- regrouped inside the library, ownership by bb_get_owned."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    import torch.multiprocessing as mp
    import _cases as c
    from conftest import make_engine
    monkeypatch.setenv("BB_TUNE_NB", "24" if case == "genotype" else "8")
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    steps = 20
    sp = c.synth(case, seed=4)
    with make_engine(sp, hip_lib, seed=5, window=4, resum_every=1, launch_mode=1, samples_per_step=2, elbo_every=1) as e1:
        e1.run(steps)
        m1, s1 = e1.posterior()
        t1 = e1.elbo_trace(0, steps)
    mp.spawn(_worker, args=(2, _free_port(), case, steps, True, str(tmp_path), True), nprocs=2, join=True)
    got = np.load(tmp_path / "sharded.npz")
    assert bool(got["on"]) and int(got["pairs"]) >= 1 and str(got["name"]).endswith(",true>"), str(got["name"])
    assert bool(got["identical"])
    assert np.abs(got["mean"] - m1).max() < 1e-8 and np.abs(got["sigma"] - s1).max() < 1e-8
    assert np.abs(got["trace"] - t1).max() <= 1e-9 * np.abs(t1).max()


@pytest.mark.parametrize("flags", [[], ["--no-p2p"]], ids=["resident", "rccl"])
def test_bench_spawns_its_own_ranks(flags):
    """`python bench.py --gpus N` as the driver starts it at N = 1: no launcher, no WORLD_SIZE."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10",
                        "--no-cpu-baseline"] + flags, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["posterior_finite"] and out["replicated_latents_identical_on_all_ranks"]
    assert out["value"] > 0


def _bench_rehearsal(n, steps, warmup):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["BB_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup),
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    return json.loads(line[0])


def test_bench_rehearsal_two_ranks_on_one_gpu():
    """One-GPU boxes: `BB_BENCH_REHEARSAL=1 python bench.py --gpus 2` (plain process, self-spawned ranks, both on device 0,
    gloo for the votes, the resident launch's inboxes through hipIpc) must print its one JSON line -- and the driver's short form
    (20 steps after 5) must measure the KERNELS, not the bracket: VERDICT r03 item 2.  With the closing barrier inside the timed region
    the 20-step form read 49.0 k steps/s against 76.6 k over 4000 steps (1.56x per step); now the region ends with each rank's own
    device idle and the job's time is the MAX over ranks, so what separates the two forms is one launch's fixed cost (prologue,
    epilogue, host: ~45 us over 20 steps; two processes on one GPU add to it)."""
    short = _bench_rehearsal(2, 20, 5)
    assert short["n_gpus"] == 2 and short["posterior_finite"] and short["replicated_latents_identical_on_all_ranks"]
    assert "resident launch" in short["config"]["collective"]
    assert len(short["rank_ms"]) == 2 and max(short["rank_ms"]) == pytest.approx(short["ms_per_step"] * 20, rel=1e-3)
    assert short["config"]["kernel_instance"].startswith("k_res<0,1,1024,true,")
    # the timed region is the launch plus the host's share of bb_run, not the bracket: with the closing gloo barrier inside it the region
    # was ~100 us longer than the launch's own HIP-event time; now 25 us (N = 1: 26 us)
    wall_us, kernel_us = short["ms_per_step"] * 20 * 1e3, short["roofline"]["avg_launch_us"]
    assert wall_us - kernel_us < 60.0, (wall_us, kernel_us)
    steady = _bench_rehearsal(2, 4000, 200)
    ratio = short["ms_per_step"] / steady["ms_per_step"]
    print(f"20-step form {short['value']:.0f} steps/s, 4000-step form {steady['value']:.0f} steps/s, per-step ratio {ratio:.3f}")
    assert ratio < 1.45, (short["ms_per_step"], steady["ms_per_step"])          # (measured 1.25 - 1.30; 1.56 with the barrier inside)


@pytest.mark.parametrize("case", ["fitness_T6", "genotype_runs"])
def test_one_handle_drives_all_devices(hip_lib, monkeypatch, case):
    """bb_advi_opts.n_devices over REAL peers: one process, one handle, hipDeviceEnablePeerAccess inboxes (genotype model: every
    device owns its genotypes' theta, gathered from the owners at the end of each run)."""
    if _n_gpus() < 2:
        pytest.skip("needs >= 2 GPUs")
    import _cases as c
    n = min(_n_gpus(), 4 if case == "fitness_T6" else 2)
    monkeypatch.setenv("BB_TUNE_NB", "8" if case == "fitness_T6" else "24")     # (a tile must hold whole genotypes; >= 8 tiles per device)
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    sp = c.synth(case, seed=4)
    from conftest import make_engine
    kw = dict(seed=5, window=4, resum_every=1)
    with make_engine(sp, hip_lib, launch_mode=1, **kw) as e1:
        e1.run(50)
        m1, s1 = e1.posterior()
    with make_engine(sp, hip_lib, device_ids=list(range(n)), **kw) as e:
        e.run(20)
        e.run(30)
        st = e.stats()
        m, s = e.posterior()
    assert st["resident_kernel"] > 0, st
    assert np.abs(m - m1).max() < 1e-8 and np.abs(s - s1).max() < 1e-8
