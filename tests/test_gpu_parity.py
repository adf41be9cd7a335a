"""GPU parity tests: the HIP engine through the C ABI against the oracle, same cases as the
emulation tests plus full-size properties.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest

import _cases as c

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["data001_single", "data002_hier-rep", "data003_multienv", "data004_multigen"])
def test_golden(hip_lib, name):
    c.case_golden(hip_lib, name)


@pytest.mark.parametrize("name", list(c.SYNTH))
def test_synth_grad(hip_lib, name):
    c.case_synth_grad(hip_lib, name)


def test_normals(hip_lib):
    c.case_normals(hip_lib)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "genotype", "replicate_ragged", "multienv_replicate"])
@pytest.mark.parametrize("opt", ["TruncatedADAGrad", "DecayedADAGrad"])
@pytest.mark.parametrize("S", [1, 2])
def test_trajectory_exact(hip_lib, name, opt, S):
    c.case_trajectory_exact(hip_lib, name, opt, S)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "replicate_3d"])
@pytest.mark.parametrize("graph", [-1, 4])
def test_trajectory_running(hip_lib, name, graph):
    c.case_trajectory_running(hip_lib, name, graph)


def test_matrix_priors(hip_lib):
    c.case_matrix_priors(hip_lib)


@pytest.mark.parametrize("name", ["fitness_T6", "multienv_T8", "genotype_runs", "replicate_R3", "multienv_replicate_T6", "fitness_multi_tile"])
@pytest.mark.parametrize("mode", [1, 2])
def test_several_samples_and_elbo_trace_resident(hip_lib, name, mode):
    """Turing.ADVI(samples_per_step, ..) with S = 2, 3 and the ELBO trace: launch_mode 2 = k_res's MS instances (every sample its own
    exchange inside the one launch, gradients summed in registers, the ELBO from the reduced moments) against the literal oracle's
    loop -- and launch_mode 1, the two-kernel step, against the same."""
    assert c.case_trajectory_exact(hip_lib, name, "TruncatedADAGrad", 2, launch_mode=mode) == (2 if mode == 2 else 0)
    assert c.case_trajectory_exact(hip_lib, name, "DecayedADAGrad", 3, launch_mode=mode) == (2 if mode == 2 else 0)
    assert c.case_trajectory_running(hip_lib, name, S=2, launch_mode=mode) == (2 if mode == 2 else 0)
    assert c.case_trajectory_running(hip_lib, name, S=1, launch_mode=mode) == (2 if mode == 2 else 0)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "replicate_ragged", "multienv_replicate"])
def test_sharded_split_phase(hip_lib, name):
    c.case_sharded_split_phase(hip_lib, name)


def test_errors(hip_lib):
    c.case_errors(hip_lib)


def test_graph_equals_eager(hip_lib):
    """hipGraph replay and eager launches run the same arithmetic."""
    from conftest import make_engine
    sp = c.synth("fitness_multi_tile", seed=8)
    outs = []
    for g in (-1, 6):
        with make_engine(sp, hip_lib, seed=3, steps_per_graph=g) as e:
            e.run(31)
            outs.append(e.get_params())
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("nb,nthr", [(7, 64), (100, 1024), (33, 128), (64, 512)])
def test_launch_geometries(hip_lib, monkeypatch, nb, nthr):
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    c.case_synth_grad(hip_lib, "fitness_multi_tile")
    c.case_synth_grad(hip_lib, "replicate_ragged")
    c.case_trajectory_exact(hip_lib, "multienv", "TruncatedADAGrad", 2)
    c.case_trajectory_exact(hip_lib, "genotype", "DecayedADAGrad", 1)


def test_full_size_properties(hip_lib):
    """BASELINE config C2 (50 000 x 8): size-independent properties at full size --
    (i) the engine's gradient equals the oracle's C port on the same draws, (ii) ELBO is finite and
    rises over 300 steps, (iii) 1-shard and 3-shard split-phase runs agree."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import advi, port, rng
    wl = synth.fitness_normal(50_000, 8, 42)
    sp = port.spec_from_workload(wl)
    p = port.Port(sp)
    with make_engine(sp, hip_lib, seed=42, elbo_every=50) as e:
        mu, om = e.get_params()
        eps = rng.normals(42, 0, 0, sp.D)[None, :]
        el, gm, go = e.elbo_grad(mu, om, eps)
        el2, gm2, go2 = p.elbo_grad(mu, om, eps, nthreads=8)
        assert abs(el - el2) <= 1e-11 * abs(el2)
        assert np.abs(gm - gm2).max() <= 1e-9 * np.abs(gm2).max()
        assert np.abs(go - go2).max() <= 1e-9 * np.abs(go2).max()
        e.run(301)
        tr = e.elbo_trace(0, 7)
        assert np.all(np.isfinite(tr)) and tr[-1] > tr[0]
        m, s = e.posterior()
        assert np.isfinite(m).all() and (s > 0).all()


def test_rccl_path_single_rank(hip_lib, monkeypatch):
    """The in-library collective path (k_reduce -> ncclAllReduce on the engine's stream -> k_update) with a
    one-rank communicator equals the collective-free path (same moments, different summation grouping)."""
    from conftest import make_engine
    sp = c.synth("fitness_multi_tile", seed=8)
    with make_engine(sp, hip_lib, seed=3) as e:
        e.run(25)
        ref = e.get_params()
    monkeypatch.setenv("BB_FORCE_ALLREDUCE", "1")
    with make_engine(sp, hip_lib, seed=3) as e:
        e.comm_init(e.make_comm_id())
        e.run(25)
        got = e.get_params()
    assert np.abs(got[0] - ref[0]).max() < 1e-9 and np.abs(got[1] - ref[1]).max() < 1e-9


@pytest.mark.parametrize("name", ["fitness_multi_tile", "fitness_T2", "multienv", "replicate_ragged", "replicate_3d", "multienv_replicate",
                                  "multienv_replicate_3d"])
def test_persistent_equals_two_kernel(hip_lib, monkeypatch, name):
    """Odd numbers of time points / an odd loglambda offset.  By default these shapes run k_persist (measured faster there); k_res
    has any-parity instances for them -- a barcode's last lane owns a single latent, pairs take their normals from two Philox
    pairs where their flat index is odd -- used where k_persist cannot run (genotype model, oversize tiles) and forced here."""
    c.case_persistent_equals_two_kernel(hip_lib, name)
    monkeypatch.setenv("BB_TUNE_AP", "1")
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "replicate_ragged", "multienv_replicate_3d", "fitness_T6"])
def test_first_generation_resident_launch(hip_lib, monkeypatch, name):
    """k_persist stays the fallback where k_res does not apply (ragged-method pairing, moment rows longer than a tile has
    threads, more than 16 time points); BB_NO_RES=1 selects it everywhere."""
    monkeypatch.setenv("BB_NO_RES", "1")
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=1)


def test_persistent_two_pairs_per_thread(hip_lib, monkeypatch):
    monkeypatch.setenv("BB_TUNE_NB", "60")            # 60 barcodes x 13 pairs on 512 threads: P = 2
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    c.case_persistent_equals_two_kernel(hip_lib, "replicate_ragged")


def test_persistent_full_size_matches_two_kernel(hip_lib):
    """C2 at full size: 256 resident workgroups, 120 steps incl. a window re-add, against launch_mode 1."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port
    sp = port.spec_from_workload(synth.fitness_normal(50_000, 8, 42))
    outs = []
    for mode in (1, 2):
        with make_engine(sp, hip_lib, seed=42, launch_mode=mode) as e:
            e.run(120)
            outs.append(e.get_params())
    assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-9 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-9


@pytest.mark.parametrize("cfg", ["C3_replicate", "C4_multienv", "C5_genotype", "multienv_replicate"])
def test_full_size_other_configs(hip_lib, cfg):
    """BASELINE configs 3-5 at full size: the engine's ELBO gradient on a fixed draw equals the oracle's C port,
    60 optimiser steps stay finite and raise the ELBO, and (where eligible) the resident launch equals the
    two-kernel path."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port, rng
    wl = {"C3_replicate": lambda: synth.replicate_fitness_normal(20_000, 6, 3, 43),
          "C4_multienv": lambda: synth.multienv_fitness_normal(20_000, 6, (1, 1, 2, 3, 4, 1), 44),
          "C5_genotype": lambda: synth.genotype_fitness_normal(200_000, 8, 5_000, 45),
          "multienv_replicate": lambda: synth.multienv_replicate_fitness_normal()}[cfg]()
    sp = port.spec_from_workload(wl)
    p = port.Port(sp)
    with make_engine(sp, hip_lib, seed=7, elbo_every=20, launch_mode=1) as e:
        mu, om = e.get_params()
        eps = rng.normals(7, 0, 0, sp.D)[None, :]
        el, gm, go = e.elbo_grad(mu, om, eps)
        el2, gm2, go2 = p.elbo_grad(mu, om, eps, nthreads=8)
        assert abs(el - el2) <= 1e-10 * abs(el2)
        assert np.abs(gm - gm2).max() <= 1e-9 * np.abs(gm2).max()
        assert np.abs(go - go2).max() <= 1e-9 * np.abs(go2).max()
        e.run(61)
        tr = e.elbo_trace(0, 4)
        assert np.all(np.isfinite(tr)) and tr[-1] > tr[0]
        ref = e.get_params()
    if cfg != "C5_genotype":
        with make_engine(sp, hip_lib, seed=7, launch_mode=2) as e:
            e.run(61)
            got = e.get_params()
        # elbo_every only adds the ELBO reduction, the parameter arithmetic is the same
        assert np.abs(got[0] - ref[0]).max() < 1e-8 and np.abs(got[1] - ref[1]).max() < 1e-8


@pytest.mark.parametrize("mode", [1, 2])
def test_ragged_method_pairing(hip_lib, mode):
    c.case_ragged_method(hip_lib, mode)


@pytest.mark.parametrize("name", ["genotype", "replicate_ragged", "multienv_replicate_3d"])
def test_hier_fitness(hip_lib, name):
    c.case_hier_fitness(hip_lib, name)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "genotype", "replicate_ragged", "multienv_replicate"])
def test_logdensity_grad(hip_lib, name):
    c.case_logdensity(hip_lib, name)


@pytest.mark.parametrize("mode", [1, 2])
def test_runs_are_bit_reproducible(hip_lib, mode):
    """Every cross-thread and cross-workgroup sum has a fixed order (no floating-point atomics anywhere): two runs of the same
    handle settings give bit-identical parameters -- at full C2 size, where a race in the exchange or in the LDS hand-offs of
    the resident launch (mode 2) / the two-kernel path (mode 1) would show as run-to-run noise."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port
    sp = port.spec_from_workload(synth.fitness_normal(50_000, 8, 42))
    outs = []
    for _ in range(2):
        with make_engine(sp, hip_lib, seed=42, launch_mode=mode) as e:
            e.run(250)                       # past the first window re-adds
            outs.append(e.get_params())
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


@pytest.mark.parametrize("workload", ["C2", "C5_one_gpu"])
def test_same_xcd_row_stores_change_nothing_but_the_way(hip_lib, monkeypatch, workload):
    """round 4: a tile that finds itself on its exchange group leader's XCD stores its row through the L2 they share (plain stores) instead of
    writing it through; the entries validate themselves either way, so the run with the switch off (BB_TUNE_ROW_L2=0) must give the same bits --
    at full size, k_res (C2) and k_stream (C5 on one GPU), over two launches (every launch decides again).  bb_stats.rows_same_xcd says how many
    tiles took the short way: none with the switch off; with it on the number depends on where the dispatcher put the tiles (all of them on
    the boxes this was developed on), so only its range is asserted."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port
    wl = synth.fitness_normal(50_000, 8, 42) if workload == "C2" else synth.genotype_fitness_normal()
    sp = port.spec_from_workload(wl)
    outs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("BB_TUNE_ROW_L2", sw)
        with make_engine(sp, hip_lib, seed=42, launch_mode=2) as e:
            assert e.stats()["resident_kernel"] == (2 if workload == "C2" else 3)
            e.run(120)
            e.run(60)
            st = e.stats()
            print(f"{workload}: BB_TUNE_ROW_L2={sw}: {st['rows_same_xcd']} of {st['n_blocks']} tiles store their row through the shared L2")
            assert (st["rows_same_xcd"] == 0) if sw == "0" else (0 <= st["rows_same_xcd"] <= st["n_blocks"])
            outs.append(e.get_params())
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


@pytest.mark.parametrize("name", ["fitness_T2", "fitness_T4", "fitness_T6", "fitness_neutral_heavy", "multienv_T6", "multienv_T8",
                                  "replicate_T6", "replicate_R3", "multienv_replicate_T6", "multienv_replicate_R3"])
def test_owner_computes_launch_equals_two_kernel(hip_lib, name):
    """Even T: launch_mode 2 is k_res (bb_resident.h, the owner of a latent computes); same arithmetic as the two-kernel
    step and as the literal oracle's optimiser trajectory."""
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name", ["replicate_T6", "replicate_R3", "multienv_replicate_T6", "multienv_replicate_R3"])
@pytest.mark.parametrize("opt", ["TruncatedADAGrad", "DecayedADAGrad"])
def test_owner_computes_launch_hierarchical_trajectory(hip_lib, name, opt):
    """The plain (even T, even offset) k_res instances of the replicate kinds against the literal oracle's optimiser loop."""
    from conftest import make_engine
    sp = c.synth(name, seed=2)
    e, a, b, _ = c._trajectory(hip_lib, sp, 12, 1, opt, window=5, resum_every=1, launch_mode=2)
    k = e.stats()["resident_kernel"]
    e.close()
    assert k == 2 and a < 1e-10 and b < 1e-10, (k, a, b)


@pytest.mark.parametrize("cfg", ["C2_fitness", "C3_replicate", "C4_multienv", "C5_rank", "C5_stream", "multienv_replicate_even"])
def test_baseline_kernel_instance_against_the_oracle_loop(hip_lib, monkeypatch, cfg):
    """The kernel instance a BASELINE workload runs at full size -- C2: k_res<fitness, 1 pair slot, 1024 threads, T = 8> at 198
    barcodes per tile; C3: k_res<replicate, 3, 512, T = 6>; C4: k_res<multienv, 1, 1024, T = 6>; C5 as one of its eight ranks holds it:
    k_res<genotype, 1, 1024, T = 8> with whole-genotype tiles of ~117 barcodes; C5 on one GPU: k_stream<genotype, 1024, T = 8> at 782
    barcodes per tile, 8 tiles -- each on a cut of the problem with the full-size tile geometry (barcodes per tile, threads), against the
    LITERAL oracle's ADVI loop (exact window), not only against the two-kernel step.  The instance is the library's own word
    (bb_kernel_name), not inferred from bb_stats."""
    from conftest import make_engine
    from oracle import fixtures
    steps = 10
    if cfg == "C2_fitness":
        sp = fixtures.synthetic("fitness", B=2000, T=8, n_neutral=40, seed=42)
        nb, nthr, inst = 198, 1024, "k_res<0,1,1024,false,8,false,false>"          # 50 000 barcodes / 256 tiles, 65 % leaders
    elif cfg == "C3_replicate":
        sp = fixtures.synthetic("replicate", B=2000, T=6, n_rep=3, n_neutral=40, seed=43)
        nb, nthr, inst = 79, 512, "k_res<3,3,512,false,6,false,false>"             # 20 000 barcodes / 256 tiles, 65 % leaders
    elif cfg == "C4_multienv":
        sp = fixtures.synthetic("multienv", B=2000, T=6, n_env=4, n_neutral=40, seed=44)
        nb, nthr, inst = 79, 1024, "k_res<1,1,1024,false,6,false,false>"
    elif cfg == "C5_rank":
        # 25 000 barcodes of 625 genotypes over ~216 tiles: ~117 barcodes, three genotypes of ~39 mutants per tile
        sp = fixtures.synthetic("genotype", B=2400, T=8, n_geno=60, n_neutral=48, seed=45, geno_runs=True)
        nb, nthr, inst = 125, 1024, "k_res<2,1,1024,false,8,false,false>"
    elif cfg == "C5_stream":
        # 200 000 barcodes over 256 tiles: 782 barcodes and ~4 300 pairs per tile -- five pair slots per thread, the state streamed
        sp = fixtures.synthetic("genotype", B=6256, T=8, n_geno=160, n_neutral=126, seed=45, geno_runs=True)
        nb, nthr, inst = 256, 1024, "k_stream<2,1024,8>"          # (the two-kernel step's own tile; the resident launch's: BB_TUNE_RES_NB)
        monkeypatch.setenv("BB_TUNE_RES_NB", "800")
        steps = 8
    else:
        sp = fixtures.synthetic("multienv_replicate", B=1500, T=6, n_rep=2, n_env=3, n_neutral=30, seed=46)
        nb, nthr, inst = 60, 512, "k_res<4,"
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    e, a, b, _ = c._trajectory(hip_lib, sp, steps, 1, "TruncatedADAGrad", seed=13, window=4, resum_every=1, launch_mode=2)
    st, name = e.stats(), e.kernel_name()
    e.close()
    assert name.startswith(inst), name
    assert st["block_threads"] == nthr and st["resident_kernel"] == (3 if cfg == "C5_stream" else 2), st
    if cfg == "C5_stream":
        assert st["n_blocks"] >= 8 and st["persistent_pairs"] >= 5, st
    assert a < 1e-10 and b < 1e-10, (a, b)


@pytest.mark.parametrize("nb,nthr", [(100, 256), (24, 128), (150, 512), (40, 128), (9, 64), (96, 1024)])
def test_owner_computes_launch_geometries(hip_lib, monkeypatch, nb, nthr):
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    c.case_persistent_equals_two_kernel(hip_lib, "fitness_T6", expect_kernel=2)
    c.case_persistent_equals_two_kernel(hip_lib, "multienv_T8", expect_kernel=2)


@pytest.mark.parametrize("nb,nthr,ng", [(0, 0, 0), (16, 128, 16), (8, 256, 16)])
def test_owner_computes_launch_exchange_shapes(hip_lib, monkeypatch, nb, nthr, ng):
    """the exchange's first hop: a moment row longer than the two-half consume handles (four replicates, 198 entries: 8 groups,
    strided form) and 16 groups on small grids (members per leader uneven)"""
    if nb:
        monkeypatch.setenv("BB_TUNE_NB", str(nb))
        monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
        monkeypatch.setenv("BB_TUNE_NG", str(ng))
    c.case_persistent_equals_two_kernel(hip_lib, "replicate_R4", expect_kernel=1 if nthr == 128 else None)   # (198 entries > 128 threads: k_persist)
    c.case_persistent_equals_two_kernel(hip_lib, "fitness_T6", expect_kernel=2)
    c.case_persistent_equals_two_kernel(hip_lib, "multienv_T8", expect_kernel=2)


@pytest.mark.parametrize("name", ["genotype_runs", "genotype_T8", "genotype_T5"])
def test_owner_computes_launch_genotype(hip_lib, name):
    """Genotype model under k_res (mutants grouped by genotype, tiles own whole genotypes and their theta): same arithmetic as
    the two-kernel step with its grid-wide per-genotype sums, and as the literal oracle's optimiser trajectory."""
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name,nb,nthr", [("fitness_neutral_heavy", 140, 1024), ("multienv_T8", 150, 512), ("genotype_T8", 500, 1024), ("genotype_T8", 250, 512),
                                          ("fitness_T4", 333, 1024),
                                          # round 4: T = 6 (four lanes per barcode, one idle), several replicates (kinds 3, 4)
                                          ("fitness_T6", 350, 1024), ("multienv_T6", 300, 1024), ("genotype_runs", 400, 1024),
                                          ("replicate_T6", 150, 1024), ("replicate_R3_T6", 100, 512), ("replicate_R3_T6", 200, 1024),
                                          ("multienv_replicate_T6", 75, 512), ("multienv_replicate_R3_T8", 80, 1024)])
def test_streaming_resident_launch(hip_lib, monkeypatch, name, nb, nthr):
    """k_stream (bb_stream.h): tiles with more pair slots than the register file holds -- the state streamed from memory in one pass per
    step, the next sample formed inside the G passes, the units' sums by the loglambda lanes -- against the two-kernel step and the literal
    oracle's loop (small shapes forced into few large tiles).  All five model kinds."""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_STREAM", "1")
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=3)


def test_streaming_resident_launch_full_size_c5(hip_lib):
    """BASELINE config 5 on ONE GPU (genotype 200 000 x 8, 5 000 genotypes): k_stream against the two-kernel step, 61 steps."""
    import os
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port
    sp = port.spec_from_workload(synth.genotype_fitness_normal(200_000, 8, 5_000, 45))
    outs = []
    for mode in (1, 2):
        with make_engine(sp, hip_lib, seed=7, launch_mode=mode) as e:
            e.run(61)
            outs.append(e.get_params())
            if mode == 2:
                assert e.stats()["resident_kernel"] == 3
    assert np.all(np.isfinite(outs[1][0])) and np.abs(outs[0][0] - outs[1][0]).max() < 1e-8 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-8


@pytest.mark.parametrize("name,nb,nthr", [("fitness_neutral_heavy", 140, 1024), ("multienv_T8", 150, 1024), ("genotype_T8", 250, 512), ("genotype_T8", 500, 1024),
                                          ("replicate_R3_T6", 100, 512), ("multienv_replicate_T6", 75, 1024)])
def test_streaming_resident_launch_several_samples_and_elbo_trace(hip_lib, monkeypatch, name, nb, nthr):
    """k_stream<.., true> (round 4): several MC samples per step and the ELBO trace with the state streamed -- against the literal oracle's loop."""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_STREAM", "1")
    sp = c.synth(name, seed=2)
    e, a, b, _ = c._trajectory(hip_lib, sp, 9, 2, "TruncatedADAGrad", window=5, resum_every=1, launch_mode=2)
    nm = e.kernel_name()
    e.close()
    assert nm.startswith("k_stream<") and nm.endswith(",true>") and a < 1e-10 and b < 1e-10, (nm, a, b)
    e, a, b, tr = c._trajectory(hip_lib, sp, 11, 3, "DecayedADAGrad", elbo_every=1, launch_mode=2)
    got = e.elbo_trace(0, 11)
    e.close()
    assert a < 1e-10 and b < 1e-10, (a, b)
    assert np.abs(got - tr).max() <= 1e-10 * np.abs(tr).max()


def test_streaming_resident_launch_full_size_replicates(hip_lib):
    """`replicate_fitness_normal` 80 000 x 6 x 3 (2.2 M latents: beyond the register file) on ONE GPU: k_stream<3,1024,6> -- three loglambda
    segments per tile, T = 6 on four lanes per barcode -- against the two-kernel step, 41 steps."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port
    sp = port.spec_from_workload(synth.replicate_fitness_normal(80_000, 6, 3, 43))
    outs = []
    for mode in (1, 2):
        with make_engine(sp, hip_lib, seed=7, launch_mode=mode) as e:
            e.run(41)
            outs.append(e.get_params())
            if mode == 2:
                assert e.kernel_name() == "k_stream<3,1024,6>", e.kernel_name()
    assert np.all(np.isfinite(outs[1][0])) and np.abs(outs[0][0] - outs[1][0]).max() < 1e-8 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-8


@pytest.mark.parametrize("name", ["genotype_runs", "genotype_T8", "genotype_odd"])
def test_genotype_regrouped_inside_the_library(hip_lib, name):
    """geno_idx in order of appearance (a genotype's mutants scattered, as utils.data_to_arrays delivers them): regrouped by the
    library, resident launch, caller's order at the ABI -- bit-equal to the sorted problem, gradient and trajectory against the
    literal oracle on the scattered problem."""
    c.case_genotype_regrouped(hip_lib, name)


@pytest.mark.parametrize("name", ["genotype_odd", "replicate_odd"])
def test_owner_computes_launch_odd_loglambda_offset(hip_lib, monkeypatch, name):
    """loglambda starting at an odd flat index: a k_res pair takes its normals from two Philox pairs and moves as 8-byte
    accesses (the LDS-DMA window-slot prefetch from an 8-byte-aligned address)."""
    monkeypatch.setenv("BB_TUNE_AP", "1")        # (the genotype model takes the any-parity instances by itself)
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=2)
    monkeypatch.setenv("BB_TUNE_NB", "24" if name == "genotype_odd" else "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    c.case_persistent_equals_two_kernel(hip_lib, name, expect_kernel=2)


@pytest.mark.parametrize("nb,nthr,lead", [(24, 128, 100), (40, 256, 65), (64, 512, 50), (100, 1024, 100)])
def test_owner_computes_launch_genotype_geometries(hip_lib, monkeypatch, nb, nthr, lead):
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_LEAD", str(lead))
    c.case_persistent_equals_two_kernel(hip_lib, "genotype_runs", expect_kernel=2)
    c.case_persistent_equals_two_kernel(hip_lib, "genotype_T8", expect_kernel=2)


@pytest.mark.parametrize("B,G", [(25_000, 625), (50_000, 1_250)])
def test_genotype_resident_at_shard_size(hip_lib, B, G):
    """BASELINE config 5 as one rank of its 8-GPU run sees it (200 000 / 8 barcodes, 5 000 / 8 genotypes; as a standalone
    problem its loglambda block starts at an odd flat index), and about the largest genotype problem whose state fits one
    GPU's registers: the resident launch against the two-kernel step."""
    from conftest import make_engine
    from barbay_jl_amd import synth
    from oracle import port
    sp = port.spec_from_workload(synth.genotype_fitness_normal(B, 8, G, 45))
    outs = []
    for mode in (1, 2):
        with make_engine(sp, hip_lib, seed=7, launch_mode=mode) as e:
            e.run(61)
            outs.append(e.get_params())
            if mode == 2:
                assert e.stats()["resident_kernel"] == 2
    assert np.all(np.isfinite(outs[1][0])) and np.abs(outs[0][0] - outs[1][0]).max() < 1e-8 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-8


@pytest.mark.parametrize("name,n", [("fitness_multi_tile", 2), ("fitness_T6", 3), ("multienv_T8", 2), ("replicate_ragged", 2), ("genotype_runs", 2)])
def test_multi_device_handle(hip_lib, monkeypatch, name, n):
    """bb_advi_opts.n_devices: one handle, one host thread, n shards -- here all on device 0, their resident
    launches co-resident, inboxes wired in process -- against the unsharded run."""
    monkeypatch.setenv("BB_TUNE_NB", "16")          # >= 8 tiles per shard; all shards' small grids fit the one GPU together
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    c.case_multi_device_handle(hip_lib, name, n)


@pytest.mark.parametrize("name", ["fitness_T6", "genotype_runs"])
def test_multi_device_handle_several_samples_and_elbo_trace(hip_lib, monkeypatch, name):
    """one handle, two shards, samples_per_step = 2 and the ELBO every step: resident launches (MS + cross-GPU instances), not the
    host-summed two-kernel fallback of round 3"""
    monkeypatch.setenv("BB_TUNE_NB", "24" if name == "genotype_runs" else "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    c.case_multi_device_handle(hip_lib, name, 2, samples_per_step=2, elbo_every=1)


def test_multi_device_handle_host_summed_fallback(hip_lib, monkeypatch):
    monkeypatch.setenv("BB_TUNE_NB", "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    c.case_multi_device_handle(hip_lib, "multienv", 2, expect_resident=False, launch_mode=1)
    c.case_multi_device_handle(hip_lib, "genotype", 3, expect_resident=False)
