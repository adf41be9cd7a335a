"""CPU tests of the HIP block programs through their sequential host emulation (g++ -DBB_EMU build of
barbay.jl_amd/csrc/bb_engine.hip).  Same cases as the GPU parity tests; covers the host logic of the
engine (layout, validation, stepping, sharding) where no GPU exists."""
import dataclasses

import numpy as np
import pytest

import _cases as c


@pytest.mark.parametrize("name", ["data001_single", "data002_hier-rep", "data003_multienv", "data004_multigen"])
def test_golden(emu_lib, name):
    c.case_golden(emu_lib, name)


@pytest.mark.parametrize("name", list(c.SYNTH))
def test_synth_grad(emu_lib, name):
    c.case_synth_grad(emu_lib, name)


def test_normals(emu_lib):
    c.case_normals(emu_lib)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "genotype", "replicate_ragged", "multienv_replicate"])
@pytest.mark.parametrize("opt", ["TruncatedADAGrad", "DecayedADAGrad"])
@pytest.mark.parametrize("S", [1, 2])
def test_trajectory_exact(emu_lib, name, opt, S):
    c.case_trajectory_exact(emu_lib, name, opt, S)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "replicate_3d"])
def test_trajectory_running(emu_lib, name):
    c.case_trajectory_running(emu_lib, name)


def test_matrix_priors(emu_lib):
    c.case_matrix_priors(emu_lib)


@pytest.mark.parametrize("name", ["fitness_T6", "multienv_T8", "genotype_runs", "replicate_R3", "multienv_replicate_T6", "fitness_multi_tile"])
@pytest.mark.parametrize("mode", [1, 2])
def test_several_samples_and_elbo_trace_resident(emu_lib, name, mode):
    """Turing.ADVI(samples_per_step, ..) with S = 2, 3 and the ELBO trace: launch_mode 2 = k_res's MS instances (every sample its own
    exchange inside the one launch, gradients summed in registers, the ELBO from the reduced moments) against the literal oracle's
    loop -- and launch_mode 1, the two-kernel step, against the same."""
    assert c.case_trajectory_exact(emu_lib, name, "TruncatedADAGrad", 2, launch_mode=mode) == (2 if mode == 2 else 0)
    assert c.case_trajectory_exact(emu_lib, name, "DecayedADAGrad", 3, launch_mode=mode) == (2 if mode == 2 else 0)
    assert c.case_trajectory_running(emu_lib, name, S=2, launch_mode=mode) == (2 if mode == 2 else 0)
    assert c.case_trajectory_running(emu_lib, name, S=1, launch_mode=mode) == (2 if mode == 2 else 0)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "replicate_ragged", "multienv_replicate"])
def test_sharded_split_phase(emu_lib, name):
    c.case_sharded_split_phase(emu_lib, name)


def test_errors(emu_lib):
    c.case_errors(emu_lib)


@pytest.mark.parametrize("nb,nthr", [(7, 64), (100, 1024), (33, 128), (64, 512)])
def test_launch_geometries(emu_lib, monkeypatch, nb, nthr):
    """Tile size / workgroup size are launch parameters: results must not depend on them."""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    c.case_synth_grad(emu_lib, "fitness_multi_tile")
    c.case_synth_grad(emu_lib, "replicate_ragged")
    c.case_trajectory_exact(emu_lib, "multienv", "TruncatedADAGrad", 2)
    c.case_trajectory_exact(emu_lib, "genotype", "DecayedADAGrad", 1)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "fitness_T2", "multienv", "replicate_ragged", "replicate_3d", "multienv_replicate",
                                  "multienv_replicate_3d"])
def test_persistent_equals_two_kernel(emu_lib, monkeypatch, name):
    """Odd numbers of time points / an odd loglambda offset.  By default these shapes run k_persist (measured faster there); k_res
    has any-parity instances for them -- a barcode's last lane owns a single latent, pairs take their normals from two Philox
    pairs where their flat index is odd -- used where k_persist cannot run (genotype model, oversize tiles) and forced here."""
    c.case_persistent_equals_two_kernel(emu_lib, name)
    monkeypatch.setenv("BB_TUNE_AP", "1")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name", ["fitness_T2", "fitness_T4", "fitness_T6", "fitness_neutral_heavy", "multienv_T6", "multienv_T8",
                                  "replicate_T6", "replicate_R3", "multienv_replicate_T6", "multienv_replicate_R3"])
def test_owner_computes_launch_equals_two_kernel(emu_lib, name):
    """Even T: launch_mode 2 is k_res (bb_resident.h); same arithmetic as the two-kernel step and the oracle."""
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name", ["fitness_T6", "genotype_runs"])
def test_same_xcd_row_stores_switch(emu_lib, monkeypatch, name):
    """round 4: a tile on its exchange group leader's XCD stores its row with plain stores (br_row_publish); BB_TUNE_ROW_L2=0 keeps every row
    store write-through.  The emulation has one 'XCD': every tile decides for plain stores once it has read its leader's entry of the launch
    (bb_stats.rows_same_xcd), none with the switch off, and the runs agree bit for bit."""
    from conftest import make_engine
    monkeypatch.setenv("BB_TUNE_NB", "24" if name == "genotype_runs" else "8")
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    sp = c.synth(name, seed=4)
    outs = []
    for sw in ("1", "0"):
        monkeypatch.setenv("BB_TUNE_ROW_L2", sw)
        with make_engine(sp, emu_lib, seed=5, window=4, launch_mode=2) as e:
            st = e.stats()
            assert st["resident_kernel"] == 2 and st["rows_same_xcd"] == 0
            e.run(3)
            e.run(3)
            st = e.stats()
            assert st["rows_same_xcd"] == (st["n_blocks"] if sw == "1" else 0), st
            outs.append(e.get_params())
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


@pytest.mark.parametrize("name", ["fitness_T6", "multienv_T8", "replicate_R3", "multienv_replicate_T6", "genotype_runs"])
def test_host_built_tables_equal_the_kernels_own(emu_lib, monkeypatch, name):
    """The tiles' segment tables and the LDS descriptor tables come from the host (bb_engine.hip, host_tables); BB_NO_HOST_TABLES=1
    makes every tile build them in its prologue, as before: the runs must agree bit for bit."""
    import numpy as np
    from conftest import make_engine
    sp = c.synth(name, seed=4)
    outs = []
    for no in ("0", "1"):
        monkeypatch.setenv("BB_NO_HOST_TABLES", no)
        with make_engine(sp, emu_lib, seed=5, window=4, launch_mode=2) as e:
            assert e.stats()["resident_kernel"] == 2
            e.run(6)
            outs.append(e.get_params())
    assert (outs[0][0] == outs[1][0]).all() and (outs[0][1] == outs[1][1]).all()


@pytest.mark.parametrize("nb,nthr", [(100, 256), (24, 128), (150, 512), (40, 128), (9, 64)])
def test_owner_computes_launch_geometries(emu_lib, monkeypatch, nb, nthr):
    """several pair slots per thread, tiles that end inside a wave, neutral / mutant boundary inside a tile"""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    c.case_persistent_equals_two_kernel(emu_lib, "fitness_T6", expect_kernel=2)
    c.case_persistent_equals_two_kernel(emu_lib, "multienv_T8", expect_kernel=2)


@pytest.mark.parametrize("nb,nthr", [(24, 128), (24, 256), (40, 512), (64, 1024)])
def test_owner_computes_launch_geometries_hierarchical(emu_lib, monkeypatch, nb, nthr):
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    c.case_persistent_equals_two_kernel(emu_lib, "replicate_R3", expect_kernel=2)
    c.case_persistent_equals_two_kernel(emu_lib, "multienv_replicate_T6", expect_kernel=2)


@pytest.mark.parametrize("lead", [50, 65, 13])
def test_owner_computes_launch_smaller_leader_tiles(emu_lib, monkeypatch, lead):
    """k_res's own tile map: the exchange's group leaders (tiles 0 .. 7) hold fewer barcodes (neutral / mutant boundary inside
    a leader tile, a last tile that is nearly empty); results must not depend on it."""
    monkeypatch.setenv("BB_TUNE_NB", "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    monkeypatch.setenv("BB_TUNE_LEAD", str(lead))
    c.case_persistent_equals_two_kernel(emu_lib, "fitness_T6", expect_kernel=2)
    c.case_persistent_equals_two_kernel(emu_lib, "multienv_T8", expect_kernel=2)
    c.case_p2p_resident(emu_lib, "fitness_T6", 2)


@pytest.mark.parametrize("name,nb,nthr", [("fitness_T6", 16, 128), ("multienv_T8", 8, 256), ("replicate_R3", 8, 256), ("genotype_runs", 24, 128)])
def test_owner_computes_launch_sixteen_groups(emu_lib, monkeypatch, name, nb, nthr):
    """The exchange's first hop with 16 groups (what a full single-GPU grid uses): 16 leaders, every tile's consume split over two
    thread groups (groups 0-7 and 8-15, partial sums through LDS); totals and results as with 8."""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_NG", "16")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)
    monkeypatch.setenv("BB_TUNE_LEAD", "50")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name,nb,nthr", [("fitness_neutral_heavy", 64, 64), ("fitness_neutral_heavy", 140, 128), ("multienv_T8", 40, 64),
                                          ("genotype_T8", 100, 128), ("genotype_T8", 130, 64), ("fitness_T4", 100, 64),
                                          # round 4: T = 6 (four lanes per barcode, one idle), several replicates, the fifth model
                                          ("fitness_T6", 90, 128), ("multienv_T6", 75, 64), ("genotype_runs", 110, 128),
                                          ("replicate_T6", 40, 128), ("replicate_R3_T6", 50, 128), ("replicate_R3_T6", 30, 256),
                                          ("multienv_replicate_T6", 20, 128), ("multienv_replicate_R3_T8", 30, 256)])
def test_streaming_resident_launch(emu_lib, monkeypatch, name, nb, nthr):
    """k_stream (bb_stream.h): tiles with more pair slots than the register file holds -- the state streamed from memory in ONE pass per
    step (the next sample formed inside the G passes, the units' sums by the loglambda lanes), contributions summed per thread and by
    class over the wave -- against the two-kernel step and the literal oracle's loop.  All five model kinds, T = 4, 6, 8."""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_STREAM", "1")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=3)
    monkeypatch.setenv("BB_TUNE_LEAD", "100")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=3)


@pytest.mark.parametrize("lead", [100, 50])
def test_owner_computes_launch_thirtytwo_groups(emu_lib, monkeypatch, lead):
    """self-validating rows with 32 groups (a leader's members fit one batch of eight loads per lane; the consume runs on four
    thread groups): 132 tiles, uneven member counts."""
    monkeypatch.setenv("BB_TUNE_NB", "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "256")
    monkeypatch.setenv("BB_TUNE_NG", "32")
    monkeypatch.setenv("BB_TUNE_LEAD", str(lead))
    c.case_persistent_equals_two_kernel(emu_lib, "fitness_wide_grid", expect_kernel=2)


@pytest.mark.parametrize("nb,nthr,kernel", [(0, 0, None), (16, 128, 1), (30, 256, 2)])
def test_owner_computes_launch_long_moment_row(emu_lib, monkeypatch, nb, nthr, kernel):
    """four replicates: the moment row has 198 entries -- more than the two-half consume handles (the exchange keeps 8 groups),
    and on a one-GPU tile with fewer threads than that k_res is not used at all (k_persist's strided consume is)"""
    if nb:
        monkeypatch.setenv("BB_TUNE_NB", str(nb))
        monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
        monkeypatch.setenv("BB_TUNE_NG", "16")          # (asked for, refused by the engine: 198 > 128)
    c.case_persistent_equals_two_kernel(emu_lib, "replicate_R4", expect_kernel=kernel)


@pytest.mark.parametrize("name", ["genotype_runs", "genotype_T8", "genotype_T5"])
def test_owner_computes_launch_genotype(emu_lib, name):
    """Genotype model under k_res: mutants grouped by genotype, tiles cut at genotype boundaries own their genotypes' theta
    (sample, stage, d/dtheta_g = sum over the genotype's mutants inside the tile, update) -- against the two-kernel step with
    its grid-wide per-genotype sums, and against the literal oracle."""
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)


@pytest.mark.parametrize("name", ["genotype_odd", "replicate_odd"])
def test_owner_computes_launch_odd_loglambda_offset(emu_lib, monkeypatch, name):
    """An odd number of latents in front of the loglambda block: a pair (b, 2k), (b, 2k+1) then straddles two Philox pairs
    (2q - 1, 2q) -- two draws per thread, 8-byte state accesses; same draws, same arithmetic as everywhere else.  Also sharded."""
    sp = c.synth(name, seed=6)
    assert sp.offsets()["loglambda"][0] % 2 == 1
    monkeypatch.setenv("BB_TUNE_AP", "1")        # (the genotype model takes the any-parity instances by itself)
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)
    monkeypatch.setenv("BB_TUNE_NB", "24" if name == "genotype_odd" else "16")      # (>= 8 tiles per rank)
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=2)
    c.case_p2p_resident(emu_lib, name, 2)


@pytest.mark.parametrize("nb,nthr,lead", [(24, 128, 100), (40, 256, 65), (64, 512, 50), (100, 1024, 100), (30, 64, 100)])
def test_owner_computes_launch_genotype_geometries(emu_lib, monkeypatch, nb, nthr, lead):
    """several pair slots per thread, theta pairs split between two tiles, smaller leader tiles, the neutral / mutant boundary
    inside a tile"""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_LEAD", str(lead))
    c.case_persistent_equals_two_kernel(emu_lib, "genotype_runs", expect_kernel=2)
    c.case_persistent_equals_two_kernel(emu_lib, "genotype_T8", expect_kernel=2)


@pytest.mark.parametrize("world", [2, 3])
def test_owner_computes_launch_genotype_sharded(emu_lib, monkeypatch, world):
    """shards cut at genotype boundaries: every rank owns its genotypes' theta, no second exchange; theta comes back from its
    owner at the end of the run"""
    monkeypatch.setenv("BB_TUNE_NB", "24")
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    c.case_p2p_resident(emu_lib, "genotype_runs", world)
    c.case_p2p_resident(emu_lib, "genotype_T8", world)
    c.case_p2p_resident(emu_lib, "genotype_T5", world)
    c.case_multi_device_handle(emu_lib, "genotype_runs", n=world)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_on_scattered_genotypes(emu_lib, monkeypatch, world):
    """geno_idx in the reference's order of appearance: bb_create regroups the mutants on every rank, the shard ranges then describe
    the INTERNAL order while get_params presents the caller's -- the gather helpers map ownership through engine.permutation()
    (without it a rank's owned indices named other ranks' entries)."""
    monkeypatch.setenv("BB_TUNE_NB", "24")
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    sp = c.synth("genotype", seed=4)
    from conftest import make_engine
    with make_engine(sp, emu_lib, rank=0, world_size=world) as e:
        assert not (e.permutation() == np.arange(e.D)).all()
    c.case_p2p_resident(emu_lib, "genotype", world)


def test_genotype_empty_genotypes_and_unsorted(emu_lib, monkeypatch):
    """genotypes without mutants (theta_g feels its prior only) ride with the tile before them; geno_idx that is not in
    consecutive runs keeps the two-kernel step"""
    import barbay_jl_amd as bb
    from conftest import make_engine
    sp = c.synth("genotype_runs", seed=6)
    gi = np.asarray(sp.geno_idx).copy()
    gi = gi + 2 * (gi >= 7) + 2 * (gi >= 20)            # genotypes 7, 8, 22, 23 are empty; the count stays even
    sp = dataclasses.replace(sp, geno_idx=gi)
    outs = []
    for mode in (1, 2):
        with make_engine(sp, emu_lib, seed=13, window=6, resum_every=1, launch_mode=mode) as e:
            e.run(11)
            outs.append(e.get_params())
            if mode == 2:
                assert e.stats()["resident_kernel"] == 2
    assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-11 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-11
    # geno_idx that is not in consecutive runs: the library regroups the mutants itself (BB_NO_REGROUP=1: the two-kernel step stays)
    with make_engine(c.synth("genotype"), emu_lib, launch_mode=2) as e:
        assert e.stats()["resident_kernel"] == 2 and not (e.permutation() == np.arange(e.D)).all()
    monkeypatch.setenv("BB_NO_REGROUP", "1")
    with pytest.raises(bb.BarBayHipError, match="consecutive runs"):
        make_engine(c.synth("genotype"), emu_lib, launch_mode=2)


@pytest.mark.parametrize("name", ["genotype_runs", "genotype_T8", "genotype_odd"])
def test_genotype_regrouped_inside_the_library(emu_lib, name):
    c.case_genotype_regrouped(emu_lib, name)


@pytest.mark.parametrize("mode", [1, 2])
def test_running_window_stays_near_the_exact_window(emu_lib, mode):
    """TruncatedADAGrad's window sum is kept as a compensated running sum (bb_opt_apply: two error-free sums per step, the
    rounding errors in a float beside the accumulator) and never re-added.  Against the reference's arithmetic (the window added
    up every step): the default, and a re-add period that never comes."""
    from conftest import make_engine
    sp = c.synth("fitness_T6", seed=6)
    outs = {}
    for resum in (1, 0, 100000):
        with make_engine(sp, emu_lib, seed=13, window=10, resum_every=resum, launch_mode=mode) as e:
            e.run(300)
            outs[resum] = e.get_params()
    for resum in (0, 100000):
        assert np.abs(outs[resum][0] - outs[1][0]).max() < 1e-9 and np.abs(outs[resum][1] - outs[1][1]).max() < 1e-9


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "replicate_ragged", "multienv_replicate_3d", "fitness_T6"])
def test_first_generation_resident_launch(emu_lib, monkeypatch, name):
    """k_persist (LDS-staged passes, bb_persist.h) stays the fallback where k_res does not apply -- the ragged-method pairing,
    moment rows longer than a tile has threads, more than 16 time points; BB_NO_RES=1 selects it everywhere."""
    monkeypatch.setenv("BB_NO_RES", "1")
    c.case_persistent_equals_two_kernel(emu_lib, name, expect_kernel=1)
    if name in ("fitness_multi_tile", "replicate_ragged"):
        monkeypatch.setenv("BB_TUNE_NB", "16")
        monkeypatch.setenv("BB_TUNE_NTHR", "512")
        c.case_p2p_resident(emu_lib, name, 2)


def test_persistent_two_pairs_per_thread(emu_lib, monkeypatch):
    monkeypatch.setenv("BB_TUNE_NB", "120")     # 120 barcodes x (16 + 1 + 9) latents / 2 > 1024 pairs -> P = 2
    monkeypatch.setenv("BB_TUNE_NTHR", "1024")
    c.case_persistent_equals_two_kernel(emu_lib, "replicate_ragged")


def test_persistent_not_eligible_is_an_error(emu_lib):
    import barbay_jl_amd as bb
    from conftest import make_engine
    # several samples per step run resident on k_res (its MS instances); where the shape has no k_res instance -- the ragged-method
    # pairing -- the resident launch is refused
    with make_engine(c.synth("multienv"), emu_lib, launch_mode=2, samples_per_step=2) as e:
        assert e.stats()["resident_kernel"] == 2
    with pytest.raises(bb.BarBayHipError, match="samples_per_step"):
        make_engine(c.synth("replicate_ragged"), emu_lib, launch_mode=2, samples_per_step=2, ragged_method=True)


@pytest.mark.parametrize("mode", [1, 2])
def test_ragged_method_pairing(emu_lib, mode):
    c.case_ragged_method(emu_lib, mode)


def test_ragged_method_neutrals_across_tiles(emu_lib, monkeypatch):
    monkeypatch.setenv("BB_TUNE_NB", "16")      # 37 neutrals span three tiles: the (t, j) table is summed over tiles
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    c.case_ragged_method(emu_lib, 1)
    c.case_ragged_method(emu_lib, 2)


@pytest.mark.parametrize("name", ["genotype", "replicate_ragged", "multienv_replicate_3d"])
def test_hier_fitness(emu_lib, name):
    c.case_hier_fitness(emu_lib, name)


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "genotype", "replicate_ragged", "multienv_replicate"])
def test_logdensity_grad(emu_lib, name):
    c.case_logdensity(emu_lib, name)


@pytest.mark.parametrize("name,world", [("fitness_multi_tile", 2), ("fitness_multi_tile", 3), ("multienv", 2), ("replicate_ragged", 2),
                                        ("multienv_replicate", 2), ("fitness_T6", 2), ("fitness_T6", 3), ("multienv_T8", 2),
                                        ("replicate_R3", 2), ("multienv_replicate_R3", 2)])
def test_sharded_resident_launch(emu_lib, monkeypatch, name, world):
    monkeypatch.setenv("BB_TUNE_NB", "16")         # >= 8 tiles on every rank, one pair per thread
    monkeypatch.setenv("BB_TUNE_NTHR", "512")
    c.case_p2p_resident(emu_lib, name, world)


def test_sharded_resident_launch_chunked_inbox(emu_lib, monkeypatch):
    """64-thread tiles: the 8 x world inbox rows do not fit the LDS stage at once and are summed chunk by chunk."""
    monkeypatch.setenv("BB_TUNE_NB", "8")
    monkeypatch.setenv("BB_TUNE_NTHR", "64")
    c.case_p2p_resident(emu_lib, "fitness_multi_tile", 3)


@pytest.mark.parametrize("cfg,world", [("C2", 2), ("C2", 4), ("C2", 8), ("C4", 4), ("C5", 8)])
def test_baseline_shards_run_the_owner_computes_launch(emu_lib, cfg, world):
    """Every shard shape BASELINE.json names -- the headline workload over 2 / 4 / 8 ranks, C4 (multienv) over 4, C5 (genotype)
    over 8 -- must select k_res with its cross-GPU exchange (resident_kernel == 2) on every rank, with >= 8 tiles: otherwise
    bench.py --gpus N silently stays on the RCCL step (C2 / 4 used to: 257 pairs on 256 threads) or on round 1's kernel."""
    from barbay_jl_amd import synth
    import barbay_jl_amd as bb
    wl = {"C2": lambda: synth.fitness_normal(50_000, 8, 42),
          "C4": lambda: synth.multienv_fitness_normal(20_000, 6, (1, 1, 2, 3, 4, 1), 44),
          "C5": lambda: synth.genotype_fitness_normal(200_000, 8, 5_000, 45)}[cfg]()
    es = [bb.Engine(wl.kind, wl.counts, wl.n_neutral, wl.n_bc, env_idx=wl.env_idx, geno_idx=wl.geno_idx, seed=42, rank=r, world_size=world,
                    _lib=emu_lib) for r in range(world)]   # (the default 100-slot window: a shard keeps only its own latents' rows)
    try:
        handles = [e.p2p_export() for e in es]
        for e in es:
            e.p2p_import(handles)
        assert all(e.p2p_selftest() for e in es)
        assert all(e.p2p_enable(True) for e in es)
        st = [e.stats() for e in es]
        assert all(s["resident_kernel"] == 2 and s["n_blocks"] >= 8 for s in st), st
        if cfg != "C5":
            assert all(s["persistent_pairs"] == 1 for s in st), st
        # the memory is sharded with the work: a row of the TruncatedADAGrad window holds the shard's own latents (+ the replicated
        # blocks), not all D, and the two-kernel step's scratch arrays are not allocated on a handle that runs the resident launch
        D = st[0]["n_latents"]
        assert all(s["window_row"] <= D / world * 1.06 + 6_000 for s in st), [(s["window_row"], D) for s in st]
        full = 100 * 2 * 8 * D + 12 * 8 * D                 # what every rank used to allocate: the whole window + 12 D-sized arrays
        assert all(s["device_bytes"] < full / world + 6 * 8 * D + 64e6 for s in st), [s["device_bytes"] for s in st]
        if cfg == "C5":
            assert all(s["device_bytes"] <= 0.6e9 for s in st), [s["device_bytes"] for s in st]      # (3.7 GB per rank before)
    finally:
        for e in es:
            e.close()


def test_default_readd_schedule_tracks_the_exact_window(emu_lib):
    """TruncatedADAGrad's running window under the default re-add schedule (once per window for ten windows, then once per
    ten) against the reference's arithmetic (the whole window re-added every step, resum_every = 1) over 2 200 steps: the
    running sum's cancellation error must stay at rounding level (a fixed re-add every 1 000 steps leaves 6e-6)."""
    import numpy as np
    from conftest import make_engine
    sp = c.synth("fitness_multi_tile", seed=4)
    out = []
    for k in (1, 0):
        with make_engine(sp, emu_lib, seed=5, resum_every=k) as e:
            e.run(2200)
            out.append(e.posterior())
    assert np.abs(out[1][0] - out[0][0]).max() < 1e-9
    assert np.abs(out[1][1] / out[0][1] - 1).max() < 1e-9


def test_running_window_survives_a_gradient_spike(emu_lib):
    """The first ADVI steps from the meanfield start carry 1e14-sized gradients: when their squares leave a short window a
    plain running sum acc + d^2 - old keeps their rounding residue while the sum itself falls by ten orders of magnitude (4e-8
    in the parameters after 160 steps with round 1's re-add schedule, 1.3e-5 without re-adds).  The compensated accumulator
    (bb_opt_apply) must stay at the exact rule's trajectory with no re-add at all."""
    import numpy as np
    from conftest import make_engine
    sp = c.synth("fitness_multi_tile", seed=6)
    out = []
    for k in (1, 0, 1000):          # exact; default (never re-added); a re-add period that does not come within this run
        with make_engine(sp, emu_lib, seed=8, window=7, resum_every=k) as e:
            e.run(160)
            out.append(e.get_params())
    for got in out[1:]:
        assert np.abs(got[0] - out[0][0]).max() < 1e-11, np.abs(got[0] - out[0][0]).max()
        assert np.abs(got[1] - out[0][1]).max() < 1e-11, np.abs(got[1] - out[0][1]).max()


@pytest.mark.parametrize("name,n", [("fitness_multi_tile", 2), ("fitness_T6", 3), ("multienv_T8", 2), ("replicate_ragged", 2)])
def test_multi_device_handle(emu_lib, monkeypatch, name, n):
    """One handle, n shards in one process (bb_advi_opts.n_devices): the SURVEY.md 8b boundary the Julia drop-in calls."""
    monkeypatch.setenv("BB_TUNE_NB", "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "256")
    c.case_multi_device_handle(emu_lib, name, n)


def test_multi_device_handle_launch_mode_2(emu_lib, monkeypatch):
    """launch_mode = 2 on a multi-device handle: the shards are created before their inboxes are wired (a shard alone cannot run
    resident then) and the mode is enforced on the group afterwards; where the group cannot run resident it is an error."""
    import barbay_jl_amd as bb
    from conftest import make_engine
    monkeypatch.setenv("BB_TUNE_NB", "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "256")
    c.case_multi_device_handle(emu_lib, "fitness_T6", 2, launch_mode=2)
    # several samples per step on a sharded handle: the MS cross-GPU instances since round 4 (odd T: their any-parity form)
    with make_engine(c.synth("multienv"), emu_lib, device_ids=[0, 0], launch_mode=2, samples_per_step=2) as e:
        assert e.stats()["resident_kernel"] == 2
    with pytest.raises(bb.BarBayHipError, match="launch_mode = 2"):          # (the ragged-method pairing runs k_persist only: one sample per step)
        make_engine(c.synth("replicate_ragged"), emu_lib, device_ids=[0, 0], launch_mode=2, samples_per_step=2, ragged_method=True)


def test_multi_device_handle_falls_back_to_the_host_summed_step(emu_lib, monkeypatch):
    """Where a shard cannot run the resident launch (genotype model; S > 1) the group steps split-phase, moments summed on the host."""
    monkeypatch.setenv("BB_TUNE_NB", "16")
    monkeypatch.setenv("BB_TUNE_NTHR", "256")
    c.case_multi_device_handle(emu_lib, "multienv", 2, expect_resident=False, launch_mode=1)
    c.case_multi_device_handle(emu_lib, "genotype", 3, expect_resident=False)     # (genotypes of ~32 mutants do not fit these 16-barcode tiles)


@pytest.mark.parametrize("name", ["fitness_T6", "multienv_T8", "genotype_runs", "replicate_T6"])
def test_sharded_resident_launch_several_samples_and_elbo_trace(emu_lib, monkeypatch, name):
    """`Turing.ADVI(samples_per_step, ...)` (src/vi.jl:98) and the ELBO trace on a SHARDED run stay in the resident launch (round 4:
    k_res<.., XG = true, .., MS = true>; every MC sample is an exchange of its own, the inbox epochs count exchanges) -- two and three ranks
    stepped in lock step by the emulation, and one handle driving two shards; results and trace equal the unsharded run's."""
    monkeypatch.setenv("BB_TUNE_NB", {"genotype_runs": "24", "multienv_T8": "8", "replicate_T6": "8"}.get(name, "16"))      # (>= 8 tiles per rank of three)
    monkeypatch.setenv("BB_TUNE_NTHR", "128")
    c.case_p2p_resident(emu_lib, name, 2, samples_per_step=2, elbo_every=1)
    c.case_p2p_resident(emu_lib, name, 3, steps=8, samples_per_step=3, elbo_every=2)
    c.case_multi_device_handle(emu_lib, name, n=2, samples_per_step=2, elbo_every=1)


@pytest.mark.parametrize("name,nb,nthr", [("fitness_neutral_heavy", 140, 128), ("multienv_T8", 40, 64), ("genotype_T8", 100, 128), ("genotype_runs", 110, 128),
                                          ("replicate_R3_T6", 50, 128), ("multienv_replicate_T6", 20, 128)])
def test_streaming_resident_launch_several_samples_and_elbo_trace(emu_lib, monkeypatch, name, nb, nthr):
    """k_stream's MS form (round 4): several MC samples per step -- each an exchange of its own, the gradient sums in memory, the last sample
    updates -- and the ELBO trace, against the literal oracle's loop (S = 2 exact window; S = 3 with the trace, running window)."""
    monkeypatch.setenv("BB_TUNE_NB", str(nb))
    monkeypatch.setenv("BB_TUNE_NTHR", str(nthr))
    monkeypatch.setenv("BB_TUNE_STREAM", "1")
    sp = c.synth(name, seed=2)
    e, a, b, _ = c._trajectory(emu_lib, sp, 9, 2, "TruncatedADAGrad", window=5, resum_every=1, launch_mode=2)
    k = e.stats()["resident_kernel"]
    e.close()
    assert k == 3 and a < 1e-10 and b < 1e-10, (k, a, b)
    e, a, b, tr = c._trajectory(emu_lib, sp, 11, 3, "DecayedADAGrad", elbo_every=1, launch_mode=2)
    got = e.elbo_trace(0, 11)
    k = e.stats()["resident_kernel"]
    e.close()
    assert k == 3 and a < 1e-10 and b < 1e-10, (k, a, b)
    assert np.abs(got - tr).max() <= 1e-10 * np.abs(tr).max()
    # two calls of run (the launch's first sample is formed again from the stored parameters), recording every second step
    with c.make_engine(sp, emu_lib, seed=11, samples_per_step=2, elbo_every=2, launch_mode=2, window=5) as e2:
        e2.run(5); e2.run(4)
        p2 = e2.get_params()
        t2 = e2.elbo_trace(0, 5)
    with c.make_engine(sp, emu_lib, seed=11, samples_per_step=2, elbo_every=2, launch_mode=1, window=5) as e1:
        e1.run(9)
        p1 = e1.get_params()
        t1 = e1.elbo_trace(0, 5)
    assert np.abs(p2[0] - p1[0]).max() < 1e-9 and np.abs(p2[1] - p1[1]).max() < 1e-9
    assert np.abs(t2 - t1).max() <= 1e-10 * np.abs(t1).max()
