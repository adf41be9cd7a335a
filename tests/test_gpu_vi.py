"""GPU: `vi.advi` end to end, written after the reference's own test/vi_tests.jl (one sample, one
iteration, then the properties it asserts) plus a convergence check the reference does not have."""
import os

import numpy as np
import pandas as pd
import pytest

import barbay_jl_amd as bb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return pd.read_csv(os.path.join(GOLD, name + ".csv"))


def test_single_condition():                                                       # vi_tests.jl:17-57
    r = bb.vi.advi(data=load("data001_single"), model=bb.model.fitness_normal, advi=bb.vi.ADVI(1, 1), verbose=False)
    assert isinstance(r, pd.DataFrame)
    assert {"mean", "std", "vartype", "varname"} <= set(r.columns)
    assert {"pop_mean_fitness", "pop_std", "bc_fitness", "bc_std", "log_poisson"} <= set(r.vartype)


def test_hierarchical_replicates_and_uneven():                                    # vi_tests.jl:62-112
    data = load("data002_hier-rep")
    r = bb.vi.advi(data=data, model=bb.model.replicate_fitness_normal, rep_col="rep", advi=bb.vi.ADVI(1, 1), verbose=False)
    assert {"rep", "id", "vartype"} <= set(r.columns)
    assert {"bc_hyperfitness", "bc_noncenter", "bc_deviations"} <= set(r.vartype)
    # derived bc_fitness rows (process_hierarchical_samples!, src/utils.jl:1284-1343) come from the device sampler:
    # draw-for-draw equal to the oracle's restatement on the posterior the frame itself reports
    from oracle import rng
    th, tt, lt = (r[r.vartype == v] for v in ("bc_hyperfitness", "bc_noncenter", "bc_deviations"))
    fit = r[r.vartype == "bc_fitness"]
    assert len(fit) == len(tt) and list(fit["id"]) == list(lt["id"])
    idx = np.arange(len(tt)) % len(th)
    med, sd = rng.hier_fitness(0, 10_000, idx[:25], th["mean"].to_numpy(), th["std"].to_numpy(), lt["mean"].to_numpy(),
                               lt["std"].to_numpy(), tt["mean"].to_numpy(), tt["std"].to_numpy())
    assert np.abs(fit["mean"].to_numpy()[:25] - med).max() < 1e-10 and np.abs(fit["std"].to_numpy()[:25] - sd).max() < 1e-10
    uneven = data[(data.rep != data.rep.max()) | (data.time != data.time.max())]
    r2 = bb.vi.advi(data=uneven, model=bb.model.replicate_fitness_normal, rep_col="rep", advi=bb.vi.ADVI(1, 1), verbose=False)
    assert isinstance(r2, pd.DataFrame)


def test_multienv():                                                               # vi_tests.jl:117-146
    r = bb.vi.advi(data=load("data003_multienv"), model=bb.model.multienv_fitness_normal, env_col="env",
                   advi=bb.vi.ADVI(1, 1), verbose=False)
    assert "env" in r.columns


def test_genotypes():                                                              # vi_tests.jl:151-185
    r = bb.vi.advi(data=load("data004_multigen"), model=bb.model.genotype_fitness_normal, genotype_col="genotype",
                   advi=bb.vi.ADVI(1, 1), verbose=False)
    assert {"bc_hyperfitness", "bc_noncenter", "bc_deviations"} <= set(r.vartype)


def test_decayed_adagrad_and_more_samples():
    r = bb.vi.advi(data=load("data001_single"), model=bb.model.fitness_normal, advi=bb.vi.ADVI(3, 5),
                   opt=bb.vi.DecayedADAGrad(), verbose=False)
    assert np.isfinite(r["mean"]).all() and (r["std"] > 0).all()


def test_output_file(tmp_path):                                                    # vi_tests.jl:212-235
    out = str(tmp_path / "res")
    r = bb.vi.advi(data=load("data001_single"), model=bb.model.fitness_normal, outputname=out, advi=bb.vi.ADVI(1, 1),
                   verbose=False)
    assert r is None and os.path.isfile(out + ".csv")
    assert isinstance(pd.read_csv(out + ".csv"), pd.DataFrame)
    with pytest.raises(bb.BarBayError, match="already processed"):                 # src/vi.jl:106-108
        bb.vi.advi(data=load("data001_single"), model=bb.model.fitness_normal, outputname=out, advi=bb.vi.ADVI(1, 1))


def test_long_run_matches_cpu_port_and_brackets_truth():
    """Not in the reference's tests: 4 000 steps on data001 (matrix-form loglambda prior, the documented usage
    docs/src/examples.md:122-140) give the same posterior as the oracle's C port run with the same Philox
    stream, and the fixture's ground-truth `fitness` column lies within 3 posterior std for every mutant."""
    from oracle import advi as oadvi, fixtures, port
    data = load("data001_single")
    arr = bb.utils.data_to_arrays(data)
    lam_prior = np.column_stack([np.log(arr.bc_count.T.reshape(-1) + 1.0), np.full(75, 3.0)])
    r = bb.vi.advi(data=data, model=bb.model.fitness_normal, model_kwargs={"logλ_prior": lam_prior},
                   advi=bb.vi.ADVI(1, 4000), seed=1, verbose=False)
    sp = fixtures.load("data001_single", loglambda_prior=(lam_prior[:, 0], lam_prior[:, 1]))
    mu0, om0 = oadvi.meanfield_init(1, sp.D)
    mu, om, _, _ = port.Port(sp).run(mu0, om0, 4000, seed=1)
    assert np.abs(r["mean"].to_numpy() - mu).max() < 1e-6
    assert np.abs(r["std"].to_numpy() - oadvi.softplus(om)).max() < 1e-6
    fit = r[r.vartype == "bc_fitness"].set_index("id")
    truth = data[~data.neutral].drop_duplicates("barcode").set_index("barcode")["fitness"].loc[fit.index]
    assert (np.abs(fit["mean"] - truth) < 3 * fit["std"]).all()


@pytest.mark.parametrize("ragged", [False, True])
def test_multienv_replicate_advi(ragged):
    """§8f rank 1 end to end: BarBay.model.multienv_replicate_fitness_normal needs rep_col and env_col."""
    from test_host_surface import _tidy_rep_env
    df = _tidy_rep_env(ragged)
    r = bb.vi.advi(data=df, model=bb.model.multienv_replicate_fitness_normal, rep_col="rep", env_col="env",
                   advi=bb.vi.ADVI(1, 30), verbose=False)
    assert {"bc_hyperfitness", "bc_noncenter", "bc_deviations", "bc_fitness"} <= set(r.vartype)
    assert np.isfinite(r["mean"]).all() and (r["std"] > 0).all()
    with pytest.raises(bb.BarBayError, match="env_col"):
        bb.vi.advi(data=df, model=bb.model.multienv_replicate_fitness_normal, rep_col="rep", advi=bb.vi.ADVI(1, 1))
