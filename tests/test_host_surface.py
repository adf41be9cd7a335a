"""Host-side mirror of BarBay.utils (no GPU): data_to_arrays / advi_to_df follow test/utils_tests.jl's
checks on the reference's own CSV fixtures."""
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest

import barbay_jl_amd as bb

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return pd.read_csv(os.path.join(GOLD, name + ".csv"))


def test_single_arrays():
    d = bb.utils.data_to_arrays(load("data001_single"))
    assert d.bc_count.shape == (5, 15) and d.bc_total.shape == (5,)
    assert (d.n_neutral, d.n_bc, d.n_rep, d.n_env, d.n_time) == (5, 10, 1, 1, 5)
    assert d.envs == "env1" and d.genotypes == "N/A"
    np.testing.assert_array_equal(d.bc_total, d.bc_count.sum(axis=1))
    df = load("data001_single")
    np.testing.assert_array_equal(d.bc_total, df.drop_duplicates("time").sort_values("time")["count_sum"].to_numpy())
    assert d.neutral_ids[0] == "neutral001" and all(str(i).startswith("neutral") for i in d.neutral_ids)


def test_replicate_arrays_3d_and_ragged():
    df = load("data002_hier-rep")
    d = bb.utils.data_to_arrays(df, rep_col="rep")
    assert d.bc_count.shape == (5, 15, 2) and d.bc_total.shape == (5, 2) and d.n_rep == 2
    assert d.bc_ids == sorted(d.bc_ids)
    uneven = df[(df.rep != df.rep.max()) | (df.time != df.time.max())]           # test/vi_tests.jl:102-111
    d2 = bb.utils.data_to_arrays(uneven, rep_col="rep")
    assert isinstance(d2.bc_count, list) and [m.shape for m in d2.bc_count] == [(5, 15), (4, 15)]
    assert d2.n_time == [5, 4]


def test_multienv_and_genotype_arrays():
    d = bb.utils.data_to_arrays(load("data003_multienv"), env_col="env")
    assert d.envs == [1, 1, 2, 3, 1, 2, 3] and d.n_env == 3 and len(d.envs) == d.n_time    # test/utils_tests.jl:355
    g = bb.utils.data_to_arrays(load("data004_multigen"), genotype_col="genotype")
    df = load("data004_multigen")
    want = dict(zip(df.barcode, df.genotype))
    assert g.genotypes == [want[b] for b in g.bc_ids] and g.n_geno == 1                    # test/utils_tests.jl:372-376


def test_group_genotypes_orders_mutants_by_genotype():
    """data_to_arrays(group_genotypes=True): a genotype's barcodes become consecutive (what the engine's resident launch and
    genotype-aligned sharding of genotype_fitness_normal need); every barcode keeps its own counts."""
    df = load("data004_multigen").copy()
    ids = sorted(df.loc[df.neutral.astype(str).str.lower() != "true", "barcode"].unique())
    lab = {b: f"g{(7 * i) % 3}" for i, b in enumerate(ids)}                 # three interleaved genotypes
    df["genotype"] = [lab.get(b, "neutral") for b in df.barcode]
    a = bb.utils.data_to_arrays(df, genotype_col="genotype")
    g = bb.utils.data_to_arrays(df, genotype_col="genotype", group_genotypes=True)
    assert sorted(g.bc_ids) == sorted(a.bc_ids) and g.n_geno == a.n_geno == 3
    runs = [x for i, x in enumerate(g.genotypes) if i == 0 or x != g.genotypes[i - 1]]
    assert len(runs) == 3 and runs == list(dict.fromkeys(a.genotypes))      # consecutive runs, genotypes in order of first appearance
    nn = a.n_neutral
    col = {b: a.bc_count[:, nn + i] for i, b in enumerate(a.bc_ids)}
    assert all((g.bc_count[:, nn + i] == col[b]).all() for i, b in enumerate(g.bc_ids))
    assert (g.bc_count[:, :nn] == a.bc_count[:, :nn]).all() and (g.bc_total == a.bc_total).all()


def test_missing_timepoint_is_an_error():
    df = load("data001_single")
    with pytest.raises(bb.BarBayError, match="Not all"):
        bb.utils.data_to_arrays(df.iloc[1:])


def _fake_q(model):
    D = sum(len(c.reshape(-1)) for c in model.counts)
    return D


@pytest.mark.parametrize("name,model,kw,cols", [
    ("data001_single", "fitness_normal", {}, {}),
    ("data002_hier-rep", "replicate_fitness_normal", {}, {"rep_col": "rep"}),
    ("data003_multienv", "multienv_fitness_normal", {}, {"env_col": "env"}),
    ("data004_multigen", "genotype_fitness_normal", {}, {"genotype_col": "genotype"}),
])
def test_advi_to_df_labels(name, model, kw, cols):
    """Label columns for a synthetic q (no engine): shapes, vartypes and ids as src/utils.jl:1409-1462."""
    df = load(name)
    arrays = bb.utils.data_to_arrays(df, **cols)
    mk = {}
    if "multienv" in model:
        mk["envs"] = arrays.envs
    if "genotype" in model:
        mk["genotypes"] = arrays.genotypes
    bm = getattr(bb.model, model)(arrays.bc_count, arrays.bc_total, arrays.n_neutral, arrays.n_bc, **mk)
    from oracle import fixtures
    sp = fixtures.load(name)
    ranges = list(sp.offsets().values())
    D = sp.D
    q = SimpleNamespace(dist=SimpleNamespace(m=np.arange(D, dtype=float), σ=np.ones(D)), transform=SimpleNamespace(ranges_out=ranges))
    names = []
    for sym, (lo, hi) in zip(bm.var_symbols(), ranges):
        names += [f"{sym}[{i}]" for i in range(1, hi - lo + 1)]
    out = bb.utils.advi_to_df(df, q, names, **cols, n_samples=200, rng=np.random.default_rng(0))
    assert {"mean", "std", "varname", "vartype", "id"} <= set(out.columns)
    base = out.iloc[:D]
    assert (base["mean"].to_numpy() == np.arange(D)).all()
    assert (base[base.vartype == "pop_mean_fitness"]["id"] == "N/A").all()
    lam = base[base.vartype == "log_poisson"]
    assert lam["id"].iloc[0] == arrays.neutral_ids[0] and lam["id"].iloc[-1] == arrays.bc_ids[-1]
    if model in ("replicate_fitness_normal", "genotype_fitness_normal"):
        assert {"bc_hyperfitness", "bc_noncenter", "bc_deviations"} <= set(out.vartype)
        n_units = (out.vartype == "bc_deviations").sum()
        assert len(out) == D + n_units and (out.iloc[D:].vartype == "bc_fitness").all()   # derived rows (:1284-1343)
    if "rep_col" in cols:
        assert set(base[base.vartype == "log_poisson"]["rep"]) == {"R1", "R2"}
        assert (base[base.vartype == "bc_hyperfitness"]["rep"] == "N/A").all()
    if "env_col" in cols:
        assert list(base[base.vartype == "pop_mean_fitness"]["env"]) == arrays.envs[1:]


def test_model_constructors_validate():
    d = bb.utils.data_to_arrays(load("data003_multienv"), env_col="env")
    with pytest.raises(bb.BarBayError, match="environments"):
        bb.model.multienv_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc, envs=[1, 2])
    with pytest.raises(bb.BarBayError, match="genotypes"):
        bb.model.genotype_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc, genotypes=["a"])
    m = bb.model.fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc, logλ_prior=[3.0, 3.0],
                                s_pop_prior=np.tile([0.0, 1.0], (6, 1)))
    assert m.priors["loglambda_prior"][0].shape == (1,) and m.priors["s_pop_prior"][0].shape == (6,)


def test_advi_argument_errors_without_gpu():
    """test/vi_tests.jl:196-207: the name-based checks fire before any device work."""
    df = load("data001_single")
    with pytest.raises(bb.BarBayError, match="rep_col"):
        bb.vi.advi(data=df, model=bb.model.replicate_fitness_normal, advi=bb.vi.ADVI(1, 1))
    with pytest.raises(bb.BarBayError, match="env_col"):
        bb.vi.advi(data=df, model=bb.model.multienv_fitness_normal, advi=bb.vi.ADVI(1, 1))


def _tidy_rep_env(ragged=False, seed=0):
    g = np.random.default_rng(seed)
    rows = []
    envs = {"R1": ["a", "a", "b", "c", "b"], "R2": ["a", "a", "b", "c", "b"] if not ragged else ["a", "c", "b", "b"]}
    for rep, ev in envs.items():
        for bc in [f"neutral{i:03d}" for i in range(3)] + [f"mut{i:03d}" for i in range(6)]:
            for t, e in enumerate(ev, start=1):
                rows.append(dict(time=t, env=e, barcode=bc, count=int(g.integers(200, 5000)), neutral=bc.startswith("neutral"), rep=rep))
    return pd.DataFrame(rows)


@pytest.mark.parametrize("ragged", [False, True])
def test_multienv_replicate_surface(ragged):
    """§8f rank 1: multienv_replicate_fitness_normal (3-D and ragged calls) through data_to_arrays -> model ->
    labels, no engine."""
    df = _tidy_rep_env(ragged)
    d = bb.utils.data_to_arrays(df, rep_col="rep", env_col="env")
    assert d.n_rep == 2 and d.n_env == 3
    bm = bb.model.multienv_replicate_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc, envs=d.envs)
    assert bm.kind == "multienv_replicate" and bm.ragged == ragged and len(bm.env_idx) == 2
    assert [len(e) for e in bm.env_idx] == [c.shape[0] for c in bm.counts]
    assert bm.env_idx[0].tolist() == [0, 0, 1, 2, 1]
    if ragged:
        assert bm.env_idx[1].tolist() == [0, 2, 1, 1]
    E, nb, R = 3, 6, 2
    nt1 = sum(c.shape[0] - 1 for c in bm.counts)
    nl = sum(c.size for c in bm.counts)
    sizes = [nt1, nt1, E * nb, E * nb * R, E * nb * R, E * nb * R, nl]
    ranges, o = [], 0
    for n in sizes:
        ranges.append((o, o + n))
        o += n
    q = SimpleNamespace(dist=SimpleNamespace(m=np.zeros(o), σ=np.ones(o)), transform=SimpleNamespace(ranges_out=ranges))
    names = []
    for sym, (lo, hi) in zip(bm.var_symbols(), ranges):
        names += [f"{sym}[{i}]" for i in range(1, hi - lo + 1)]
    out = bb.utils.advi_to_df(df, q, names, rep_col="rep", env_col="env", n_samples=50, rng=np.random.default_rng(0))
    base = out.iloc[:o]
    th = base[base.vartype == "bc_hyperfitness"]
    assert list(th["env"][:3]) == ["a", "b", "c"] and list(th["id"][:3]) == [d.bc_ids[0]] * 3
    assert len(base[base.vartype == "pop_mean_fitness"]["env"]) == nt1
    assert (out.iloc[o:].vartype == "bc_fitness").all() and len(out) == o + E * nb * R
    with pytest.raises(bb.BarBayError, match="environments"):
        bb.model.multienv_replicate_fitness_normal(d.bc_count, d.bc_total, d.n_neutral, d.n_bc, envs=["a"])


def test_advi_to_df_is_linear_in_barcodes():
    """The label columns are built in one pass (a sum-of-lists version took 40 s at 50 000 barcodes, dwarfing the 0.2 s device run)."""
    import time
    B, T = 30_000, 8
    g = np.random.default_rng(0)
    ids = np.array([f"bc{i:06d}" for i in range(B)])
    df = pd.DataFrame({"barcode": np.repeat(ids, T), "time": np.tile(np.arange(T), B), "count": g.integers(1, 1000, B * T),
                       "neutral": np.repeat(np.arange(B) < 600, T)})
    arr = bb.utils.data_to_arrays(df)
    bm = bb.model.fitness_normal(arr.bc_count, arr.bc_total, arr.n_neutral, arr.n_bc)
    n1, nb = T - 1, arr.n_bc
    ranges = [(0, n1), (n1, 2 * n1), (2 * n1, 2 * n1 + nb), (2 * n1 + nb, 2 * n1 + 2 * nb), (2 * n1 + 2 * nb, 2 * n1 + 2 * nb + T * B)]
    D = ranges[-1][1]
    q = SimpleNamespace(dist=SimpleNamespace(m=np.zeros(D), σ=np.ones(D)), transform=SimpleNamespace(ranges_out=ranges))
    names = []
    for sym, (lo, hi) in zip(bm.var_symbols(), ranges):
        names += [f"{sym}[{i}]" for i in range(1, hi - lo + 1)]
    t0 = time.perf_counter()
    out = bb.utils.advi_to_df(df, q, names)
    assert time.perf_counter() - t0 < 5.0
    assert len(out) == D and out["id"].iloc[-1] == ids[-1]
