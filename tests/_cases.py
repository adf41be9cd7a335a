"""Parity cases shared by the emulation tests (CPU, `-m "not gpu"`) and the GPU tests (`-m gpu`).
Each takes the loaded C-ABI library (`lib`) and compares the engine with the oracle."""
import os

import numpy as np

from conftest import make_engine
from oracle import advi, fixtures, literal, rng

GOLD = os.path.join(os.path.dirname(__file__), "golden")

# deterministic tolerances: fp64, |dELBO|/|ELBO| <= 1e-12, max|dgrad| <= 1e-9 max|grad|
ELBO_RTOL = 1e-12
GRAD_RTOL = 1e-9

SYNTH = {
    "fitness_multi_tile": ("fitness", dict(B=700, T=5, n_neutral=37)),
    "fitness_neutral_heavy": ("fitness", dict(B=513, T=8, n_neutral=300)),
    "fitness_T2": ("fitness", dict(B=130, T=2, n_neutral=3)),
    "fitness_T6": ("fitness", dict(B=700, T=6, n_neutral=37)),          # even T: the owner-computes resident launch (k_res), LPB 4 with an idle lane
    "fitness_T4": ("fitness", dict(B=333, T=4, n_neutral=70)),
    "fitness_wide_grid": ("fitness", dict(B=2100, T=6, n_neutral=90)),      # > 64 tiles of 16 barcodes: the exchange with 32 groups
    "multienv_T6": ("multienv", dict(B=600, T=6, n_env=3, n_neutral=11)),
    "multienv_T8": ("multienv", dict(B=300, T=8, n_env=4, n_neutral=40)),
    "replicate_T6": ("replicate", dict(B=301, T=6, n_rep=2, n_neutral=1)),                 # hierarchical kinds under k_res
    "replicate_R3": ("replicate", dict(B=400, T=[6, 8, 4], n_rep=3, n_neutral=20)),        # ... ragged (all T_r even)
    "replicate_R3_T6": ("replicate", dict(B=400, T=6, n_rep=3, n_neutral=20)),                # ... three replicates with the same T (C3's shape): k_stream where forced
    "multienv_replicate_R3_T8": ("multienv_replicate", dict(B=240, T=8, n_rep=3, n_env=3, n_neutral=8)),
    "replicate_R4": ("replicate", dict(B=240, T=[8, 8, 6, 6], n_rep=4, n_neutral=12)),           # K + 2 nt1 = 198 moment-row entries: more than a small tile's threads
    "multienv_replicate_T6": ("multienv_replicate", dict(B=150, T=6, n_rep=2, n_env=2, n_neutral=10)),
    "multienv_replicate_R3": ("multienv_replicate", dict(B=300, T=[6, 4, 8], n_rep=3, n_env=3, n_neutral=7)),
    "multienv": ("multienv", dict(B=600, T=7, n_env=3, n_neutral=11)),
    "genotype": ("genotype", dict(B=800, T=6, n_geno=17, n_neutral=256)),
    "genotype_runs": ("genotype", dict(B=800, T=6, n_geno=40, n_neutral=256, geno_runs=True)),   # mutants grouped by genotype: k_res owns whole genotypes per tile
    "genotype_T8": ("genotype", dict(B=1000, T=8, n_geno=60, n_neutral=30, geno_runs=True)),
    "genotype_T5": ("genotype", dict(B=600, T=5, n_geno=30, n_neutral=50, geno_runs=True)),          # odd number of time points: k_res any-parity instances
    "genotype_odd": ("genotype", dict(B=800, T=6, n_geno=41, n_neutral=256, geno_runs=True)),      # loglambda starts at an ODD flat index: a k_res pair is two Philox pairs' halves
    "replicate_odd": ("replicate", dict(B=302, T=[6, 4], n_rep=2, n_neutral=1)),                  # ... likewise (301 mutants, two replicates): k_persist
    "replicate_ragged": ("replicate", dict(B=530, T=[5, 7, 4], n_rep=3, n_neutral=20)),
    "replicate_3d": ("replicate", dict(B=300, T=6, n_rep=2, n_neutral=1)),
    "multienv_replicate": ("multienv_replicate", dict(B=420, T=[5, 7, 4], n_rep=3, n_env=3, n_neutral=23)),
    "multienv_replicate_3d": ("multienv_replicate", dict(B=150, T=5, n_rep=2, n_env=2, n_neutral=9)),
}


def synth(name, seed=5, **pri):
    kind, kw = SYNTH[name]
    return fixtures.synthetic(kind, seed=seed, **kw, **pri)


def caller_normals(e, seed, step, stream, D):
    """The engine's normals of (step, stream) in the CALLER's latent order: the stream is keyed by the handle's internal index, which
    differs from the caller's only where the genotype model's mutants were regrouped inside the library (bb_get_permutation)."""
    n = rng.normals(seed, step, stream, D)
    out = np.empty(D)
    out[e.permutation()] = n
    return out


def check_grad(e, sp, mu, om, eps):
    el, gm, go = literal.elbo_and_grad(mu, om, eps, sp)
    el2, gm2, go2 = e.elbo_grad(mu, om, eps)
    assert abs(el - el2) <= 50 * ELBO_RTOL * abs(el), (el, el2)   # lgamma/sum order: a few ulps of 1e9-sized sums
    assert np.abs(gm - gm2).max() <= GRAD_RTOL * np.abs(gm).max()
    assert np.abs(go - go2).max() <= GRAD_RTOL * np.abs(go).max()


def case_golden(lib, name):
    gold = np.load(os.path.join(GOLD, f"golden_{name}.npz"))
    sp = fixtures.load(name)
    with make_engine(sp, lib) as e:
        assert e.D == int(gold["D"])
        lay = e.layout()
        assert [(n, lo, hi) for n, (lo, hi) in sp.offsets().items()] == lay
        el, gm, go = e.elbo_grad(gold["mu"], gold["omega"], gold["eps"])
        assert abs(el - float(gold["elbo"])) <= 50 * ELBO_RTOL * abs(el)
        assert np.abs(gm - gold["grad_mu"]).max() <= GRAD_RTOL * np.abs(gold["grad_mu"]).max()
        assert np.abs(go - gold["grad_omega"]).max() <= GRAD_RTOL * np.abs(gold["grad_omega"]).max()
        # log-joint and its gradient: eps = 0 => z = mu, dELBO/dmu = grad logjoint, ELBO = logjoint + H
        z = gold["z"]
        om = np.full(sp.D, -1.0)
        H = 0.5 * sp.D * (1 + np.log(2 * np.pi)) + np.log(advi.softplus(om)).sum()
        el0, g0, _ = e.elbo_grad(z, om, np.zeros((1, sp.D)))
        assert abs((el0 - H) - float(gold["logjoint"])) <= 50 * ELBO_RTOL * abs(el0)
        assert np.abs(g0 - gold["grad_z"]).max() <= GRAD_RTOL * np.abs(gold["grad_z"]).max()


def case_synth_grad(lib, name):
    sp = synth(name)
    with make_engine(sp, lib, seed=9) as e:
        mu0, om0 = advi.meanfield_init(9, sp.D)
        m, o = e.get_params()
        cidx = e.permutation()
        assert np.abs(m[cidx] - mu0).max() < 1e-13 and np.abs(o[cidx] - om0).max() < 1e-13
        mu0, om0 = m, o
        mu, om = mu0 * 0.2 + 3, om0 * 0.5 - 2
        eps = np.stack([rng.normals(9, 4, s, sp.D) for s in range(3)])
        check_grad(e, sp, mu, om, eps)
        # engine's own Philox stream (eps = NULL) at its current step (0), two samples
        e.set_params(mu, om)
        el3, gm3, go3 = e.elbo_grad(mu, om, None, 2)
        eps0 = np.stack([caller_normals(e, 9, 0, s, sp.D) for s in range(2)])
        el4, gm4, go4 = literal.elbo_and_grad(mu, om, eps0, sp)
        assert abs(el3 - el4) <= 1e-10 * abs(el4)
        assert np.abs(gm3 - gm4).max() <= 1e-8 * np.abs(gm4).max()


def case_normals(lib):
    sp = fixtures.load("data001_single")
    with make_engine(sp, lib, seed=0xDEADBEEF12345) as e:
        for (step, stream, lo, hi) in [(0, 0, 0, 103), (7, 1, 3, 50), (123456, rng.STREAM_INIT_MU, 10, 11), (2 ** 32 - 1, 5, 0, 2)]:
            got = e.normals(step, stream, lo, hi)
            want = rng.normals(0xDEADBEEF12345, step, stream, 103, lo, hi)
            assert np.abs(got - want).max() < 1e-13
        big = e.normals(3, 0, 0, 400000)
        assert abs(big.mean()) < 0.01 and abs(big.std() - 1) < 0.01


def _trajectory(lib, sp, nsteps, S, optname, seed=11, use_priors=False, **ekw):
    e = make_engine(sp, lib, use_priors=use_priors, seed=seed, samples_per_step=S, optimizer=optname, **ekw)
    mu0, om0 = e.get_params()
    e.run(nsteps)
    mu, om = e.get_params()
    opt = advi.TruncatedADAGrad(n=ekw.get("window", 100)) if optname == "TruncatedADAGrad" else advi.DecayedADAGrad()
    f = lambda m, o, eps: literal.elbo_and_grad(m, o, eps, sp)
    eps_fn = lambda i: np.stack([caller_normals(e, seed, i, s, sp.D) for s in range(S)])
    m2, o2, tr = advi.run_advi(sp, f, mu0, om0, nsteps, S, opt, seed, eps_fn=eps_fn)
    return e, np.abs(mu - m2).max(), np.abs(om - o2).max(), tr


def case_trajectory_exact(lib, name, optname, S, tol=1e-10, **ekw):
    """resum_every=1: the window is re-added every step, the reference's own arithmetic."""
    sp = synth(name, seed=2)
    e, a, b, _ = _trajectory(lib, sp, 12, S, optname, window=5, resum_every=1, **ekw)
    k = e.stats()["resident_kernel"]
    e.close()
    assert a < tol and b < tol, (a, b)
    return k


def case_trajectory_running(lib, name, graph=0, S=1, **ekw):
    """default running-window sum (exact re-add once per window) + ELBO trace."""
    sp = synth(name, seed=2)
    e, a, b, tr = _trajectory(lib, sp, 23, S, "TruncatedADAGrad", window=5, elbo_every=1, steps_per_graph=graph, **ekw)
    got = e.elbo_trace(0, 23)
    k = e.stats()["resident_kernel"]
    e.close()
    assert a < 1e-6 and b < 1e-6, (a, b)
    assert np.abs(got - tr).max() <= 1e-6 * np.abs(tr).max()
    return k


def case_matrix_priors(lib):
    sp0 = synth("fitness_multi_tile", seed=3)
    g = np.random.default_rng(0)
    T = sp0.n_time[0]
    pri = dict(loglambda_prior=(np.log(sp0.counts[0].T.reshape(-1) + 1.0), g.uniform(0.5, 3, sp0.B * T)),
               s_pop_prior=(g.normal(0, 1, T - 1), g.uniform(0.05, 1, T - 1)),
               s_bc_prior=(g.normal(0, 1, sp0.n_bc), g.uniform(0.5, 2, sp0.n_bc)),
               logsigma_bc_prior=(0.5, 0.7),
               logsigma_pop_prior=(g.normal(0, 1, T - 1), g.uniform(0.5, 1, T - 1)))
    sp = synth("fitness_multi_tile", seed=3, **pri)
    e, a, b, tr = _trajectory(lib, sp, 8, 1, "TruncatedADAGrad", use_priors=True, elbo_every=1, resum_every=1, window=4)
    got = e.elbo_trace(0, 8)
    e.close()
    assert a < 1e-10 and b < 1e-10
    assert np.abs(got - tr).max() <= 1e-11 * np.abs(tr).max()


def case_sharded_split_phase(lib, name, W=3, S=2, nsteps=5):
    """Barcode shards driven through bb_step_moments / bb_step_apply with a caller-side sum equal
    the unsharded run (up to summation order)."""
    sp = synth(name, seed=4)
    with make_engine(sp, lib, seed=5, samples_per_step=S, window=4) as e1:
        e1.run(nsteps)
        m1, o1 = e1.get_params()
        lay = {n: (lo, hi) for n, lo, hi in e1.layout()}
    es = [make_engine(sp, lib, seed=5, samples_per_step=S, window=4, rank=r, world_size=W) for r in range(W)]
    for _ in range(nsteps * S):
        tot = sum(e.step_moments() for e in es)
        for e in es:
            e.step_apply(tot)
    from barbay_jl_amd.sharding import gather_params
    own = [e.owned() for e in es]          # (shard ranges are in the handles' internal order, the vectors in the caller's)
    mu = gather_params([e.get_params()[0] for e in es], [e.stats() for e in es], sp.kind, lay, sp.n_neutral, sp.n_bc,
                       sp.n_time, sp.n_rep, sp.n_env, owned=own)
    om = gather_params([e.get_params()[1] for e in es], [e.stats() for e in es], sp.kind, lay, sp.n_neutral, sp.n_bc,
                       sp.n_time, sp.n_rep, sp.n_env, owned=own)
    for e in es:
        e.close()
    assert np.abs(mu - m1).max() < 1e-10 and np.abs(om - o1).max() < 1e-10


def case_errors(lib):
    import barbay_jl_amd as bb
    import pytest
    sp = fixtures.load("data001_single")
    bad = [t.copy() for t in sp.totals]
    bad[0][1] += 1
    with pytest.raises(bb.BarBayHipError, match="totals"):
        bb.Engine(sp.kind, sp.counts, sp.n_neutral, sp.n_bc, totals=bad, _lib=lib)
    with pytest.raises(bb.BarBayHipError, match="Matrix form"):
        bb.Engine(sp.kind, sp.counts, sp.n_neutral, sp.n_bc, priors={"s_bc_prior": (np.zeros(3), np.ones(3))}, _lib=lib)
    with pytest.raises(bb.BarBayHipError, match="std"):
        bb.Engine(sp.kind, sp.counts, sp.n_neutral, sp.n_bc, priors={"s_bc_prior": (0.0, -1.0)}, _lib=lib)
    with pytest.raises(bb.BarBayHipError):
        bb.Engine("multienv", sp.counts, sp.n_neutral, sp.n_bc, _lib=lib)   # env_idx missing
    with pytest.raises(bb.BarBayHipError):
        bb.Engine("fitness", sp.counts, sp.n_neutral, sp.n_bc, samples_per_step=0, _lib=lib)


def case_persistent_equals_two_kernel(lib, name, tol=1e-11, expect_kernel=None, **geom):
    """launch_mode 2 (one resident launch, state in registers, grid barrier per step) and launch_mode 1
    (two kernels per sample) run the same arithmetic; also against the oracle with the exact window."""
    sp = synth(name, seed=6)
    outs = []
    for mode in (1, 2):
        # exact window (resum_every=1): with the running window, 1e-16 differences between the two code objects
        # are amplified by the cancellation when the huge first gradients leave the window
        with make_engine(sp, lib, seed=13, window=6, resum_every=1, launch_mode=mode, **geom) as e:
            mu0, om0 = e.get_params()
            e.run(7)          # odd count, then a second call: state must survive leaving / re-entering the launch
            e.run(10)
            outs.append(e.get_params())
            assert e.stats()["steps_done"] == 17
            if mode == 2 and expect_kernel is not None:
                assert e.stats()["resident_kernel"] == expect_kernel, e.stats()
    assert np.abs(outs[0][0] - outs[1][0]).max() < tol and np.abs(outs[0][1] - outs[1][1]).max() < tol
    e, a, b, _ = _trajectory(lib, sp, 9, 1, "TruncatedADAGrad", seed=13, window=4, resum_every=1, launch_mode=2)
    e.close()
    assert a < 1e-10 and b < 1e-10, (a, b)


def case_genotype_regrouped(lib, name="genotype_runs", seed=6):
    """geno_idx as the reference hands it over (barcodes in order of appearance, a genotype's mutants scattered;
    utils.data_to_arrays, src/utils.jl:692-731): the library groups the mutants itself, runs the resident launch and presents the
    caller's order.  The scatter here interleaves the genotype runs and keeps every genotype's mutants in their relative order, so
    that the library's stable grouping restores exactly the sorted problem: same draws, bit-equal results after mapping back."""
    import dataclasses
    sp = synth(name, seed=seed)
    g = np.random.default_rng(3)
    gi = np.asarray(sp.geno_idx)
    order = np.argsort(g.random(sp.n_bc) + 1e-9 * np.arange(sp.n_bc), kind="stable")       # a random merge ...
    keys = np.sort(g.random(sp.n_bc))
    # ... that keeps each genotype's mutants in order: give mutant m the m-th smallest key among its genotype's draws
    shuffled = np.empty(sp.n_bc, dtype=np.int64)       # shuffled[j] = sorted-problem mutant at caller position j
    pos_keys = g.random(sp.n_bc)
    for gg in np.unique(gi):
        mem = np.nonzero(gi == gg)[0]
        pos_keys[mem] = np.sort(pos_keys[mem])
    shuffled = np.argsort(pos_keys, kind="stable")
    nn = sp.n_neutral
    cols = np.concatenate([np.arange(nn), nn + shuffled])
    sp2 = dataclasses.replace(sp, counts=[c[:, cols] for c in sp.counts], geno_idx=gi[shuffled])
    assert not np.all(np.diff(np.asarray(sp2.geno_idx)) >= 0)
    kw = dict(seed=13, window=6, resum_every=1)
    with make_engine(sp, lib, launch_mode=2, **kw) as e:
        e.run(11)
        ref = e.get_params()
        cidx_ref = e.permutation()
        # nothing to regroup here; the library's internal order still differs from the caller's where loglambda would start at an odd
        # flat index (n_geno + n_bc odd): it then sits in front of the theta block (bb_create)
        assert (cidx_ref == np.arange(sp.D)).all() == (sp.offsets()["loglambda"][0] % 2 == 0)
    with make_engine(sp2, lib, launch_mode=2, **kw) as e:
        assert e.stats()["resident_kernel"] == 2
        cidx = e.permutation()
        mu0, om0 = e.get_params()
        # gradient at a fixed point with explicit draws, in the caller's order, against the literal oracle on the scattered problem
        eps = np.stack([rng.normals(5, 1, s_, sp.D) for s_ in range(2)])
        check_grad(e, sp2, mu0 * 0.3 + 2, om0 * 0.5 - 1, eps)
        e.run(11)
        got = e.get_params()
        med, sd = e.hier_fitness(300, seed=4)
    # internal latent i of the scattered handle IS internal latent i of the sorted problem's handle
    assert (got[0][cidx] == ref[0][cidx_ref]).all() and (got[1][cidx] == ref[1][cidx_ref]).all()
    off = sp.offsets()
    lo_tt = off["theta_tilde"][0]
    with make_engine(sp, lib, launch_mode=2, **kw) as e:
        e.run(11)
        med1, sd1 = e.hier_fitness(300, seed=4)
    pu = cidx[(cidx >= lo_tt) & (cidx < lo_tt + sp.n_bc)] - lo_tt          # (the theta_tilde block in internal order)
    assert (med[pu] == med1).all() and (sd[pu] == sd1).all()
    # trajectory against the oracle loop on the scattered problem (draws mapped through the permutation)
    e, a, b, _ = _trajectory(lib, sp2, 9, 1, "TruncatedADAGrad", seed=13, window=4, resum_every=1, launch_mode=2)
    e.close()
    assert a < 1e-10 and b < 1e-10, (a, b)


def case_ragged_method(lib, launch_mode=0):
    """BB_FLAG_RAGGED_METHOD reproduces the ragged replicate method's neutral pairing exactly as the reference
    writes it (oracle/literal.py ragged_quirk=True, model_fitness_normal_hierarchical_replicates.jl:596-610)."""
    for name, seed in (("replicate_ragged", 5), ("replicate_3d", 6)):
        sp = synth(name, seed=seed)
        sp2 = fixtures.synthetic("replicate", B=90, T=[4, 6, 3], n_rep=3, n_neutral=37, seed=seed) if name == "replicate_ragged" else sp
        for spx in (sp, sp2):
            with make_engine(spx, lib, seed=9, ragged_method=True, window=5, resum_every=1, launch_mode=launch_mode) as e:
                mu0, om0 = advi.meanfield_init(9, spx.D)
                mu, om = mu0 * 0.2 + 3, om0 * 0.5 - 2
                eps = np.stack([rng.normals(9, 4, s, spx.D) for s in range(2)])
                el, gm, go = literal.elbo_and_grad(mu, om, eps, spx, ragged_quirk=True)
                el2, gm2, go2 = e.elbo_grad(mu, om, eps)
                assert abs(el - el2) <= 50 * ELBO_RTOL * abs(el), (el, el2)
                assert np.abs(gm - gm2).max() <= GRAD_RTOL * np.abs(gm).max()
                assert np.abs(go - go2).max() <= GRAD_RTOL * np.abs(go).max()
                # and it differs from the consistent pairing
                el3, _, _ = literal.elbo_and_grad(mu, om, eps, spx)
                if spx.n_neutral > 1:      # the pairings coincide when n_neutral == 1 (or T_r == 2)
                    assert abs(el3 - el2) > 1e-9 * abs(el2)
                # optimiser trajectory against the oracle loop
                m0, o0 = e.get_params()
                e.run(9)
                m1, o1 = e.get_params()
                f = lambda m, o, ee: literal.elbo_and_grad(m, o, ee, spx, ragged_quirk=True)
                m2, o2, _ = advi.run_advi(spx, f, m0, o0, 9, 1, advi.TruncatedADAGrad(n=5), 9)
                assert np.abs(m1 - m2).max() < 1e-10 and np.abs(o1 - o2).max() < 1e-10


def case_hier_fitness(lib, name):
    """bb_hier_fitness (device-side `process_hierarchical_samples!`, src/utils.jl:1284-1343) against the oracle on the
    same Philox draws (exact order statistics: median of an even / odd sample count), and against a large numpy sample."""
    sp = synth(name, seed=3)
    with make_engine(sp, lib, seed=4) as e:
        e.run(3)
        mean, sigma = e.posterior()
        off = sp.offsets()
        (lo_th, hi_th), (lo_tt, hi_tt), (lo_lt, _) = off["theta"], off["theta_tilde"], off["logtau"]
        n_units = hi_tt - lo_tt
        idx = np.asarray(sp.geno_idx) if sp.kind == "genotype" else np.arange(n_units) % (hi_th - lo_th)
        # (the per-unit draws are keyed by the handle's INTERNAL unit number: pu[u'] = the caller's unit of internal unit u')
        cidx = e.permutation()
        pu = cidx[(cidx >= lo_tt) & (cidx < hi_tt)] - lo_tt          # (the theta_tilde block in internal order)
        for n in (1000, 777):
            med, sd = e.hier_fitness(n, seed=21)
            med2, sd2 = rng.hier_fitness(21, n, idx[pu][:40], mean[lo_th:hi_th], sigma[lo_th:hi_th], mean[lo_lt:lo_lt + n_units][pu],
                                         sigma[lo_lt:lo_lt + n_units][pu], mean[lo_tt:hi_tt][pu], sigma[lo_tt:hi_tt][pu])
            assert med.shape == (n_units,)
            assert np.abs(med[pu][:40] - med2).max() < 1e-10 and np.abs(sd[pu][:40] - sd2).max() < 1e-10
        med, sd = e.hier_fitness(10_000, seed=5)
        g = np.random.default_rng(0)
        u = 7
        big = (g.normal(mean[lo_th + idx[u]], sigma[lo_th + idx[u]], 400_000)
               + np.exp(g.normal(mean[lo_lt + u], sigma[lo_lt + u], 400_000)) * g.normal(mean[lo_tt + u], sigma[lo_tt + u], 400_000))
        assert abs(med[u] - np.median(big)) < 6 * 1.2533 * big.std() / np.sqrt(10_000)
        # (std of a lognormal-scaled product is tail-dominated; it is pinned draw for draw above, not statistically)


def case_logdensity(lib, name):
    """bb_logdensity_grad against the literal oracle's log-joint and autograd gradient."""
    sp = synth(name, seed=6)
    g = np.random.default_rng(8)
    with make_engine(sp, lib, seed=1) as e:
        mu0, om0 = e.get_params()
        for scale in (0.3, 1.0):
            z = g.normal(0.0, scale, sp.D)
            lp, gr = e.logdensity_grad(z)
            lp2, gr2 = literal.logjoint_and_grad(z, sp)
            assert abs(lp - lp2) <= 1e-11 * abs(lp2), (lp, lp2)
            assert np.abs(gr - gr2).max() <= 1e-9 * np.abs(gr2).max()
        mu1, om1 = e.get_params()
        assert (mu0 == mu1).all() and (om0 == om1).all()      # the variational state is untouched


def case_p2p_resident(lib, name, world, steps=7, **ekw):
    """Sharded resident launch (bb_p2p_*): `world` handles of one process, stepped in lock step by the emulation's
    bb_emu_run_group, against the unsharded run -- rows cross ranks through the inboxes exactly as on xGMI.
    ekw: samples_per_step / elbo_every -- the MS instances sharded (round 4); the ELBO trace of every rank equals the unsharded one's."""
    import ctypes as C
    from barbay_jl_amd import sharding
    sp = synth(name, seed=4)
    kw = dict(seed=5, window=4, resum_every=1, **ekw)
    refs = []
    trace1 = None
    with make_engine(sp, lib, launch_mode=1, **kw) as e1:
        for _ in range(2):                                             # second pass: restarted from the initial state
            e1.init_meanfield()
            e1.run(steps)
            refs.append(e1.get_params())
        if ekw.get("elbo_every"):
            trace1 = e1.elbo_trace(0, steps // ekw["elbo_every"])
    es = [make_engine(sp, lib, rank=r, world_size=world, **kw) for r in range(world)]
    try:
        handles = [e.p2p_export() for e in es]
        for e in es:
            e.p2p_import(handles)
        assert all(e.p2p_selftest() for e in es)
        assert all(e.p2p_enable(True) for e in es)
        arr = (C.c_void_p * world)(*[e._h for e in es])
        lib.bb_emu_run_group.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        lay = {n: (lo, hi) for n, lo, hi in es[0].layout()}
        for m1, o1 in refs:                                            # the restart must not meet the first pass's inbox words
            for e in es:
                e.init_meanfield()
            for n in (3, steps - 3):                                   # two "launches": inbox words keep counting
                assert lib.bb_emu_run_group(arr, world, n) == 0, lib.bb_last_error()
            per, st = zip(*[(e.get_params(), e.stats()) for e in es])
            own = [e.owned() for e in es]
            for i, ref in ((0, m1), (1, o1)):
                full = sharding.gather_params([p[i] for p in per], st, sp.kind, lay, sp.n_neutral, sp.n_bc, sp.n_time, sp.n_rep, sp.n_env, owned=own)
                assert np.abs(full - ref).max() < 1e-10
        if sp.kind == "genotype":       # theta_g moved on its owner only; the end of the run brought every copy up to date
            tlo, thi = lay["theta"]
            for p in per[1:]:
                assert (p[0][tlo:thi] == per[0][0][tlo:thi]).all() and (p[1][tlo:thi] == per[0][1][tlo:thi]).all()
            assert [s["geno_lo"] for s in st][1:] == [s["geno_hi"] for s in st][:-1] and st[0]["geno_lo"] == 0 and st[-1]["geno_hi"] == sp.n_geno
        # the replicated global blocks agree bit for bit on every rank
        glo = lay["s_pop"][0], lay["logsigma_pop"][1]
        for p in per[1:]:
            assert (p[0][glo[0]:glo[1]] == per[0][0][glo[0]:glo[1]]).all() and (p[1][glo[0]:glo[1]] == per[0][1][glo[0]:glo[1]]).all()
        assert all(s["persistent_pairs"] >= 1 for s in st)
        if trace1 is not None:
            for e in es:
                tr = e.elbo_trace(0, len(trace1))
                assert np.abs(tr - trace1).max() <= 1e-10 * np.abs(trace1).max(), (tr, trace1)
        assert all(e.p2p_enable(False) for e in es)
    finally:
        for e in es:
            e.close()


def case_multi_device_handle(lib, name, n=2, steps=9, expect_resident=True, **ekw):
    """bb_advi_opts.n_devices > 1: ONE handle, one host thread, n shards (here all on device 0) -- resident
    launches with in-process peer-mapped inboxes; posterior, layout and stats as from a single-device handle."""
    sp = synth(name, seed=4)
    kw = dict(seed=5, window=4, resum_every=1)
    for k in ("samples_per_step", "elbo_every"):          # (what changes the arithmetic goes to the reference run too)
        if k in ekw:
            kw[k] = ekw.pop(k)
    with make_engine(sp, lib, launch_mode=1, **kw) as e1:
        e1.run(steps)
        m1, s1 = e1.posterior()
        p1 = e1.get_params()
        lay1 = e1.layout()
        trace1 = e1.elbo_trace(0, steps // kw["elbo_every"]) if kw.get("elbo_every") else None
    with make_engine(sp, lib, device_ids=[0] * n, **kw, **ekw) as e:
        assert e.layout() == lay1 and e.D == sp.D
        for _ in range(2):                      # the second pass restarts with the inboxes still holding the first's words
            e.init_meanfield()
            e.run(3)
            e.run(steps - 3)
        st = e.stats()
        m, s = e.posterior()
        mu, om = e.get_params()
        assert st["steps_done"] == steps and (st["shard_lo"], st["shard_hi"]) == (0, sp.n_neutral + sp.n_bc)
        assert (st["resident_kernel"] > 0) == bool(expect_resident), st
        assert np.abs(m - m1).max() < 1e-9 and np.abs(s - s1).max() < 1e-9
        assert np.abs(mu - p1[0]).max() < 1e-9 and np.abs(om - p1[1]).max() < 1e-9
        if trace1 is not None:
            tr = e.elbo_trace(0, len(trace1))
            assert np.abs(tr - trace1).max() <= 1e-10 * np.abs(trace1).max(), (tr, trace1)
        if e.hier_units() > 0:                  # the derived-fitness sampler sees the whole posterior through the group handle
            with make_engine(sp, lib, launch_mode=1, **kw) as eh:
                eh.set_params(mu, om)
                med1, sd1 = eh.hier_fitness(500, seed=3)
            med, sd = e.hier_fitness(500, seed=3)
            assert np.abs(med - med1).max() < 1e-12 and np.abs(sd - sd1).max() < 1e-12
        # restart from explicit parameters on every shard
        e.set_params(p1[0], p1[1])
        e.run(2)
        e1b = make_engine(sp, lib, launch_mode=1, **kw)
        e1b.set_params(p1[0], p1[1])
        e1b.run(2)
        assert np.abs(e.get_params()[0] - e1b.get_params()[0]).max() < 1e-9
        e1b.close()
