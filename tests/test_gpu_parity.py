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


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "genotype", "replicate_ragged"])
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


@pytest.mark.parametrize("name", ["fitness_multi_tile", "multienv", "replicate_ragged"])
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
