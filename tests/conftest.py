import ctypes
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """Sequential host emulation of the HIP block programs (test-only; built with g++ -DBB_EMU)."""
    import __graft_entry__ as g
    from barbay_jl_amd import _capi
    return _capi._declare(ctypes.CDLL(g.build_emu()))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library; on a GPU box it must be the prebuilt in-tree .so."""
    import __graft_entry__ as g
    from barbay_jl_amd import _capi
    if not os.path.exists(g.LIB):
        g.build_hip()
    return _capi.load_library()


def make_engine(sp, lib=None, use_priors=False, **kw):
    """kw may carry ragged_method=True (the reference's Vector{Matrix} replicate method)."""
    import barbay_jl_amd as bb
    return bb.Engine(sp.kind, sp.counts, sp.n_neutral, sp.n_bc, env_idx=sp.env_idx, geno_idx=sp.geno_idx,
                     priors=sp.priors if use_priors else None, _lib=lib, **kw)
