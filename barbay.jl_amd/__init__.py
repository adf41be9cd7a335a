"""barbay.jl_amd -- MI355X-native ADVI engine behind BarBay.vi.advi() / BarBay.model.*.

The directory name carries a dot, so import it through the repo-root alias module
``barbay_jl_amd`` (``import barbay_jl_amd as bb``):

    bb.vi.advi(data=df, model=bb.model.fitness_normal, advi=bb.vi.ADVI(1, 10_000))
"""
from . import _capi  # noqa: F401
from ._capi import BarBayHipError, BarBayNonFinite, Engine, load_library  # noqa: F401
from . import dist, mcmc, model, sharding, stats, synth, utils, vi  # noqa: F401
from .model import BarBayError  # noqa: F401
