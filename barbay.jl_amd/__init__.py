"""barbay.jl_amd -- MI355X-native ADVI engine behind BarBay.vi.advi() / BarBay.model.*.

The directory name carries a dot, so import it through the repo-root alias module
``barbay_jl_amd`` (``import barbay_jl_amd as bb``).
"""
from . import _capi  # noqa: F401
from ._capi import BarBayHipError, Engine, load_library  # noqa: F401
