"""Import alias for the package directory ``barbay.jl_amd/`` (a dotted directory name is not
importable with a plain ``import`` statement).  ``import barbay_jl_amd`` yields that package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "barbay.jl_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
