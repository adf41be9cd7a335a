"""The C-ABI shared library loads without a GPU and exports every symbol include/barbay_hip.h
declares; the ctypes structs match the header's layout (checked through bb_default_opts)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "barbay_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bb_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    from barbay_jl_amd import _capi
    assert header_functions() == sorted(_capi.EXPORTS)


def test_library_exports_every_declared_symbol(hip_lib):
    for name in header_functions():
        assert hasattr(hip_lib, name), name
    assert b"gfx950" in hip_lib.bb_version()


def test_default_opts_roundtrip(hip_lib):
    from barbay_jl_amd import _capi
    o = _capi.bb_advi_opts()
    hip_lib.bb_default_opts(ctypes.byref(o))
    assert (o.samples_per_step, o.optimizer, o.window, o.world_size) == (1, 0, 100, 1)
    assert (o.eta, o.tau, o.pre, o.post) == (0.1, 40.0, 1.0, 0.9)


def test_missing_library_fails_loudly(tmp_path):
    from barbay_jl_amd import _capi
    with pytest.raises(_capi.BarBayHipError, match="no CPU fallback"):
        _capi.load_library(str(tmp_path / "nope.so"))


def test_product_does_not_import_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/ or the emulation build."""
    pkg = os.path.join(ROOT, "barbay.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libbb_emu" not in txt, f
