"""Accuracy of the kernels' fp64 elementary functions (barbay.jl_amd/csrc/bb_math.h, host build)
against long-double libm: tests/bb_math_check.cpp over 2M random arguments per function."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bb_math_accuracy(tmp_path):
    exe = str(tmp_path / "bb_math_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "bb_math_check.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    errs = dict(re.findall(r"(\w+) ([0-9.e+-]+)", out.splitlines()[0]))
    for name, bound in [("exp", 4e-16), ("log", 6e-16), ("rcp", 3e-16), ("div", 3e-16), ("sqrt", 3e-16),
                        ("softplus", 1e-15), ("sigmoid", 1e-15), ("sincospi", 3e-16)]:
        assert float(errs[name]) < bound, (name, errs[name])
    assert "sqrt0 0 exp(-800) 0 exp(800) inf" in out


