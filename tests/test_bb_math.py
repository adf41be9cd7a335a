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


def test_emulation_under_address_sanitizer(tmp_path):
    """The block programs (host emulation build) run clean under ASan on a multi-tile, ragged case."""
    lib = str(tmp_path / "libbb_emu_asan.so")
    src = os.path.join(ROOT, "barbay.jl_amd", "csrc", "bb_engine.hip")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-DBB_EMU", "-fPIC",
                    "-shared", "-Wl,-Bsymbolic", "-x", "c++", src, "-o", lib], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    code = f"""
import ctypes, sys
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import _cases as c
from barbay_jl_amd import _capi
lib = _capi._declare(ctypes.CDLL({lib!r}))
c.case_synth_grad(lib, "replicate_ragged")
c.case_trajectory_exact(lib, "genotype", "TruncatedADAGrad", 2)
c.case_sharded_split_phase(lib, "multienv")
print("asan-ok")
"""
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, env=env)
    assert "asan-ok" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr, r.stderr[-2000:]
