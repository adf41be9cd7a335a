"""The block programs' host emulation under AddressSanitizer (GPU ASan is not available on the pool): LDS carve-up, per-thread
LDS slots, exchange rows and the multi-rank inboxes are all plain host memory there, so an index that strays is caught."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "barbay.jl_amd", "csrc", "bb_engine.hip")
OUT = os.path.join(ROOT, "tests", "_emu", "libbb_emu_asan.so")

SCRIPT = r'''
import ctypes, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from barbay_jl_amd import _capi
import _cases as c
lib = _capi._declare(ctypes.CDLL(%(lib)r))
os.environ["BB_TUNE_NB"] = "16"; os.environ["BB_TUNE_NTHR"] = "512"
c.case_p2p_resident(lib, "fitness_multi_tile", 3, steps=5)
c.case_p2p_resident(lib, "multienv_replicate", 2, steps=5)
c.case_p2p_resident(lib, "multienv_T8", 2, steps=5)                       # k_res (bb_resident.h), sharded
c.case_p2p_resident(lib, "genotype_runs", 3, steps=5)                     # k_res, genotype model: tiles / shards own whole genotypes
del os.environ["BB_TUNE_NB"]; del os.environ["BB_TUNE_NTHR"]
c.case_persistent_equals_two_kernel(lib, "replicate_ragged")
c.case_persistent_equals_two_kernel(lib, "fitness_T6", expect_kernel=2)   # k_res: LPB 4 with an idle lane per barcode
c.case_persistent_equals_two_kernel(lib, "multienv_T8", expect_kernel=2)
c.case_persistent_equals_two_kernel(lib, "replicate_R3", expect_kernel=2)           # k_res, hierarchical, ragged
c.case_persistent_equals_two_kernel(lib, "multienv_replicate_R3", expect_kernel=2)
c.case_persistent_equals_two_kernel(lib, "genotype_T8", expect_kernel=2)
os.environ["BB_TUNE_NB"] = "40"; os.environ["BB_TUNE_NTHR"] = "128"        # two pair slots per thread, tiles ending inside a wave
c.case_persistent_equals_two_kernel(lib, "fitness_T4", expect_kernel=2)
del os.environ["BB_TUNE_NB"]; del os.environ["BB_TUNE_NTHR"]
c.case_hier_fitness(lib, "genotype")
c.case_synth_grad(lib, "replicate_ragged")                                # the two-kernel block programs
c.case_trajectory_exact(lib, "genotype", "TruncatedADAGrad", 2)
c.case_sharded_split_phase(lib, "multienv")
print("ASAN-CLEAN")
'''


def test_emulated_engine_under_asan():
    libasan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not found")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [os.path.join(os.path.dirname(SRC), f) for f in os.listdir(os.path.dirname(SRC))]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer", "-DBB_EMU", "-fPIC",
                        "-shared", "-Wl,-Bsymbolic", "-x", "c++", SRC, "-o", OUT], check=True)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT, "lib": OUT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ASAN-CLEAN" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
