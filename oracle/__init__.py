"""CPU oracle for the BarBay ADVI hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's algorithm for the one hot
path this repository accelerates (mean-field ADVI over the ``fitness_normal``
model family).  It exists to *check* the HIP engine.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; nothing under ``barbay.jl_amd/`` does, and the product path fails
loudly when the HIP library is missing.

PARITY UNPINNED.  The reference (mrazomej/BarBay.jl, Julia) cannot run in this
image (no ``julia``), its ADVI loop lives in third-party packages that are not
vendored (Turing 0.36 / AdvancedVI 0.2 / DynamicPPL 0.32 / Distributions 0.25,
``Project.toml:26-42``, no Manifest), and its own tests assert no numeric value
on this path (``test/vi_tests.jl`` runs one iteration and checks column names).
What pins this oracle instead:

* ``literal.py`` transcribes each model body statement by statement
  (file:line cited per function) and gets gradients from torch autograd, so
  it shares no algebra with the fused kernels;
* the distribution log-densities it uses are cross-checked against
  ``scipy.stats`` (an independent published implementation) in
  ``tests/test_oracle_literal.py``;
* ``c/bb_port.c`` (the fused, analytic-gradient C port used as the timed CPU
  baseline) is checked against ``literal.py`` to <=1e-12 relative.
"""
