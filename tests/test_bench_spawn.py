"""CPU: `python bench.py --gpus N` started as a plain process spawns N ranks itself, touches no GPU in the parent, and
relays a failing rank as a non-zero exit instead of hanging (here every rank fails: this container has no GPU)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parent_spawns_and_reports_failure_without_hanging():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""           # also on a GPU box this test stays a CPU test
    env["CUDA_VISIBLE_DEVICES"] = ""
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t0 < 240


def test_parent_does_not_import_torch_before_spawning():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("raise SystemExit(spawn_ranks(args.gpus))")]
    assert "import torch" not in head.split("def main():")[1]
