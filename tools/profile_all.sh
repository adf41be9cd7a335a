#!/bin/bash
# Reproduces a profiles/<tag>/ directory on the GPU box:  bash tools/profile_all.sh r01f
# (1) bench line, (2) rocprofv3 kernel trace + stats of the same command, (3)+(4) PMC passes (FETCH_SIZE, WRITE_SIZE:
# separate runs, kernel-trace only, as MI355X_MICROARCH.md prescribes), (5) stamp shares, (6) the other configs.
set -eo pipefail
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 4000 --warmup 200 --no-cpu-baseline"
timeout -k 10 300 python3 "$ROOT/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"; tail -c 400 "$OUT/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH > "$OUT/trace.log" 2>&1
echo "trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "pmc write done"
# HBM traffic per step of the resident kernel from the two PMC passes -> profiles/hbm_traffic_latest.json (copied to $OUT too)
KERNEL=$(python3 -c "import json,sys; print(json.load(open('$OUT/bench.json'))['roofline']['kernel'])")
python3 "$ROOT/tools/hbm_traffic.py" "$OUT/pmc_fetch" "$OUT/pmc_write" "$KERNEL" 4000 "profiles/$TAG (tools/profile_all.sh)" > "$OUT/hbm_traffic.json"
cp "$ROOT/profiles/hbm_traffic_latest.json" "$OUT/hbm_traffic_latest.json"
echo "traffic done"
# phase stamps of the step (lib/ab/base_st.so = -DBB_STAMPS build of the same sources: python tools/xp.py build base)
timeout -k 10 300 python3 "$ROOT/tools/xp.py" run base > "$OUT/stamps.txt" 2>&1 || true
echo "stamps done"
timeout -k 10 200 python3 "$ROOT/tools/fixed_cost.py" > "$OUT/fixed_cost.txt" 2>&1 || true
echo "fixed cost done"
timeout -k 10 500 python3 "$ROOT/tools/bench_configs.py" > "$OUT/configs.txt" 2>&1 || true
tail -8 "$OUT/configs.txt"
