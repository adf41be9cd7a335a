#!/bin/bash
# Instruction mix / stall counters of the resident launch on C2 (diagnostic): bash tools/pmc_mix.sh TAG
set -o pipefail
TAG=${1:-mix}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 2000 --warmup 100 --no-cpu-baseline"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- $BENCH > "$OUT/p$i.log" 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.OrderedDict()
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_res" in r["Kernel_Name"] or "k_persist" in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] = per[r["Counter_Name"]].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    for c, d in per.items():
        tot[c] = max(d.values())          # the timed 2000-step launch
for c, v in tot.items():
    print(f"{c:28s} {v:16.0f}   per step {v / 2000:14.1f}   per step per wave {v / 2000 / 4096:10.2f}")
PY
