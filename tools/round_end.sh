#!/bin/bash
# Everything profiles/<tag>_* holds, in one GPU call:  bash tools/round_end.sh r02t
# GPU tests, smoke, tools/profile_all.sh, the instruction mix, bench variants, window accuracy, per-wave stamps, C2/8, validate_c2.
TAG=${1:-run}
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/${TAG}_tests.log 2>&1; echo "pytest rc=$?" >> $O/${TAG}_tests.log; tail -3 $O/${TAG}_tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/profile_all.sh $TAG > $O/${TAG}_profile.log 2>&1; echo "profile rc=$?" >> $O/${TAG}_profile.log; tail -2 $O/${TAG}_profile.log | cut -c1-150
bash tools/pmc_mix.sh ${TAG}_mix > $O/${TAG}_mix.txt 2>&1
timeout -k 10 300 python bench.py > $O/${TAG}_bench.json 2>/dev/null
timeout -k 10 30 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_bench20.json 2>/dev/null
BB_BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --no-cpu-baseline > $O/${TAG}_rehearsal2.json 2>/dev/null
timeout -k 10 300 python tools/resum_accuracy.py 2>&1 | grep -v amdgpu > $O/${TAG}_window_accuracy.txt
WAVES=100 timeout -k 10 100 python tools/xp.py run base 2>&1 | grep -v amdgpu > $O/${TAG}_waves.txt
timeout -k 10 100 python bench.py --barcodes 6250 --no-cpu-baseline 2>/dev/null > $O/${TAG}_c2_eighth.json
timeout -k 10 600 python tools/validate_c2.py > $O/${TAG}_validate_c2.json 2>/dev/null
cut -c1-160 $O/${TAG}_bench.json; cut -c1-160 $O/${TAG}_bench20.json
