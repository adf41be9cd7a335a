#!/bin/bash
# Second GPU call of a round's evidence (a call is limited to 20 minutes):  bash tools/round_end2.sh r03z
TAG=${1:-run}
O=gpurun_out
mkdir -p $O
# the other BASELINE configs: rocprofv3 kernel stats of one launch each; C5 on one GPU (k_stream, the bandwidth-bound one) also the FETCH / WRITE passes
ROOT=$(pwd); export TMPDIR=/tmp
for CFG in C3 C4 C5rank C5rank_plain C5; do
  ST=4000; [ $CFG = C5 ] && ST=1000
  (cd /tmp && CFG=$CFG STEPS=$ST timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/${TAG}_trace_$CFG -- python3 $ROOT/tools/kernel_trace_config.py) > $O/${TAG}_trace_$CFG.log 2>&1
done
(cd /tmp && CFG=C5 STEPS=1000 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$O/${TAG}_pmcf_C5 -- python3 $ROOT/tools/kernel_trace_config.py) > $O/${TAG}_pmcf_C5.log 2>&1
(cd /tmp && CFG=C5 STEPS=1000 timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$O/${TAG}_pmcw_C5 -- python3 $ROOT/tools/kernel_trace_config.py) > $O/${TAG}_pmcw_C5.log 2>&1
python3 tools/hbm_traffic.py $O/${TAG}_pmcf_C5 $O/${TAG}_pmcw_C5 k_stream 1000 "profiles/$TAG C5 on one GPU" $O/${TAG}_hbm_traffic_C5.json > /dev/null 2>&1
# ... and of C3 / C4 (k_res: only the window slot should reach the memory side)
for CFG in C3 C4; do
  (cd /tmp && CFG=$CFG STEPS=4000 timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$O/${TAG}_pmcf_$CFG -- python3 $ROOT/tools/kernel_trace_config.py) > $O/${TAG}_pmcf_$CFG.log 2>&1
  (cd /tmp && CFG=$CFG STEPS=4000 timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$O/${TAG}_pmcw_$CFG -- python3 $ROOT/tools/kernel_trace_config.py) > $O/${TAG}_pmcw_$CFG.log 2>&1
  python3 tools/hbm_traffic.py $O/${TAG}_pmcf_$CFG $O/${TAG}_pmcw_$CFG k_res 4000 "profiles/$TAG $CFG" $O/${TAG}_hbm_traffic_$CFG.json > /dev/null 2>&1
done
for CFG in C3 C4 C5rank; do CFG=$CFG timeout -k 10 300 python tools/validate_c2.py 3000 > $O/${TAG}_validate_$CFG.json 2>/dev/null; done
(WL=c2 timeout -k 10 200 python tools/p2p_rehearsal.py 2 2000 && WL=c2 timeout -k 10 200 python tools/p2p_rehearsal.py 4 2000 && WL=c4 timeout -k 10 200 python tools/p2p_rehearsal.py 4 2000 && WL=c5 SHAPE_W=8 timeout -k 10 200 python tools/p2p_rehearsal.py 4 1000 && WL=c2r SHAPE_W=8 timeout -k 10 200 python tools/p2p_rehearsal.py 4 2000) 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo\|peer ranks" > $O/${TAG}_p2p_rehearsal.txt
timeout -k 10 200 python tools/ms_rates.py 2>&1 | grep -v amdgpu > $O/${TAG}_ms_rates.txt
tail -n 12 $O/${TAG}_p2p_rehearsal.txt
# round 4: the N > 1 bench line in the driver's short form against long launches (one-GPU rehearsal: all ranks on device 0, gloo), k_stream on a
# replicate problem beyond the register file, phase stamps of C5 on one GPU and of C5's rank shape
for N in 2 4; do
  BB_BENCH_REHEARSAL=1 timeout -k 10 200 python bench.py --gpus $N --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_rehearsal${N}_20.json 2>/dev/null
  BB_BENCH_REHEARSAL=1 timeout -k 10 200 python bench.py --gpus $N --steps 4000 --warmup 200 --no-cpu-baseline > $O/${TAG}_rehearsal${N}_4000.json 2>/dev/null
done
timeout -k 10 300 python tools/stream_replicates.py 2>&1 | grep -v amdgpu > $O/${TAG}_stream_replicates.txt
WL=genotype_fitness_normal timeout -k 10 300 python tools/xp.py run base 2>&1 | grep -v amdgpu > $O/${TAG}_stamps_C5.txt
WL=genotype_fitness_normal B=25000 G=626 timeout -k 10 200 python tools/xp.py run base 2>&1 | grep -v amdgpu > $O/${TAG}_stamps_C5rank.txt
WL=replicate_fitness_normal timeout -k 10 200 python tools/xp.py run base 2>&1 | grep -v amdgpu > $O/${TAG}_stamps_C3.txt
WL=C5 timeout -k 10 400 python tools/ms_rates.py 2>&1 | grep -v amdgpu > $O/${TAG}_ms_rates_C5.txt
