#!/usr/bin/env python3
"""Copies what tools/round_end.sh + tools/round_end2.sh left under gpurun_out/ into profiles/<dir> (the tracked summaries) and prints the
headline numbers:   python tools/collect_profiles.py r03z r03z_round3_final"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
G, D = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles", name)
os.makedirs(D, exist_ok=True)


def cp(src, dst):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(D, dst))
    else:
        print("missing", src)


def newest(pattern):
    fs = glob.glob(os.path.join(G, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


for src, dst in ((f"{tag}_bench.json", "bench.json"), (f"{tag}_bench20.json", "bench_20_steps_warmup_5.json"), (f"{tag}_rehearsal2.json", "bench_2ranks_one_gpu_rehearsal.json"),
                 (f"{tag}_c2_eighth.json", "bench_6250_barcodes.json"), (f"{tag}/configs.txt", "configs.txt"), (f"{tag}/fixed_cost.txt", "fixed_cost.txt"),
                 (f"{tag}/stamps.txt", "stamps.txt"), (f"{tag}/hbm_traffic.json", "hbm_traffic.json"), (f"{tag}_hbm_traffic_C5.json", "hbm_traffic_C5_k_stream.json"), (f"{tag}_hbm_traffic_C3.json", "hbm_traffic_C3.json"), (f"{tag}_hbm_traffic_C4.json", "hbm_traffic_C4.json"),
                 (f"{tag}_mix.txt", "pmc_instruction_mix.txt"), (f"{tag}_waves.txt", "per_wave_stamps.txt"), (f"{tag}_window_accuracy.txt", "window_accuracy.txt"),
                 (f"{tag}_validate_c2.json", "validate_c2_10000_iterations.json"), (f"{tag}_p2p_rehearsal.txt", "p2p_rehearsal_one_gpu.txt"),
                 (f"{tag}_ms_rates.txt", "ms_rates.txt"), (f"{tag}_tests.log", "gpu_tests.log")):
    cp(src, dst)
for c in ("C3", "C4", "C5rank"):
    cp(f"{tag}_validate_{c}.json", f"validate_{c}_3000_iterations.json")
for n in (2, 4):          # (round 4: the N > 1 bench line, short form against long launches -- one-GPU rehearsal)
    cp(f"{tag}_rehearsal{n}_20.json", f"bench_{n}ranks_one_gpu_rehearsal_20_steps_warmup_5.json")
    cp(f"{tag}_rehearsal{n}_4000.json", f"bench_{n}ranks_one_gpu_rehearsal_4000_steps.json")
for src, dst in ((f"{tag}_stream_replicates.txt", "k_stream_replicate_80000x6x3.txt"), (f"{tag}_stamps_C5.txt", "stamps_C5_k_stream.txt"),
                 (f"{tag}_stamps_C5rank.txt", "stamps_C5rank.txt"), (f"{tag}_stamps_C3.txt", "stamps_C3.txt"), (f"{tag}_ms_rates_C5.txt", "ms_rates_C5.txt")):
    cp(src, dst)
# (the traffic record names the COMMITTED directory its counter rows sit in, not the scratch tag; it is restamped only when this round's
#  counter passes exist -- a partial round must neither crash the collection nor point bench.py's replayed `traffic` at a directory
#  without the matching rows: ADVICE r03)
_tp = os.path.join(G, tag, "hbm_traffic_latest.json")
_have_rows = all(newest(os.path.join(f"{tag}/{d}", "**", "*counter_collection.csv")) for d in ("pmc_fetch", "pmc_write"))
if os.path.exists(_tp) and _have_rows:
    _t = json.load(open(_tp))
    _t["source"] = f"profiles/{name} (bash tools/round_end.sh {tag} -> tools/profile_all.sh; pmc_FETCH_SIZE_k_res.csv, pmc_WRITE_SIZE_k_res.csv)"
    json.dump(_t, open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json"), "w"), indent=1)
    json.dump(_t, open(os.path.join(D, "hbm_traffic.json"), "w"), indent=1)
else:
    print(f"missing: {_tp if not os.path.exists(_tp) else 'the FETCH / WRITE counter rows'} -- profiles/hbm_traffic_latest.json left as it was")
for kind in ("kernel_stats", "domain_stats"):
    f = newest(f"{tag}/trace/**/*{kind}.csv")
    if f:
        shutil.copy(f, os.path.join(D, kind + ".csv"))
for nm, d, k in (("pmc_FETCH_SIZE_k_res.csv", f"{tag}/pmc_fetch", "k_res"), ("pmc_WRITE_SIZE_k_res.csv", f"{tag}/pmc_write", "k_res"),
                 ("pmc_FETCH_SIZE_k_stream_C5.csv", f"{tag}_pmcf_C5", "k_stream"), ("pmc_WRITE_SIZE_k_stream_C5.csv", f"{tag}_pmcw_C5", "k_stream")):
    f = newest(os.path.join(d, "**", "*counter_collection.csv"))
    rows = [x for x in csv.DictReader(open(f)) if k in x.get("Kernel_Name", "")] if f else []
    if rows:      # (only the resident kernel's rows: the full files are MBs)
        w = csv.DictWriter(open(os.path.join(D, nm), "w"), fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
lines = []
for cfg in ("C3", "C4", "C5rank", "C5rank_plain", "C5"):
    f = newest(f"{tag}_trace_{cfg}/**/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(D, f"kernel_stats_{cfg}.csv"))
    log = os.path.join(G, f"{tag}_trace_{cfg}.log")
    if os.path.exists(log):
        lines += [ln for ln in open(log) if "HIP events" in ln]
open(os.path.join(D, "kernel_stats_other_configs.txt"), "w").writelines(lines)

b = json.load(open(os.path.join(D, "bench.json")))
r = b["roofline"]
print("bench", b["value"], "frac", r["frac"], "launch us", r["avg_launch_us"], "traffic", r["traffic"], "hbm_side_frac", r.get("hbm_side_frac"), "|", r["traffic_source"][:50])
print("20-step", json.load(open(os.path.join(D, "bench_20_steps_warmup_5.json")))["value"], "| 2 ranks on one GPU", json.load(open(os.path.join(D, "bench_2ranks_one_gpu_rehearsal.json")))["value"],
      "| C2/8", json.load(open(os.path.join(D, "bench_6250_barcodes.json")))["value"])
for ln in open(os.path.join(D, "configs.txt")):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(" ", d["config"][:52], d["us_per_step"], d["steps_per_s"], d["frac_hbm_peak"])
print("".join(lines))
