#!/bin/bash
# issue / wait / MFMA counters of ONE kernel of the update (name substring $1) over tools/prof_update_eager.py -> gpurun_out/upd/<tag>_pmc.json
KERNEL=${1:-k_attn_bwd17_mfma}
TAG=${2:-$KERNEL}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/puk && mkdir -p gpurun_out/upd
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/puk/kt -- python3 tools/prof_update_eager.py > gpurun_out/upd/k_kt.log 2>&1
cp /tmp/puk/kt/*/*_kernel_stats.csv gpurun_out/upd/k_kernel_stats.csv
timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/puk/p1 -- python3 tools/prof_update_eager.py > gpurun_out/upd/k_p1.log 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d /tmp/puk/p2 -- python3 tools/prof_update_eager.py > gpurun_out/upd/k_p2.log 2>&1
for p in p1 p2; do cp /tmp/puk/$p/*/*_counter_collection.csv gpurun_out/upd/k_$p.csv || true; done
KERNEL="$KERNEL" TAG="$TAG" python3 - <<'PY'
import csv, collections, json, os
kern, tag = os.environ["KERNEL"], os.environ["TAG"]
d = {}
for p in ("p1", "p2"):
    agg = collections.defaultdict(list)
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/upd/k_{p}.csv")))
    except OSError:
        continue
    for r in rows:
        if kern in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        d[k] = sum(v) / len(v)
for r in csv.DictReader(open("gpurun_out/upd/k_kernel_stats.csv")):
    if kern in r["Name"]:
        d["avg_ns"] = float(r["AverageNs"]); d["calls"] = int(r["Calls"])
if "SQ_WAVE_CYCLES" in d:
    d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
    d["wait_inst_frac"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
if "SQ_WAVES" in d:
    d["valu_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
    d["valu_active_cycles_per_simd"] = 4 * d["SQ_ACTIVE_INST_VALU"] / 1024
    d["gpu_cycles"] = d["GRBM_GUI_ACTIVE"] / 8
json.dump(d, open(f"gpurun_out/upd/{tag}_pmc.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
