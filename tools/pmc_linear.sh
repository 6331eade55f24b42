#!/bin/bash
# rocprofv3 passes for k_linear_ws at [34816, 256] x [256 -> 1024], cold operands -> gpurun_out/lin/linear_pmc.json
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pl && mkdir -p gpurun_out/lin
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl/kt -- python3 tools/prof_linear.py > gpurun_out/lin/kt.log 2>&1
cp /tmp/pl/kt/*/*_kernel_stats.csv gpurun_out/lin/kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pl/p1 -- python3 tools/prof_linear.py > gpurun_out/lin/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pl/p2 -- python3 tools/prof_linear.py > gpurun_out/lin/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pl/p3 -- python3 tools/prof_linear.py > gpurun_out/lin/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pl/p4 -- python3 tools/prof_linear.py > gpurun_out/lin/p4.log 2>&1
for p in p1 p2 p3 p4; do cp /tmp/pl/$p/*/*_counter_collection.csv gpurun_out/lin/$p.csv; done
python3 - <<'PY'
import csv, collections, json
d = {}
for p in ("p1", "p2", "p3", "p4"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/lin/{p}.csv")):
        if "k_linear_ws" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        d[k] = sum(v) / len(v)
for r in csv.DictReader(open("gpurun_out/lin/kernel_stats.csv")):
    if "k_linear_ws" in r["Name"]:
        d["avg_ns"] = float(r["AverageNs"]); d["calls"] = int(r["Calls"])
d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
d["wait_inst_frac"] = d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"]
d["hbm_read_MB"] = 2 * 1024 * d["FETCH_SIZE"] / 1e6
d["hbm_write_MB"] = 1024 * d["WRITE_SIZE"] / 1e6
json.dump(d, open("gpurun_out/lin/linear_pmc.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
