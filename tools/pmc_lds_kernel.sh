#!/bin/bash
# LDS counters of ONE kernel of the update (name substring $1) over tools/prof_update_eager.py
KERNEL=${1:-k_dweight}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pul && mkdir -p gpurun_out/upd
timeout -k 5 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pul/p1 -- python3 tools/prof_update_eager.py > gpurun_out/upd/l_p1.log 2>&1
cp /tmp/pul/p1/*/*_counter_collection.csv gpurun_out/upd/l_p1.csv
KERNEL="$KERNEL" python3 - <<'PY'
import csv, collections, json, os
kern = os.environ["KERNEL"]
agg = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/upd/l_p1.csv")):
    if kern in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = {k: sum(v) / len(v) for k, v in agg.items()}
print(json.dumps(d, indent=1))
PY
