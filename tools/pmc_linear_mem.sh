#!/bin/bash
# memory-side stall counters of k_linear_ws at [34816, 256] x [256 -> 1024], cold operands -> gpurun_out/lin/linear_mem_pmc.json
# (at most 4 counters of one block per pass: more aborts rocprofv3 with 'exceeds the capabilities of the hardware')
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/plm && mkdir -p gpurun_out/lin
timeout -k 5 150 rocprofv3 --pmc TCC_CYCLE_sum TCC_BUSY_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum --output-format csv -d /tmp/plm/m1 -- python3 tools/prof_linear.py > gpurun_out/lin/m1.log 2>&1
timeout -k 5 150 rocprofv3 --pmc TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_WRITE_sum --output-format csv -d /tmp/plm/m2 -- python3 tools/prof_linear.py > gpurun_out/lin/m2.log 2>&1
timeout -k 5 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d /tmp/plm/m3 -- python3 tools/prof_linear.py > gpurun_out/lin/m3.log 2>&1
for p in m1 m2 m3; do cp /tmp/plm/$p/*/*_counter_collection.csv gpurun_out/lin/$p.csv || true; done
python3 - <<'PY'
import csv, collections, json
d = {}
for p in ("m1", "m2", "m3"):
    agg = collections.defaultdict(list)
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/lin/{p}.csv")))
    except OSError:
        continue
    for r in rows:
        if "k_linear_ws" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        d[k] = sum(v) / len(v)
json.dump(d, open("gpurun_out/lin/linear_mem_pmc.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PY
