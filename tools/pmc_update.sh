#!/bin/bash
# per-kernel duration + HBM bytes of the update's kernels (eager launches so that every dispatch is counted)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pu && mkdir -p gpurun_out/upd
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pu/kt -- python3 tools/prof_update_eager.py > gpurun_out/upd/kt.log 2>&1
cp /tmp/pu/kt/*/*_kernel_stats.csv gpurun_out/upd/kernel_stats.csv
# FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC counter slots): one pass each
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pu/p1 -- python3 tools/prof_update_eager.py > gpurun_out/upd/p1.log 2>&1
cp /tmp/pu/p1/*/*_counter_collection.csv gpurun_out/upd/p1.csv
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pu/p2 -- python3 tools/prof_update_eager.py > gpurun_out/upd/p2.log 2>&1
cp /tmp/pu/p2/*/*_counter_collection.csv gpurun_out/upd/p2.csv
python3 - <<'PY'
import csv, collections, json
dur = {}
for r in csv.DictReader(open("gpurun_out/upd/kernel_stats.csv")):
    dur[r["Name"]] = (float(r["AverageNs"]), int(r["Calls"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    for r in csv.DictReader(open(f"gpurun_out/upd/{p}.csv")):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for name, c in agg.items():
    if name not in dur or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    avg_ns, calls = dur[name]
    rd = 2 * 1024 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])   # KiB, x2 gfx950 correction for wide coalesced loads
    wr = 1024 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    if avg_ns * calls < 2e5:
        continue
    out[name[:110]] = {"calls": calls, "avg_us": round(avg_ns / 1e3, 2), "hbm_read_MB": round(rd / 1e6, 2),
                       "hbm_write_MB": round(wr / 1e6, 2), "GBps": round((rd + wr) / avg_ns, 1)}
out = dict(sorted(out.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["calls"]))
out["_note"] = ("rocprofv3: kernel-trace pass + separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over tools/prof_update_eager.py (eager "
                "launches of the PPO update at minibatch 2048); per-launch means; FETCH_SIZE in KiB x2 (gfx950 correction), "
                "WRITE_SIZE in KiB; traffic served by L2/MALL does not show here, so GB/s can exceed what HBM alone delivers")
json.dump(out, open("gpurun_out/upd/update_kernels_pmc.json", "w"), indent=1)
for k, v in list(out.items())[:28]:
    print(v, k[:70])
PY
