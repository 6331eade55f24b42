#!/bin/bash
# rocprofv3 passes for the rollout encoder pair (k_encoder_main<HEAD>, k_encoder_tail) at 65536 boards: kernel trace + two PMC passes
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pe && mkdir -p gpurun_out/enc
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pe/kt -- python3 tools/prof_fused.py 65536 > gpurun_out/enc/kt.log 2>&1
cp /tmp/pe/kt/*/*_kernel_stats.csv gpurun_out/enc/kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pe/p1 -- python3 tools/prof_fused.py 65536 > gpurun_out/enc/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pe/p2 -- python3 tools/prof_fused.py 65536 > gpurun_out/enc/p2.log 2>&1
for p in p1 p2; do cp /tmp/pe/$p/*/*_counter_collection.csv gpurun_out/enc/$p.csv; done
python3 - <<'PY'
import csv, collections, json
out = {}
for pats, tag in ((("mainILi1E", "k_encoder_main<1>"), "head"), (("k_encoder_tail",), "tail"), (("mainILi0E", "k_encoder_main<0>"), "single")):
    d = {}
    for p in ("p1", "p2"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f"gpurun_out/enc/{p}.csv")):
            if any(pat in r["Kernel_Name"] for pat in pats):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            d[k] = sum(v) / len(v)
    if d:
        d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
        d["mfma_per_wave"] = d["SQ_INSTS_MFMA"] / d["SQ_WAVES"]
        d["valu_instr_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
        d["wait_any_frac"] = d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"]
        out[tag] = d
for r in csv.DictReader(open("gpurun_out/enc/kernel_stats.csv")):
    if "k_encoder" in r["Name"]:
        out.setdefault("kernel_trace", {})[r["Name"][:60]] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
out["_note"] = ("rocprofv3 on tools/prof_fused.py 65536 (4 layers): k_encoder_main<1> = HEAD (layers 0-2 + K/V of layer 3), "
                "k_encoder_tail = CLS-only rest of layer 3 (128 boards per workgroup), k_encoder_main<0> = single-kernel form; PMC in two separate passes; "
                "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)")
json.dump(out, open("gpurun_out/enc/encoder_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
