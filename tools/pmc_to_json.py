import csv, collections, json, sys
tag, out_path = sys.argv[1], sys.argv[2]
out = {}
for p in ("p1", "p2", "p3"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f"gpurun_out/{tag}/{p}.csv")):
        if "k_step" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out[k] = sum(v) / len(v)
for r in csv.DictReader(open(f"gpurun_out/{tag}/kernel_stats.csv")):
    if "k_step" in r["Name"]:
        out["kernel_avg_ns_kernel_trace"] = float(r["AverageNs"]); out["kernel_calls"] = int(r["Calls"])
fetch = out["FETCH_SIZE"] * 1024 * 2; write = out["WRITE_SIZE"] * 1024
out.update(hbm_read_bytes_corrected=fetch, hbm_write_bytes=write, hbm_bytes_per_launch=fetch + write,
           hbm_bytes_per_env_step=(fetch + write) / (1 << 24),
           valu_instr_per_wave=out["SQ_INSTS_VALU"] / out["SQ_WAVES"],
           cycles_per_valu_wave_instr=(out["GRBM_GUI_ACTIVE"] / 8) / (out["SQ_INSTS_VALU"] / 1024),
           _note="rocprofv3, k_step<1> (g2048_step) at 2^24 boards/launch; PMC in separate passes (mean of 3 launches); "
                 "FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE x2 = gfx950 correction for wide coalesced loads "
                 "(MI355X_MICROARCH.md, HBM); GRBM_GUI_ACTIVE is summed over 8 XCDs; 1024 SIMDs")
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
