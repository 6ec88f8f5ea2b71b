"""Summarise the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of scripts/prof_pmc.sh into
profiles/<tag>_pmc_summary.json, applying the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md."""
import csv
import glob
import json
import sys

tag = sys.argv[1]
kernel_sub = sys.argv[2] if len(sys.argv) > 2 else "filter_"
N, T, bytes_per_step = 4096, 1000, 224
out = {"round": 1, "tag": tag,
       "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-saturation",
       "workload": "N=4096, T=1000, fp64, 4 outputs, native [T,comp,N] layout"}
for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = glob.glob(f"gpurun_out/prof_{tag}/pmc_{kind}/runc/*_counter_collection.csv")[0]
    rows = [r for r in csv.DictReader(open(f)) if kernel_sub in r["Kernel_Name"]]  # bench.py --no-saturation: every launch is the N x T batch
    out["kernel"] = rows[0]["Kernel_Name"][:120]
    vals = [float(r["Counter_Value"]) for r in rows][1:]
    out[name + "_KB_per_launch_raw"] = sum(vals) / len(vals)
    out["vgpr_count"], out["sgpr_count"] = int(rows[0]["VGPR_Count"]), int(rows[0]["SGPR_Count"])
    out["grid_size_lanes"] = int(rows[0]["Grid_Size"])
out["hbm_read_bytes_per_launch"] = out["FETCH_SIZE_KB_per_launch_raw"] * 1024 * 2
out["hbm_write_bytes_per_launch"] = out["WRITE_SIZE_KB_per_launch_raw"] * 1024
out["hbm_traffic_bytes_per_launch"] = out["hbm_read_bytes_per_launch"] + out["hbm_write_bytes_per_launch"]
out["algorithmic_bytes_per_launch"] = N * T * bytes_per_step
out["note"] = ("FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM (gfx950 tallies 128-B requests at 64 B): the doubled "
               "value equals the bytes of the inputs (t, y), each read exactly once; WRITE_SIZE taken as is.")
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
