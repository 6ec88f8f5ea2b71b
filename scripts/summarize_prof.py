"""Summarise rocprofv3 output directories into tracked files under profiles/.

    python scripts/summarize_prof.py <tag> <dir> [--alg-bytes N] [--workload TEXT] [--command TEXT] [kernel substring ...]

<dir> holds any number of rocprofv3 -d outputs (kernel-trace --stats and/or --pmc passes, CSV format).  Writes
profiles/<tag>_kernel_stats.csv (the kernel rows of every *_kernel_stats.csv found, device kernels only) and
profiles/<tag>_counters.json: per kernel, the mean per launch of every counter collected (first launch of each kernel dropped),
plus -- when FETCH_SIZE and WRITE_SIZE are both present -- the HBM bytes per launch with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md (section HBM).  Only kernels whose name contains one of the substrings (default: "cdkf::") are kept.
--alg-bytes stamps the algorithmic bytes per launch of the profiled workload's sweep into the file: bench.py cites a file's traffic
only when that number AND the kernel name match the launch it has just made."""
import collections, csv, glob, json, os, sys

argv = sys.argv[1:]
stamp = {}
for flag, key, conv in (("--alg-bytes", "algorithmic_bytes_per_launch", int), ("--workload", "workload", str), ("--command", "command", str)):
    if flag in argv:
        k = argv.index(flag)
        stamp[key] = conv(argv[k + 1])
        del argv[k:k + 2]
tag, root = argv[0], argv[1]
subs = argv[2:] or ["cdkf::"]
keep = lambda name: any(s in name for s in subs)
os.makedirs("profiles", exist_ok=True)
rows_out, header = [], None
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    rd = csv.reader(open(f))
    h = next(rd)
    header = header or h
    for r in rd:
        if keep(r[0]):
            rows_out.append(r)
if rows_out:
    with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(header)
        w.writerows(rows_out)
agg = collections.OrderedDict()
meta = {}
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if keep(r["Kernel_Name"]):
            per[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta.setdefault(r["Kernel_Name"], {"vgpr_count": int(r["VGPR_Count"]), "sgpr_count": int(r["SGPR_Count"]),
                                               "lds_block_size": int(r.get("LDS_Block_Size", 0) or 0), "grid_size": int(r["Grid_Size"]),
                                               "workgroup_size": int(r["Workgroup_Size"])})
    for (k, c), v in per.items():
        vals = v[1:] if len(v) > 1 else v
        agg.setdefault(k, collections.OrderedDict())[c] = {"mean_per_launch": sum(vals) / len(vals), "launches": len(vals)}
out = {"tag": tag, **stamp, "kernels": []}
for k, cs in agg.items():
    rec = {"kernel": k[:200], **meta.get(k, {}), "counters": cs}
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        rd_b = cs["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
        wr_b = cs["WRITE_SIZE"]["mean_per_launch"] * 1024
        rec["hbm_read_bytes_per_launch"], rec["hbm_write_bytes_per_launch"] = rd_b, wr_b
        rec["hbm_traffic_bytes_per_launch"] = rd_b + wr_b
        rec["note"] = "FETCH_SIZE (KB) x 1024 x 2 per MI355X_MICROARCH.md section HBM; WRITE_SIZE (KB) x 1024"
    out["kernels"].append(rec)
if out["kernels"]:
    json.dump(out, open(f"profiles/{tag}_counters.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
