#!/bin/bash
# Usage (GPU box): scripts/prof_lpe_grad_hbm.sh -- HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) of the gradient's two sweeps
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lpe_grad_hbm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$C -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time_grad.py bench all > $OUT/$C.log 2> $OUT/$C.err
done
python3 - <<PY
import csv, glob, collections
for key in ("grad_lpe_l63_kernel<double, 3, true, false", "grad_lpe_l63_kernel<double, 3, true, true", "filter_lpe_kernel<double"):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob("$OUT/*/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(key, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
