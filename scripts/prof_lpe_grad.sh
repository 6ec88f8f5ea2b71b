#!/bin/bash
# Usage (GPU box): scripts/prof_lpe_grad.sh -- SQ counters of the gradient's reverse sweep on the lane grid
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lpe_grad
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time_grad.py bench > $OUT/p$i.log 2> $OUT/p$i.err
done
python3 - <<PY
import csv, glob, collections
for key in ("grad_lpe_l63_kernel<double", "filter_lpe_kernel<double"):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if key in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print(key, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/scripts/gpu_time_grad.py bench > $OUT/kt.log 2> $OUT/kt.err
cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats.csv
