#!/bin/bash
# Usage (GPU box): scripts/prof_lpe.sh -- SQ counters of the headline sweep (whichever kernel the library picks)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lpe
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-saturation > $OUT/p$i.json 2> $OUT/p$i.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "filter_" in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            name = r["Kernel_Name"][:60]
print(name, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
