#!/bin/bash
# Usage (GPU box): scripts/prof_lanes.sh  -- SQ counters of the headline kernel at 64 / 32 / 16 trajectories per wavefront
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_lanes
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${LANES_LIST:-64 32 16}; do
  export CDKF_LANES_PER_WAVE=$W
  i=0
  for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/w${W}_p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-saturation > $OUT/w${W}_p$i.json 2> $OUT/w${W}_p$i.err
  done
done
python3 - <<PY
import csv, glob, collections
for W in [int(w) for w in "${LANES_LIST:-64 32 16}".split()]:
    agg = collections.OrderedDict()
    for f in sorted(glob.glob("$OUT/w%d_p*/*/*_counter_collection.csv" % W)):
        for r in csv.DictReader(open(f)):
            if "filter_reg_kernel" in r["Kernel_Name"]:
                agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    print("W=%d" % W, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
