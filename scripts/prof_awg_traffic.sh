#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / scratch-related counters of the workgroup reverse sweep in its two forms (gpurun: scripts/prof_awg_traffic.sh <tag>)
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/awgt_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for MODE in all drift; do
  i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SMEM"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/${MODE}_pmc$i -- python3 $GRAFT_REPO_ROOT/scripts/awg_traffic.py $MODE ${2:-256} ${3:-100} > $OUT/${MODE}_pmc$i.out 2>&1
  done
done
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv, glob, collections
for mode in ("all", "drift"):
    tot = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/awgt_$TAG/%s_pmc*/**/*counter_collection.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in tot.items():
        if "adjoint" not in k and "filter" not in k: continue
        print(mode, k, {c: (sum(v[1:]) / max(1, len(v) - 1)) for c, v in cs.items()})
PY
