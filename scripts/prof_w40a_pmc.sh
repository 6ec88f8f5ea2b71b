#!/bin/bash
# Counters of the wavefront-per-trajectory Lorenz-96 reverse sweep (gpurun: bash scripts/prof_w40a_pmc.sh <tag> [N] [T])
TAG=$1; OUT=$GRAFT_REPO_ROOT/gpurun_out/w40a_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- python3 $GRAFT_REPO_ROOT/scripts/time_awg.py ${2:-256} ${3:-100} 2 > $OUT/pmc$i.out 2>&1
done
cd $GRAFT_REPO_ROOT; python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/w40a_$TAG/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in tot.items():
    if "adjoint" not in k and "filter" not in k: continue
    print(k)
    for c, v in sorted(cs.items()): print("   %-28s %.4g" % (c, sum(v[1:]) / max(1, len(v) - 1)))
PY
