#!/bin/bash
# Usage (GPU box, via gpurun): scripts/prof_r02.sh <tag> <target> [target args]
#   target: bench (the headline line) | any name scripts/run_config.py takes (config3, config4, config5, config2_with_smoother, grad)
# One --kernel-trace --stats run, then separate --pmc passes (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2; 8 SQ
# slots per pass), the program itself directly after `--`.  Output under gpurun_out/prof_<tag>/; summarise with scripts/summarize_prof.py.
set -u
TAG=$1; TARGET=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$TARGET" = bench ]; then
  CMD="python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-saturation"
else
  CMD="python3 $GRAFT_REPO_ROOT/scripts/run_config.py $TARGET 1"
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.out 2> $OUT/trace.err
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.out 2> $OUT/pmc$i.err
done
EXTRA=""
if [ "$TARGET" = bench ]; then EXTRA="--alg-bytes 917504000"; fi
cd $GRAFT_REPO_ROOT && python3 scripts/summarize_prof.py $TAG $OUT $EXTRA --command "scripts/prof_r02.sh $TAG $TARGET: $CMD" > $OUT/summary.txt 2>&1
tail -5 $OUT/trace.out
