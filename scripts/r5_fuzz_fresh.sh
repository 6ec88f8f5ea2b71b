#!/bin/bash
# Round 5: the random-problem fuzzers (scripts/gpu_fuzz_*.py) with seeds never used before, under the SHIPPED run-time build policy
# (launch_custom.hip: rtc_policy) and an empty code-object cache -- what the fixed-seed slice of tests/test_gpu_fuzz.py cannot say about
# problems nobody has looked at.  gpurun -- 'bash scripts/r5_fuzz_fresh.sh > gpurun_out/r5_fuzz_fresh.txt 2>&1'
cd $GRAFT_REPO_ROOT
export CDKF_RTC_CACHE_DIR=/tmp/r5_fuzz_cache; rm -rf $CDKF_RTC_CACHE_DIR; mkdir -p $CDKF_RTC_CACHE_DIR; chmod 755 $CDKF_RTC_CACHE_DIR
export PYTHONUNBUFFERED=1
run() { echo "=== $*"; timeout ${TMO:-1500} "$@" 2>&1 | grep -E "MISMATCH|worst|Traceback|Error|status" | tail -6; echo "rc=$?"; }
S=${R5_SEED:-905000}
run python scripts/gpu_fuzz_custom.py $((S+1)) 8
CDKF_FUZZ_D=4 run python scripts/gpu_fuzz_custom.py $((S+2)) 5
CDKF_FUZZ_D=6 run python scripts/gpu_fuzz_custom.py $((S+3)) 5
CDKF_FUZZ_D=2 run python scripts/gpu_fuzz_custom.py $((S+4)) 4
for f in filters grads batches solvers misc r03; do run python scripts/gpu_fuzz_$f.py $((S+10)) 8; done
ls $CDKF_RTC_CACHE_DIR | wc -l
grep -c "O1 (vgpr spills at -O3" $CDKF_RTC_CACHE_DIR/MANIFEST
grep "vgpr spills at -O3" $CDKF_RTC_CACHE_DIR/MANIFEST | sed 's/.*bytes //' | sort | uniq -c | sort -rn | head -40
