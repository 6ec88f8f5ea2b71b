#!/bin/bash
# gpurun -- 'R5_TRACE_STEPS=3000 bash scripts/r5_gdb/trace.sh bad'
cd $GRAFT_REPO_ROOT
for w in bad:all good:none; do [ -f scripts/r5_gdb/d2grad_${w%%:*}.co ] || cp gpurun_out/r5_o3/d2grad_${w##*:}.co scripts/r5_gdb/d2grad_${w%%:*}.co; done
export CDKF_RTC_CACHE_DIR=/tmp/r5gdb_cache; mkdir -p $CDKF_RTC_CACHE_DIR; chmod 755 $CDKF_RTC_CACHE_DIR
export CDKF_RTC_OVERRIDE_CO=$GRAFT_REPO_ROOT/scripts/r5_gdb/d2grad_${1:-bad}.co CDKF_RTC_EXTRA_OPTS_ONLY="reg ukf=0 algo=2" CDKF_RTC_EXTRA_OPTS=""
export R5_TRACE_OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_trace_${1:-bad}.txt
timeout ${TMO:-1200} /opt/rocm/bin/rocgdb --batch -x scripts/r5_gdb/trace.py --args python3 scripts/r5_o3_probe.py case d2grad 2>&1 | grep -v "New Thread\|exited\]\|AMDGPU Wave" | tail -5
tail -3 $R5_TRACE_OUT
