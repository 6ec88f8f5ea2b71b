#!/bin/bash
# Round 5: rocgdb on the WRONG build of the d = 2 forward-sensitivity kernel (scripts/r5_mir_delta.py's hybrid `all`): what do the
# registers of the final `if (live)` compare hold?  gpurun -- 'bash scripts/r5_gdb/run.sh'
cd $GRAFT_REPO_ROOT
# the two code objects are build products (git-ignored): `python scripts/r5_mir_delta.py d2grad kinds` writes them to gpurun_out/r5_o3/
for w in bad:all good:none; do [ -f scripts/r5_gdb/d2grad_${w%%:*}.co ] || cp gpurun_out/r5_o3/d2grad_${w##*:}.co scripts/r5_gdb/d2grad_${w%%:*}.co; done
export CDKF_RTC_CACHE_DIR=/tmp/r5gdb_cache; mkdir -p $CDKF_RTC_CACHE_DIR; chmod 755 $CDKF_RTC_CACHE_DIR
export CDKF_RTC_OVERRIDE_CO=$GRAFT_REPO_ROOT/scripts/r5_gdb/d2grad_${1:-bad}.co CDKF_RTC_EXTRA_OPTS_ONLY="reg ukf=0 algo=2" CDKF_RTC_EXTRA_OPTS=""
timeout 900 /opt/rocm/bin/rocgdb --batch -x scripts/r5_gdb/${2:-final_compare}.gdb --args python3 scripts/r5_o3_probe.py case d2grad 2>&1 | grep -v "New Thread\|exited\]\|AMDGPU Wave"
