#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== default"; timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids
echo "== CDKF_NO_WAVE8=1"; CDKF_NO_WAVE8=1 timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids | grep "mlp5\|l96_6 "
