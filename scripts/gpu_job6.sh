#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== HEAD library (round-3 first commit)"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_head.so timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids | grep "tsit5\|euler" | grep "mlp5\|l96_6 "
echo "== current library"; timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids | grep "tsit5\|euler" | grep "mlp5\|l96_6 "
