#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== mb_mfma16"; timeout 300 ./scripts/mb/mb_mfma16 2>&1 | tee gpurun_out/j16_mb_mfma16.log
echo "== ept8 O1"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_wgO1.so timeout 300 python scripts/dbg_wg_ept8.py 2>&1 | grep -v amdgpu.ids | grep "^4[68] 4"
