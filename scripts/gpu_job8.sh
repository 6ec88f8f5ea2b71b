#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in v1 v2; do
  if [ -f cd_dynamax_amd/lib/libcdkf_hip_$v.so ]; then
    echo "== variant $v"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_$v.so timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids | grep "tsit5" | grep "mlp5\|l96_6 "
  fi
done
