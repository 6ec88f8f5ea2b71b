#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== ckm on"; timeout 120 python scripts/dbg_w8.py 2>&1 | grep -v amdgpu.ids
echo "== ckm off"; CDKF_ADJ_MLP_CKPT=0 timeout 120 python scripts/dbg_w8.py 2>&1 | grep -v amdgpu.ids
echo "== tsit5"; timeout 120 python scripts/dbg_tsit5.py 2>&1 | grep -v amdgpu.ids
