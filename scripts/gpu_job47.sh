#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/dbg_w8_f32.py 2>&1 | grep -v amdgpu.ids
for v in w8_1fb9633 w8_d9e7087; do CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_$v.so timeout 300 python scripts/dbg_w8_f32.py 2>&1 | grep -v amdgpu.ids; done
