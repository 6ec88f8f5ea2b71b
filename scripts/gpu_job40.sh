#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== base"; timeout 300 python scripts/gpu_time_w40dims.py 2>&1 | grep "^d=" | cut -c1-110
echo "== ldp odd"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_ldpodd.so timeout 300 python scripts/gpu_time_w40dims.py 2>&1 | grep "^d=" | cut -c1-110
echo "== tests under variant"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_ldpodd.so timeout 600 python -m pytest tests/test_gpu_wg.py -m gpu -q --timeout=300 -k "lorenz96 or c4" 2>&1 | tail -2
