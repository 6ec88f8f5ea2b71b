#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== wg O3"; timeout 300 python scripts/gpu_time_w40dims.py 2>&1 | grep "^d="
echo "== wg O1"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_wgO1.so timeout 300 python scripts/gpu_time_w40dims.py 2>&1 | grep "^d="
echo "== wg tests under O1"; CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_wgO1.so timeout 600 python -m pytest tests/test_gpu_wg.py tests/test_gpu_soak.py -m gpu -q --timeout=300 2>&1 | tail -3
echo "== prof config5"; bash scripts/prof_r02.sh r03_c_config5 config5 2>&1 | tail -3
cat gpurun_out/prof_r03_c_config5/summary.txt | tail -40
