#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/dbg_w40_d32.py 2>&1 | grep -v amdgpu.ids | grep -v step0 | cut -c1-60,250-330
for s in 21 22 23 24; do timeout 900 python scripts/gpu_fuzz_r03.py $s 25 2>&1 | grep -v amdgpu.ids | tail -4; done
timeout 900 python -m pytest tests/test_gpu_wg.py -m gpu -q --timeout=600 -k "lorenz96 or c4" 2>&1 | tail -2
