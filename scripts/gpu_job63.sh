#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 11 12; do timeout 1500 python scripts/gpu_fuzz_custom.py $s 10 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -12 | cut -c1-300; done
timeout 900 python -m pytest tests/test_custom_drift.py -q -m gpu -x --timeout=900 -k wide 2>&1 | tail -3
