#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_custom_drift.py -q -m gpu -x --timeout=900 -k "wide or forty" > gpurun_out/j58.log 2>&1
grep -v amdgpu.ids gpurun_out/j58.log | grep -n "Fatal\|Error\|error\|File \"/root\|File \"/tmp\|tests/\|cd_dynamax" | head -40 | cut -c1-300
