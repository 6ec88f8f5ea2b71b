#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== current library (new WgArgs fields at the end)"; timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids | grep "tsit5\|euler" | grep "mlp5\|l96_6 "
