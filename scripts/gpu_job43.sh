#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/dbg_w40_d32.py 2>&1 | grep -v amdgpu.ids
