#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/dbg_wg28.py 2>&1 | grep -v amdgpu.ids
