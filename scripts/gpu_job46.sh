#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/dbg_wg_mlp.py 2>&1 | grep -v amdgpu.ids
