#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_parity.py -m gpu -q --timeout=600 -k "two_ranks or unsupported_raises or comm" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12
