#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -k "linear_drift_on_the_lane_grid or mlp_loglik_gradient_reverse_sweep or unscented_loglik_gradient or lower_fidelity or literal_sigma or reverse_sweep_on_the_lane_grid or other_runge_kutta or adaptive_steps" > gpurun_out/j10_pytest.log 2>&1; echo "rc $?"; tail -25 gpurun_out/j10_pytest.log | cut -c1-250
