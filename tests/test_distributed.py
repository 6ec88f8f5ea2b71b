"""The N > 1 path on CPU: world_size-2 gloo processes shard the trajectories, filter their block (through the
oracle, standing in for the GPU library) and all-reduce the log-likelihood sum."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from cd_dynamax_amd.distributed import shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"], os.path.join(os.environ["CDKF_ROOT"], "oracle")]
import numpy as np
import torch.distributed as dist
import cdkf_oracle as o
from cd_dynamax_amd import distributed as D

rank, local_rank, world = D.init_process_group("gloo")
assert world == 2 and dist.get_backend() == "gloo"
rng = np.random.default_rng(0)           # every rank builds the same global batch and takes its block
mdl = o.lorenz63_model(3)
N, T = 7, 15
t = o.irregular_times(rng, N, T, 0.1)
y = o.simulate(mdl, t, rng)
calls = []
def local(lo, hi):
    calls.append((lo, hi))
    return o.ekf_filter(mdl, t[lo:hi], y[lo:hi])["marginal_loglik"]
total = D.sharded_marginal_log_prob(local, N)
full = o.ekf_filter(mdl, t, y)["marginal_loglik"].sum()
assert calls == [D.shard_bounds(N, rank, world)], calls
assert abs(total - full) < 1e-9 * abs(full), (total, full)
assert abs(D.allreduce_sum(float(rank + 1)) - 3.0) < 1e-12
# value-and-gradient: 1 + n_theta sums in one collective
tot, g = D.sharded_loglik_and_grad(lambda lo, hi: o.ekf_loglik_grad(mdl, t[lo:hi], y[lo:hi]), N, 3)
ll_full, g_full = o.ekf_loglik_grad(mdl, t, y)
assert abs(tot - ll_full.sum()) < 1e-9 * abs(ll_full.sum())
assert np.allclose(g, g_full.sum(axis=0), rtol=1e-9, atol=1e-9), (g, g_full.sum(axis=0))
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("RANK_OK_%d\n" % rank); sys.stdout.flush()
'''


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 4096, 16384 + 3):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_two_process_gloo_loglik_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CDKF_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("RANK_OK_") == 2, out.stdout
