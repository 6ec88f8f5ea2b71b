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
lo_, hi_ = D.shard_bounds(N, rank, world)
assert calls == ([(lo_, hi_)] if hi_ > lo_ else []), calls   # (world 8, N 7: the last rank's block is empty and is not evaluated)
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


FIT_WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"], os.path.join(os.environ["CDKF_ROOT"], "oracle"), os.path.join(os.environ["CDKF_ROOT"], "tests")]
import numpy as np
import torch.distributed as dist
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import distributed as D, fit
from test_fit import _l63_problem

rank, local_rank, world = D.init_process_group("gloo")
assert world == 2

class OracleBatch:
    """Stands in for fit._ResidentBatch (the GPU sweep) on CPU: same interface, value and gradient from the oracle."""
    def __init__(self, y, t, t_shared, n_theta, n_model, dtype):
        self.y, self.t, self.B = np.asarray(y, np.float64), np.asarray(t, np.float64), y.shape[0]
    def value_and_grad(self, mdl, opts, suffix):
        m = o.lorenz63_model(1)
        cur = o.Model(o.Lorenz63Drift(*mdl.theta), m.L, m.Qc, m.H, m.bias, m.R, m.m0, 100 * np.eye(3))
        ll, g = o.ekf_loglik_grad(cur, self.t, self.y)
        return float(ll.sum()), g.sum(0), np.zeros(0)
    def free(self):
        pass
fit._ResidentBatch = OracleBatch

model, params, props = _l63_problem(m=1)
rng = np.random.default_rng(5)              # every rank builds the same global data and takes its (unequal) block
mdl = o.lorenz63_model(1)
mdl = o.Model(mdl.drift, mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, 100 * np.eye(3))
N, T, lr = 7, 12, 0.05
t = o.irregular_times(rng, N, T, 0.05)
y = o.simulate(mdl, t, rng)
lo, hi = D.shard_bounds(N, rank, world)      # 4 and 3 sequences
calls = []
def allreduce(x):
    calls.append(len(x))
    return D.allreduce_sum_array(x)

def expected(batches):
    """The single-process loop of fit_sgd over the given global minibatches (index arrays), written out with the oracle."""
    th, losses = mdl.drift.theta().copy(), []
    for idx in batches:
        cur = o.Model(o.Lorenz63Drift(*th), mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, mdl.P0)
        ll, g = o.ekf_loglik_grad(cur, t[idx], y[idx])
        scale = N / len(idx)
        losses.append(-(ll.sum() * scale) / y.size)
        th = th - lr * (-(g.sum(0) * scale) / y.size)
    return th, float(np.mean(losses))

# (a) full batch: identical to the single-process fit of all seven sequences
new, losses = model.fit_sgd(params, props, y[lo:hi], t[lo:hi, :, None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=N,
                            num_epochs=1, allreduce=allreduce)
th, loss = expected([np.arange(N)])
got = np.array([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta])
assert np.allclose(got, th, rtol=1e-10), (got, th)
assert abs(losses[0] - loss) < 1e-10 * abs(loss), (losses, loss)
assert calls == [2, 2 + 3], calls            # one set-up reduction, one per step

# (b) minibatches of 3 out of 7 with blocks of 4 and 3: three steps on BOTH ranks (no rank runs out of collectives), step b made
# of every rank's b-th piece
calls.clear()
new, losses = model.fit_sgd(params, props, y[lo:hi], t[lo:hi, :, None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=3,
                            num_epochs=1, allreduce=allreduce)
assert calls == [2, 5, 5, 5], calls
pieces = [np.array_split(np.arange(*D.shard_bounds(N, r, world)), 3) for r in range(world)]
th, loss = expected([np.concatenate([pieces[0][b], pieces[1][b]]) for b in range(3)])
got = np.array([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta])
assert np.allclose(got, th, rtol=1e-10), (got, th)
assert abs(losses[0] - loss) < 1e-10 * abs(loss), (losses, loss)

# (c) more steps than a rank has sequences: its empty pieces contribute zeros, globally empty steps are skipped
new, losses = model.fit_sgd(params, props, y[lo:hi], t[lo:hi, :, None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=1,
                            num_epochs=1, allreduce=allreduce)
pieces = [np.array_split(np.arange(*D.shard_bounds(N, r, world)), 7) for r in range(world)]
steps = [np.concatenate([pieces[0][b], pieces[1][b]]) for b in range(7)]
assert [len(b) for b in steps] == [2, 2, 2, 1, 0, 0, 0]          # steps that no rank has a sequence for are skipped everywhere
th, loss = expected([b for b in steps if len(b)])
got = np.array([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta])
assert np.allclose(got, th, rtol=1e-10), (got, th)
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("RANK_OK_%d\n" % rank); sys.stdout.flush()
'''


def test_two_process_gloo_fit_sgd_matches_single_process(tmp_path):
    """fit_sgd(allreduce=...) on two ranks with unequal blocks: the global loss scaling (N_total / B_global, emissions.size of
    the whole data set), the same number of collectives on every rank, empty pieces (round-1 advisor finding)."""
    script = tmp_path / "fit_worker.py"
    script.write_text(FIT_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CDKF_ROOT=ROOT, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("RANK_OK_") == 2, out.stdout


COMM_WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"], os.path.join(os.environ["CDKF_ROOT"], "oracle")]
import numpy as np
import cdkf_oracle as o
from cd_dynamax_amd import distributed as D

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comm = D.Comm(rank, world, "127.0.0.1", port, device=None)     # host-only: the library's rendezvous, no RCCL, no torch
assert "torch" not in sys.modules
rng = np.random.default_rng(0)
mdl = o.lorenz63_model(3)
N, T = 7, 15
t = o.irregular_times(rng, N, T, 0.1)
y = o.simulate(mdl, t, rng)
full = o.ekf_filter(mdl, t, y)["marginal_loglik"]
# shard -> local log-likelihoods -> the library's all-reduce
calls = []
def local(lo, hi):
    calls.append((lo, hi))
    return o.ekf_filter(mdl, t[lo:hi], y[lo:hi])["marginal_loglik"]
total = D.sharded_marginal_log_prob(local, N, comm=comm)
lo_, hi_ = D.shard_bounds(N, rank, world)
assert calls == ([(lo_, hi_)] if hi_ > lo_ else []), calls   # (world 8, N 7: the last rank's block is empty and is not evaluated)
# rank 0 adds in rank order: the same bits on every rank, equal to the sequential sum of the block sums
blocks = [full[slice(*D.shard_bounds(N, r, world))].sum() for r in range(world)]
seq = blocks[0]
for b in blocks[1:]:
    seq = seq + b
assert total == seq, (total, seq)
assert np.array_equal(comm.allreduce_sum_host(np.arange(5.0) * (rank + 1)), np.arange(5.0) * sum(range(1, world + 1)))
assert np.array_equal(comm.allreduce_max_host([float(rank), -float(rank)]), [world - 1.0, 0.0])
tot, g = D.sharded_loglik_and_grad(lambda lo, hi: o.ekf_loglik_grad(mdl, t[lo:hi], y[lo:hi]), N, 3, comm=comm)
ll_full, g_full = o.ekf_loglik_grad(mdl, t, y)
assert abs(tot - ll_full.sum()) < 1e-12 * abs(ll_full.sum())
assert np.allclose(g, g_full.sum(axis=0), rtol=1e-12, atol=1e-12)
try:
    comm.allreduce_sum_dev(None, 1)
    raise SystemExit("a host-only communicator must refuse the device collective")
except RuntimeError:
    pass
comm.barrier()
comm.close()
sys.stdout.write("RANK_OK_%d\n" % rank); sys.stdout.flush()
'''


@pytest.mark.parametrize("world", [2, 3, 8])
def test_library_rendezvous_allreduce_across_processes(tmp_path, world):
    """The N > 1 composition through the library's OWN collective entry points (cdkf_rdv_*; the RCCL leg needs GPUs): plain
    processes, no torch, shard -> local sums -> all-reduce."""
    script = tmp_path / "comm_worker.py"
    script.write_text(COMM_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CDKF_ROOT=ROOT, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert sum(so.count("RANK_OK_") for so, _ in outs) == world


FIT_COMM_WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"], os.path.join(os.environ["CDKF_ROOT"], "oracle"), os.path.join(os.environ["CDKF_ROOT"], "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import distributed as D, fit
from test_fit import _l63_problem

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comm = D.Comm(rank, world, "127.0.0.1", port, device=None)   # host-only: fit_sgd(comm=...) then sums over the library's rendezvous
assert "torch" not in sys.modules

class OracleBatch:
    """Stands in for fit._ResidentBatch (the GPU sweep) on CPU: same interface, value and gradient from the oracle."""
    def __init__(self, y, t, t_shared, n_theta, n_model, dtype):
        self.y, self.t, self.B = np.asarray(y, np.float64), np.asarray(t, np.float64), y.shape[0]
    def value_and_grad(self, mdl, opts, suffix):
        m = o.lorenz63_model(1)
        cur = o.Model(o.Lorenz63Drift(*mdl.theta), m.L, m.Qc, m.H, m.bias, m.R, m.m0, 100 * np.eye(3))
        ll, g = o.ekf_loglik_grad(cur, self.t, self.y)
        return float(ll.sum()), g.sum(0), np.zeros(0)
    def free(self):
        pass
fit._ResidentBatch = OracleBatch

model, params, props = _l63_problem(m=1)
rng = np.random.default_rng(5)
mdl = o.lorenz63_model(1)
mdl = o.Model(mdl.drift, mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, 100 * np.eye(3))
N, T, lr = 7, 12, 0.05
t = o.irregular_times(rng, N, T, 0.05)
y = o.simulate(mdl, t, rng)
lo, hi = D.shard_bounds(N, rank, world)

def expected(batches):
    th, losses = mdl.drift.theta().copy(), []
    for idx in batches:
        cur = o.Model(o.Lorenz63Drift(*th), mdl.L, mdl.Qc, mdl.H, mdl.bias, mdl.R, mdl.m0, mdl.P0)
        ll, g = o.ekf_loglik_grad(cur, t[idx], y[idx])
        scale = N / len(idx)
        losses.append(-(ll.sum() * scale) / y.size)
        th = th - lr * (-(g.sum(0) * scale) / y.size)
    return th, float(np.mean(losses))

for bs in (N, 3, 1):   # full batch; minibatches over unequal blocks; more steps than a rank has sequences (empty pieces)
    nb = -(-N // bs)
    new, losses = model.fit_sgd(params, props, y[lo:hi], t[lo:hi, :, None], cd.EKFHyperParams(), optimizer=fit.SGD(lr), batch_size=bs,
                                num_epochs=1, comm=comm)
    pieces = [np.array_split(np.arange(*D.shard_bounds(N, r, world)), nb) for r in range(world)]
    steps = [np.concatenate([pieces[r][b] for r in range(world)]) for b in range(nb)]
    th, loss = expected([b for b in steps if len(b)])
    got = np.array([new.dynamics.drift.sigma, new.dynamics.drift.rho, new.dynamics.drift.beta])
    assert np.allclose(got, th, rtol=1e-10), (bs, got, th)
    assert abs(losses[0] - loss) < 1e-10 * abs(loss), (bs, losses, loss)
try:
    model.fit_sgd(params, props, y[lo:hi], t[lo:hi, :, None], cd.EKFHyperParams(), comm=comm, allreduce=lambda x: x)
    raise SystemExit("comm= and allreduce= together must be refused")
except ValueError:
    pass
comm.barrier()
comm.close()
sys.stdout.write("RANK_OK_%d\\n" % rank); sys.stdout.flush()
'''


@pytest.mark.parametrize("world", [2, 3, 8])
def test_fit_sgd_through_the_library_communicator(tmp_path, world):
    """fit_sgd(comm=Comm(...)): the data-parallel SGD step through the library's own communicator object -- host-only here (TCP
    rendezvous; the RCCL leg of the same call is tests/test_gpu_comm.py::test_fit_sgd_reduces_on_the_device_through_rccl) --
    with unequal blocks, minibatches and empty pieces; plain processes, no torch."""
    script = tmp_path / "fit_comm_worker.py"
    script.write_text(FIT_COMM_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CDKF_ROOT=ROOT, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert sum(so.count("RANK_OK_") for so, _ in outs) == world


def test_rendezvous_argument_and_timeout_errors():
    import ctypes as C
    from cd_dynamax_amd import _ffi
    L = _ffi.lib()
    h = C.c_void_p()
    assert L.cdkf_rdv_create(C.byref(h), b"127.0.0.1", 0, 0, 2, 100) == _ffi.CDKF_EINVAL
    assert L.cdkf_rdv_create(C.byref(h), b"127.0.0.1", 1234, 2, 2, 100) == _ffi.CDKF_EINVAL
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    # nobody listens: rank 1 gives up after the timeout with a message, no hang
    rc = L.cdkf_rdv_create(C.byref(h), b"127.0.0.1", port, 1, 2, 300)
    assert rc != 0 and b"cannot reach rank 0" in L.cdkf_last_error()
    # rank 0 alone: the others never arrive
    rc = L.cdkf_rdv_create(C.byref(h), b"127.0.0.1", port, 0, 2, 300)
    assert rc != 0 and b"ranks arrived" in L.cdkf_last_error()
    assert L.cdkf_ll_allreduce(None, None, 1, None) == _ffi.CDKF_EINVAL


def test_rendezvous_turns_strangers_and_duplicates_away():
    """Rank 0 answers every hello: a connection that does not carry this job's nonce (a stray client, another job whose store sits on
    the same port) and a second process claiming a rank that is taken are NACKed -- they fail at once with a message instead of
    taking a slot or waiting out the I/O timeout -- and the real ranks still meet (round-2 advisor finding)."""
    import ctypes as C
    import struct
    import threading
    import time
    from cd_dynamax_amd import _ffi
    L = _ffi.lib()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 3
    res = {}

    def join(name, rank, timeout_ms=8000):
        h = C.c_void_p()
        rc = L.cdkf_rdv_create(C.byref(h), b"127.0.0.1", port, rank, world, timeout_ms)
        res[name] = (rc, L.cdkf_last_error().decode() if rc else "", h)

    t0 = threading.Thread(target=join, args=("r0", 0))
    t0.start()
    time.sleep(0.3)
    # a stranger: right size, wrong nonce -> NACK (0), connection closed, no slot taken
    with socket.create_connection(("127.0.0.1", port), timeout=5) as c:
        c.sendall(struct.pack("<iiiiq", 0x43444B52, 1, world, 0, 12345))
        assert struct.unpack("<i", c.recv(4))[0] == 0
    # a client that says nothing: dropped after two seconds at most, the accept loop goes on
    idle = socket.create_connection(("127.0.0.1", port), timeout=5)
    t1 = threading.Thread(target=join, args=("r1", 1))
    t1.start()
    t1.join(timeout=10)
    assert res["r1"][0] == 0, res["r1"]
    # a second rank 1: refused, with a message
    join("dup", 1, 3000)
    assert res["dup"][0] != 0 and "refused" in res["dup"][1], res["dup"]
    t2 = threading.Thread(target=join, args=("r2", 2))
    t2.start()
    t2.join(timeout=10)
    t0.join(timeout=10)
    idle.close()
    assert res["r0"][0] == 0 and res["r2"][0] == 0, res
    for k in ("r0", "r1", "r2"):
        L.cdkf_rdv_destroy(res[k][2])


FALLBACK_WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"]]
import numpy as np
from cd_dynamax_amd import distributed as D

rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
comm = D.Comm(rank, world, "127.0.0.1", port, device=rank)   # asks for RCCL on a box that has no GPU
assert not comm._comm and comm.rccl_error, "a communicator cannot exist here"
s = comm.allreduce_sum_host([rank + 1.0, 10.0 * (rank + 1)])
assert np.allclose(s, [world * (world + 1) / 2, 10.0 * world * (world + 1) / 2]), s
comm.barrier()
comm.close()
sys.stdout.write("RANK_OK_%d %s\n" % (rank, comm.rccl_error.replace("\n", " ")[:80])); sys.stdout.flush()
'''


@pytest.mark.parametrize("world", [2, 3, 8])
def test_comm_without_rccl_falls_back_on_every_rank(tmp_path, world):
    """Comm(device=...) where RCCL cannot be joined (here: no GPU): no rank is left waiting inside the collective set-up -- every step
    of it is taken by all ranks and followed by an agreement over the rendezvous -- and all of them end in the host fallback with the
    reason in ``rccl_error`` (bench.py --gpus N then reports the host path instead of dying)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: RCCL would join")
    script = tmp_path / "fallback_worker.py"
    script.write_text(FALLBACK_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CDKF_ROOT=ROOT, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert sum(so.count("RANK_OK_") for so, _ in outs) == world


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_launches_its_own_ranks(n):
    """`python bench.py --gpus 2` with no launcher in the environment starts its own two ranks as child processes: on
    this GPU-less box both children must get as far as the library's "no HIP device" error -- not a SystemExit in the parent asking
    for torch.distributed.run -- and the parent's exit code is the failing rank's."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the children would run the benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    import time
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-saturation"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert time.time() - t0 < 60, "the ranks of a launch that cannot work must be gone within a minute"
    # every child reaped: no process of this user still runs bench.py with this launch's rendezvous port in its environment
    left = subprocess.run(["pgrep", "-f", "bench.py --gpus %d --steps 1 --warmup 0" % n], capture_output=True, text=True).stdout.split()
    assert not left, left
    assert "torch.distributed.run" not in p.stderr, p.stderr
    assert p.stderr.count("hipSetDevice") + p.stderr.count("no ROCm-capable device") >= 2 or "stopping the other ranks" in p.stderr, p.stderr
    assert "bench: rank" in p.stderr and p.stdout.strip() == "", (p.stdout, p.stderr)


ONE_RANK_FAILS_WORKER = r'''
import os, sys
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if rank == 1:
    os.environ["CDKF_RCCL_PATH"] = "/nonexistent/librccl.so"   # this rank alone cannot even load RCCL
sys.path[:0] = [os.environ["CDKF_ROOT"]]
import numpy as np
from cd_dynamax_amd import distributed as D
comm = D.Comm(rank, world, "127.0.0.1", port, device=rank, timeout_ms=60000)
assert not comm._comm and comm.rccl_error, "a communicator cannot exist here"
s = comm.allreduce_sum_host([rank + 1.0])
assert np.allclose(s, [world * (world + 1) / 2]), s
comm.barrier()
comm.close()
sys.stdout.write("RANK_OK_%d %s\n" % (rank, comm.rccl_error.replace("\n", " ")[:120])); sys.stdout.flush()
'''


@pytest.mark.parametrize("world", [2, 3, 8])
def test_comm_preflight_failure_of_one_rank_strands_nobody(tmp_path, world):
    """a rank that fails BEFORE it would reach ncclCommInitRank (no librccl on its node: CDKF_RCCL_PATH names a missing
    file and is now the only candidate) is found out in the non-collective preflight (cdkf_comm_preflight) that all ranks take first;
    the ranks agree on it over the rendezvous and nobody enters the collective.  Rank 1's reason names its missing library, the other
    ranks learn that a peer failed (on this GPU-less box their own preflight fails on the device instead: different reasons on different
    ranks, one agreement).  The GPU leg -- rank 0 passes, rank 1 names a device that does not exist -- is tests/test_gpu_comm.py."""
    script = tmp_path / "one_rank_fails_worker.py"
    script.write_text(ONE_RANK_FAILS_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CDKF_ROOT=ROOT, OMP_NUM_THREADS="1")
    env.pop("CDKF_RCCL_PATH", None)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert sum(so.count("RANK_OK_") for so, _ in outs) == world
    assert "CDKF_RCCL_PATH=/nonexistent/librccl.so" in outs[1][0], outs[1][0]
