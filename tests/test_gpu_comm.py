"""The data-parallel reduction of the C ABI on the device (cdkf_comm_*, cdkf_ll_allreduce: RCCL): the composition
sweep -> cdkf_ll_sum_*_dev -> all-reduce executes on the GPU with no host round trip in between
(replaces `vmap(...)(...).sum()` of /root/reference/src/ssm_temissions.py:555-568, 665-679).

A one-GPU box can only run world size 1 through RCCL itself; the multi-rank arithmetic is covered by the CPU tests of the
rendezvous (tests/test_distributed.py) and by the two-ranks-on-one-device attempt below, which RCCL may refuse."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import cdkf_oracle as o
from cd_dynamax_amd import _ffi, distributed as D
from cd_dynamax_amd.models import _model_block
from helpers import params_from

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(seed=0, N=37, T=40):
    rng = np.random.default_rng(seed)
    mdl = o.lorenz63_model(3)
    t = o.irregular_times(rng, N, T, 0.006 * T)
    return mdl, t, o.simulate(mdl, t, rng)


def test_sweep_llsum_allreduce_on_device_world_1(hip_lib):
    L = hip_lib
    mdl, t, y = _batch()
    N, T = t.shape
    blk = _model_block(params_from(mdl))
    opts = _ffi.default_opts()
    opts.layout = _ffi.LAYOUT_TCN
    comm = D.Comm(0, 1, "127.0.0.1", _free_port(), device=0)
    assert comm._comm, "a communicator on a device must hold an RCCL communicator"
    t_d = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(t.T))
    y_d = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(y.transpose(1, 2, 0)))
    ll = _ffi.DeviceArray((N,), np.float64)
    st = _ffi.DeviceArray((N,), np.int32)
    sums = _ffi.DeviceArray.from_numpy(np.full(4, np.nan))
    stream = C.c_void_p()
    _ffi.check(L.cdkf_stream_create(C.byref(stream)))
    try:
        _ffi.check(L.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), N, T, t_d.ptr, y_d.ptr, ll.ptr, None, None, None, None,
                                             st.ptr, stream))
        D.sharded_loglik_sum_dev(comm, ll.ptr, N, sums.ptr, stream)
        _ffi.check(L.cdkf_synchronize(stream))
        ref = o.ekf_filter(mdl, t, y)["marginal_loglik"]
        got = sums.numpy()
        assert abs(got[0] - ref.sum()) <= 1e-10 * abs(ref.sum())
        assert np.all(np.isnan(got[1:]))  # count = 1: the neighbours are untouched
        # max-reduction and a longer vector (the 1 + n_theta sums of the SGD objective)
        vec = _ffi.DeviceArray.from_numpy(np.arange(1.0, 5.0))
        comm.allreduce_sum_dev(vec.ptr, 4, stream)
        comm.allreduce_max_dev(vec.ptr, 4, stream)
        _ffi.check(L.cdkf_synchronize(stream))
        np.testing.assert_array_equal(vec.numpy(), np.arange(1.0, 5.0))
    finally:
        L.cdkf_stream_destroy(stream)
        comm.close()


def test_comm_init_all_single_process(hip_lib):
    """The single-process form (ncclCommInitAll) on the devices this box has."""
    L = hip_lib
    ndev = L.cdkf_device_count()
    comms = (C.c_void_p * ndev)()
    _ffi.check(L.cdkf_comm_init_all(comms, ndev, None))
    bufs, ptrs = [], (C.c_void_p * ndev)()
    for i in range(ndev):
        _ffi.check(L.cdkf_set_device(i))
        bufs.append(_ffi.DeviceArray.from_numpy(np.array([1.0 + i, 10.0 * (1 + i)])))
        ptrs[i] = bufs[-1].ptr
    _ffi.check(L.cdkf_set_device(0))
    _ffi.check(L.cdkf_ll_allreduce_all(comms, ndev, ptrs, 2, None))
    for i in range(ndev):
        _ffi.check(L.cdkf_set_device(i))
        _ffi.check(L.cdkf_synchronize(None))
        tot = sum(1.0 + j for j in range(ndev))
        np.testing.assert_array_equal(bufs[i].numpy(), [tot, 10.0 * tot])
        assert L.cdkf_comm_rank(comms[i]) == i and L.cdkf_comm_world(comms[i]) == ndev
    _ffi.check(L.cdkf_set_device(0))
    for c in comms:
        L.cdkf_comm_destroy(c)


WORKER = r'''
import ctypes as C, os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"], os.path.join(os.environ["CDKF_ROOT"], "oracle"), os.path.join(os.environ["CDKF_ROOT"], "tests")]
import numpy as np
import cdkf_oracle as o
from cd_dynamax_amd import _ffi, distributed as D
from cd_dynamax_amd.models import _model_block
from helpers import params_from
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = _ffi.lib()
comm = D.Comm(rank, world, "127.0.0.1", port, device=0, timeout_ms=60000)
if comm.rccl_error:   # both ranks then hold the same answer and go on through the host all-reduce
    assert not comm._comm
    sys.stdout.write("RCCL_REFUSED %s\n" % comm.rccl_error.replace("\n", " ")[:160])
rng = np.random.default_rng(0)
mdl = o.lorenz63_model(3)
N, T = 37, 40
t = o.irregular_times(rng, N, T, 0.006 * T)
y = o.simulate(mdl, t, rng)
lo, hi = D.shard_bounds(N, rank, world)
blk = _model_block(params_from(mdl))
opts = _ffi.default_opts(); opts.layout = _ffi.LAYOUT_TCN
t_d = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(t[lo:hi].T)); y_d = _ffi.DeviceArray.from_numpy(np.ascontiguousarray(y[lo:hi].transpose(1, 2, 0)))
ll = _ffi.DeviceArray((hi - lo,), np.float64); st = _ffi.DeviceArray((hi - lo,), np.int32); sums = _ffi.DeviceArray((1,), np.float64)
_ffi.check(L.cdkf_ekf_filter_f64_dev(C.byref(blk.c), C.byref(opts), hi - lo, T, t_d.ptr, y_d.ptr, ll.ptr, None, None, None, None, st.ptr, None))
D.sharded_loglik_sum_dev(comm, ll.ptr, hi - lo, sums.ptr, None)
_ffi.check(L.cdkf_synchronize(None))
ref = o.ekf_filter(mdl, t, y)["marginal_loglik"].sum()
got = float(sums.numpy()[0])
assert abs(got - ref) <= 1e-10 * abs(ref), (got, ref)
comm.barrier(); comm.close()
sys.stdout.write("RANK_OK_%d\n" % rank)
'''


def test_two_ranks_share_the_one_device_if_rccl_allows(tmp_path, hip_lib):
    """Two processes, both on device 0, through the whole composition.  RCCL refuses two ranks on one device on most builds
    ("Duplicate GPU detected"): neither rank may hang in the collective set-up -- both must learn of the refusal (Comm's agreement
    over the rendezvous), fall back to the host all-reduce and still produce the sharded log-likelihood sum."""
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    port = _free_port()
    env = dict(os.environ, CDKF_ROOT=ROOT)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two ranks on one device: RCCL neither joined nor refused within 240 s")
    text = "".join(so for so, _ in outs)
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert text.count("RANK_OK_") == 2, text
    assert text.count("RCCL_REFUSED") in (0, 2), text  # either both joined RCCL or both took the host path


PREFLIGHT_WORKER = r'''
import os, sys
sys.path[:0] = [os.environ["CDKF_ROOT"]]
import numpy as np
from cd_dynamax_amd import _ffi, distributed as D
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n_dev = _ffi.lib().cdkf_device_count()
device = 0 if rank == 0 else n_dev + 5      # rank 1 alone names a device that does not exist (LOCAL_RANK beyond its node's GPUs)
comm = D.Comm(rank, world, "127.0.0.1", port, device=device, timeout_ms=60000)
assert not comm._comm and comm.rccl_error, "rank 1 cannot join: nobody may"
s = comm.allreduce_sum_host([rank + 1.0])
assert np.allclose(s, [3.0]), s
comm.barrier(); comm.close()
sys.stdout.write("RANK_OK_%d %s\n" % (rank, comm.rccl_error.replace("\n", " ")[:120]))
'''


def test_preflight_failure_of_one_rank_strands_nobody(tmp_path, hip_lib):
    """rank 0 is healthy (device 0, RCCL loads), rank 1 names a device beyond the visible ones.  Before the
    preflight rank 1 returned early from cdkf_comm_init_rank while rank 0 sat in ncclCommInitRank's bootstrap (no timeout) for ever;
    now rank 1 fails cdkf_comm_preflight, the agreement that follows tells rank 0, and both finish on the host all-reduce."""
    script = tmp_path / "pf.py"
    script.write_text(PREFLIGHT_WORKER)
    port = _free_port()
    env = dict(os.environ, CDKF_ROOT=ROOT)
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(port)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=180))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank was left waiting inside the RCCL set-up")
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    assert "RANK_OK_0 another rank could not join" in outs[0][0], outs[0][0]
    assert "RANK_OK_1 cdkf_comm_preflight: device" in outs[1][0], outs[1][0]


def test_fit_sgd_reduces_on_the_device_through_rccl(hip_lib):
    """fit_sgd(comm=Comm(..., device=0)): the SGD step's reduction (ssm_temissions.py:555-568) stays on the device -- sweeps ->
    cdkf_ll_sum / cdkf_grad_sum -> ONE in-place ncclAllReduce of 2 + n_theta + n_model doubles through a real RCCL communicator
    (world 1 on this box) -> a single copy of the reduced block.  Same numbers as the plain single-process fit: drift-only
    (three parameters) and every leaf (the model block as well), minibatches that do not divide the data."""
    import cd_dynamax_amd as cd
    from cd_dynamax_amd import fit
    from cd_dynamax_amd.bijectors import RealToPSDBijector
    from cd_dynamax_amd.params import ParameterProperties as PP
    from test_fit import _l63_problem
    comm = D.Comm(0, 1, "127.0.0.1", _free_port(), device=0)
    assert comm._comm
    calls = []
    orig = comm.allreduce_sum_dev
    comm.allreduce_sum_dev = lambda ptr, count, stream=None: (calls.append(count), orig(ptr, count, stream))[1]
    try:
        model, params, props = _l63_problem(m=3)
        rng = np.random.default_rng(4)
        true = o.lorenz63_model(3)
        N, T = 11, 40
        t = o.irregular_times(rng, N, T, 0.01 * T)
        y = o.simulate(true, t, rng)
        kw = dict(optimizer=fit.SGD(0.05), batch_size=4, num_epochs=2)
        ref, ref_losses = model.fit_sgd(params, props, y, t[..., None], cd.EKFHyperParams(), **kw)
        got, losses = model.fit_sgd(params, props, y, t[..., None], cd.EKFHyperParams(), comm=comm, **kw)
        # data-parallel pieces are np.array_split pieces (4, 4, 3) -- here identical to the consecutive slices of 4, 4, 3
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-13)
        np.testing.assert_allclose([got.dynamics.drift.sigma, got.dynamics.drift.rho, got.dynamics.drift.beta],
                                   [ref.dynamics.drift.sigma, ref.dynamics.drift.rho, ref.dynamics.drift.beta], rtol=1e-13)
        assert calls == [2 + 3] * 6, calls  # [sum ll | d/d(sigma, rho, beta) | B], one collective per step
        # every leaf: + m0, P0, L Qc L^T, H, bias, R through the same single collective
        calls.clear()
        psd, free = PP(constrainer=RealToPSDBijector()), PP()
        allp = params._replace(
            initial=params.initial._replace(mean=cd.LearnableVector(free), cov=cd.LearnableMatrix(psd)),
            dynamics=params.dynamics._replace(drift=cd.LearnableLorenz63(free, free, free), diffusion_coefficient=cd.LearnableMatrix(free),
                                              diffusion_cov=cd.LearnableMatrix(psd), approx_order=PP(False)),
            emissions=params.emissions._replace(emission_function=cd.LearnableLinear(free, free), emission_cov=cd.LearnableMatrix(psd)))
        ref, ref_losses = model.fit_sgd(params, allp, y, t[..., None], cd.EKFHyperParams(), **kw)
        got, losses = model.fit_sgd(params, allp, y, t[..., None], cd.EKFHyperParams(), comm=comm, **kw)
        np.testing.assert_allclose(losses, ref_losses, rtol=1e-13)
        np.testing.assert_allclose(got.emissions.emission_cov.params, ref.emissions.emission_cov.params, rtol=1e-12)
        np.testing.assert_allclose(got.initial.mean.params, ref.initial.mean.params, rtol=1e-12, atol=1e-15)
        assert calls == [2 + 3 + _ffi.model_grad_size(3, 3)] * 6, calls
    finally:
        comm.close()
