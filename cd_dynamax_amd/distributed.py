"""Multi-GPU layer: one process per GPU, trajectories sharded in contiguous blocks, ONE collective per
sweep -- the all-reduce of the summed marginal log-likelihood (RCCL over xGMI).

This replaces the single-device ``vmap(marginal_log_prob)(...).sum()`` of the reference's training losses
(/root/reference/src/ssm_temissions.py:555-568 for fit_sgd, :665-679 for fit_mcmc).  Trajectories share
the parameters and nothing else, so no data-path collective exists: each rank filters its own block and
only the scalar (or the 1 + n_theta value-and-gradient sums) crosses the fabric.

``Comm`` is the library's own path (no torch): the C ABI's TCP rendezvous hands RCCL's id around and sums host
doubles, ``cdkf_ll_allreduce`` sums the device-resident block sums in place on the sweep's stream.  The
``torch.distributed`` helpers further down serve callers that already run a process group (they are what the CPU
tests drive over gloo).
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import numpy as np


class Comm:
    """Data-parallel communicator of one rank.  ``from_env()`` reads what ``torch.distributed.run`` / any launcher exports:
    RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR and MASTER_PORT -- the rendezvous listens on MASTER_PORT + 1 (the launcher's own
    store owns MASTER_PORT); CDKF_RDV_PORT overrides.  ``device`` None: host-only (no RCCL communicator; CPU tests)."""

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29501, device=None, timeout_ms: int = 120000):
        import ctypes as C
        from . import _ffi
        self._C, self._ffi, self._L = C, _ffi, _ffi.lib()
        self.rank, self.world, self.device = int(rank), int(world), device
        self._rdv, self._comm = C.c_void_p(), C.c_void_p()
        _ffi.check(self._L.cdkf_rdv_create(C.byref(self._rdv), addr.encode(), int(port), self.rank, self.world, int(timeout_ms)))
        self.rccl_error = None  # why there is no RCCL communicator although a device was named (every rank then holds the same answer)
        if device is not None:
            # Joining RCCL is a collective: a rank that cannot (no library, no device) must not leave the others waiting in it.  Every
            # step of the set-up is therefore taken by ALL ranks, and after each they agree over the rendezvous (a max of failure flags)
            # whether to go on; if any rank failed, all of them fall back to the host all-reduce of the rendezvous and say so.
            # First the part a rank can check ALONE (cdkf_comm_preflight: RCCL loads, the device exists): ncclCommInitRank's bootstrap has
            # no timeout, so a rank that would fail before reaching it -- no librccl on its node, LOCAL_RANK beyond its devices -- must be
            # found out while nobody is inside the collective yet.
            ident = C.create_string_buffer(128)
            err = ""
            if self._L.cdkf_comm_preflight(int(device)) != 0:
                err = self._L.cdkf_last_error().decode("utf-8", "replace")
            failed = bool(self._host([1.0 if err else 0.0], 1)[0])
            if not failed:
                if self.rank == 0 and self._L.cdkf_comm_unique_id(ident) != 0:
                    err = self._L.cdkf_last_error().decode("utf-8", "replace")
                _ffi.check(self._L.cdkf_rdv_broadcast(self._rdv, ident, 128))
                failed = bool(self._host([1.0 if err else 0.0], 1)[0])
            if not failed:
                if self._L.cdkf_comm_init_rank(C.byref(self._comm), ident, self.rank, self.world, int(device)) != 0:
                    err = self._L.cdkf_last_error().decode("utf-8", "replace")
                failed = bool(self._host([1.0 if err else 0.0], 1)[0])
            if failed:
                if self._comm:
                    self._L.cdkf_comm_destroy(self._comm)
                self._comm = C.c_void_p()
                self.rccl_error = err or "another rank could not join the RCCL communicator"

    @classmethod
    def from_env(cls, gpu: bool = True, timeout_ms: int = 120000) -> "Comm":
        rank, local_rank, world = env_rank_world()
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("CDKF_RDV_PORT", int(os.environ.get("MASTER_PORT", 29500)) + 1))
        return cls(rank, world, addr, port, device=local_rank if gpu else None, timeout_ms=timeout_ms)

    def _host(self, values, op: int) -> np.ndarray:
        arr = np.ascontiguousarray(values, dtype=np.float64).copy()
        flat = arr.reshape(-1)
        self._ffi.check(self._L.cdkf_rdv_allreduce(self._rdv, flat.ctypes.data_as(self._C.c_void_p), flat.size, op))
        return arr

    def allreduce_sum_host(self, values) -> np.ndarray:
        """Element-wise sum of a small float64 array over all ranks (rank order: the same bits on every run and rank)."""
        return self._host(values, 0)

    def allreduce_max_host(self, values) -> np.ndarray:
        return self._host(values, 1)

    def barrier(self) -> None:
        self._ffi.check(self._L.cdkf_rdv_barrier(self._rdv))

    def allreduce_sum_dev(self, sums_ptr, count: int, stream=None) -> None:
        """In-place RCCL sum of ``count`` device doubles (the output of cdkf_ll_sum_*_dev / cdkf_grad_sum_*_dev) on ``stream``."""
        if not self._comm:
            raise RuntimeError("Comm was created without a device: no RCCL communicator")
        self._ffi.check(self._L.cdkf_ll_allreduce(self._comm, sums_ptr, int(count), stream))

    def allreduce_sum_any(self, sums_ptr, count: int, stream=None) -> None:
        """``allreduce_sum_dev`` where an RCCL communicator exists; otherwise (``rccl_error``) the same sum through the host: the stream is
        drained, the doubles cross the rendezvous' TCP star and return to the device buffer -- slower, same result on every rank."""
        if self._comm:
            return self.allreduce_sum_dev(sums_ptr, count, stream)
        C, L = self._C, self._L
        self._ffi.check(L.cdkf_synchronize(stream))
        host = np.zeros(int(count))
        self._ffi.check(L.cdkf_memcpy_d2h(host.ctypes.data_as(C.c_void_p), sums_ptr, host.nbytes))
        host = self._host(host, 0)
        self._ffi.check(L.cdkf_memcpy_h2d(sums_ptr, host.ctypes.data_as(C.c_void_p), host.nbytes))

    def allreduce_max_dev(self, ptr, count: int, stream=None) -> None:
        if not self._comm:
            raise RuntimeError("Comm was created without a device: no RCCL communicator")
        self._ffi.check(self._L.cdkf_comm_allreduce_max(self._comm, ptr, int(count), stream))

    def close(self) -> None:
        if self._comm:
            self._L.cdkf_comm_destroy(self._comm)
            self._comm = self._C.c_void_p()
        if self._rdv:
            self._L.cdkf_rdv_destroy(self._rdv)
            self._rdv = self._C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of trajectories owned by ``rank``; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend: str | None = None):
    """Rendezvous from the torchrun environment (MASTER_ADDR/PORT, RANK, WORLD_SIZE).  Returns
    (rank, local_rank, world).  backend defaults to nccl (= RCCL) when a GPU is visible, else gloo."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def allreduce_sum(value, device=None) -> float:
    """Sum a Python float / 0-d tensor over all ranks (no-op for a single process)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    return float(allreduce_sum_array(np.array([float(value)]), device=device)[0])


def _rank_world(comm):
    if comm is not None:
        return comm.rank, comm.world
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def sharded_marginal_log_prob(local_ll_fn: Callable[[int, int], np.ndarray], n_total: int, comm: "Comm | None" = None) -> float:
    """sum_n log p(y_n): every rank evaluates ``local_ll_fn(lo, hi)`` (per-trajectory log-likelihoods of its
    block, e.g. ``model.marginal_log_prob(params, y[lo:hi], t[lo:hi])``) and the block sums are all-reduced --
    through ``comm`` (the library's own collective) or, without one, through the torch process group."""
    rank, world = _rank_world(comm)
    lo, hi = shard_bounds(n_total, rank, world)
    local = float(np.sum(np.asarray(local_ll_fn(lo, hi), dtype=np.float64))) if hi > lo else 0.0
    if comm is not None:
        return float(comm.allreduce_sum_host([local])[0])
    return allreduce_sum(local)


def sharded_loglik_sum_dev(comm: "Comm", ll_ptr, n_local: int, sums_ptr, stream=None, suffix: str = "f64") -> None:
    """The device-resident composition of the sharded marginal log-likelihood: the block's per-trajectory log-likelihoods
    (``ll_ptr``, just written by a ``_dev`` sweep on ``stream``) are summed on the device into ``sums_ptr[0]`` and that double is
    all-reduced in place over RCCL on the same stream.  Nothing returns to the host; read ``sums_ptr`` after a synchronize."""
    from . import _ffi
    L = _ffi.lib()
    _ffi.check(getattr(L, f"cdkf_ll_sum_{suffix}_dev")(ll_ptr, int(n_local), sums_ptr, stream))
    if comm.world > 1 or comm._comm:
        comm.allreduce_sum_any(sums_ptr, 1, stream)  # RCCL, or -- where it could not be joined (comm.rccl_error) -- through the host


def allreduce_sum_array(values, device=None) -> np.ndarray:
    """Element-wise sum of a small float64 array over all ranks (one collective)."""
    import torch
    import torch.distributed as dist
    arr = np.ascontiguousarray(values, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return arr.copy()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    buf = torch.from_numpy(arr.reshape(-1).copy()).to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf.cpu().numpy().reshape(arr.shape)


def sharded_loglik_and_grad(local_fn: Callable[[int, int], Tuple[np.ndarray, np.ndarray]], n_total: int, n_theta: int,
                            comm: "Comm | None" = None):
    """(sum_n ll_n, sum_n d ll_n / d theta): every rank evaluates ``local_fn(lo, hi) -> (ll [B], grad [B, n_theta])`` on
    its block (cdkf_ekf_loglik_grad_*), and the 1 + n_theta block sums cross the fabric in ONE all-reduce -- the data-
    parallel form of ``value_and_grad`` of the fit_sgd loss (ssm_temissions.py:550-568)."""
    rank, world = _rank_world(comm)
    lo, hi = shard_bounds(n_total, rank, world)
    packed = np.zeros(1 + n_theta)
    if hi > lo:
        ll, grad = local_fn(lo, hi)
        packed[0] = np.sum(np.asarray(ll, dtype=np.float64))
        packed[1:] = np.sum(np.asarray(grad, dtype=np.float64).reshape(hi - lo, n_theta), axis=0)
    packed = comm.allreduce_sum_host(packed) if comm is not None else allreduce_sum_array(packed)
    return float(packed[0]), packed[1:]
