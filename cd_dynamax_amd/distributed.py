"""Multi-GPU layer: one process per GPU, trajectories sharded in contiguous blocks, ONE collective per
sweep -- the all-reduce of the summed marginal log-likelihood (RCCL over xGMI; backend "nccl" on ROCm).

This replaces the single-device ``vmap(marginal_log_prob)(...).sum()`` of the reference's training losses
(/root/reference/src/ssm_temissions.py:555-568 for fit_sgd, :665-679 for fit_mcmc).  Trajectories share
the parameters and nothing else, so no data-path collective exists: each rank filters its own block and
only the scalar crosses the fabric.  torch.distributed is plumbing here (rendezvous + the collective);
the arithmetic is in the HIP library.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import numpy as np


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
    buf = torch.as_tensor(value, dtype=torch.float64, device=device).reshape(1).clone()
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return float(buf.item())


def sharded_marginal_log_prob(local_ll_fn: Callable[[int, int], np.ndarray], n_total: int) -> float:
    """sum_n log p(y_n): every rank evaluates ``local_ll_fn(lo, hi)`` (per-trajectory log-likelihoods of its
    block, e.g. ``model.marginal_log_prob(params, y[lo:hi], t[lo:hi])``) and the block sums are all-reduced."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n_total, rank, world)
    local = float(np.sum(np.asarray(local_ll_fn(lo, hi), dtype=np.float64))) if hi > lo else 0.0
    return allreduce_sum(local)


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


def sharded_loglik_and_grad(local_fn: Callable[[int, int], Tuple[np.ndarray, np.ndarray]], n_total: int, n_theta: int):
    """(sum_n ll_n, sum_n d ll_n / d theta): every rank evaluates ``local_fn(lo, hi) -> (ll [B], grad [B, n_theta])`` on
    its block (cdkf_ekf_loglik_grad_*), and the 1 + n_theta block sums cross the fabric in ONE all-reduce -- the data-
    parallel form of ``value_and_grad`` of the fit_sgd loss (ssm_temissions.py:550-568)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n_total, rank, world)
    packed = np.zeros(1 + n_theta)
    if hi > lo:
        ll, grad = local_fn(lo, hi)
        packed[0] = np.sum(np.asarray(ll, dtype=np.float64))
        packed[1:] = np.sum(np.asarray(grad, dtype=np.float64).reshape(hi - lo, n_theta), axis=0)
    packed = allreduce_sum_array(packed)
    return float(packed[0]), packed[1:]
