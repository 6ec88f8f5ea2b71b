"""Device-resident front-end: when ``emissions`` is a torch tensor that already lives on the GPU, the sweeps run on it in
place -- no PCIe transfer in either direction -- through the ``_dev`` entry points of the C ABI, on torch's current stream.
torch is plumbing here (device memory, the stream); the arithmetic is the HIP library's.  Results are torch tensors with the
reference's shapes: permuted VIEWS of buffers in the engine's native layout (``.contiguous()`` them if C-order is needed).

Host (NumPy) callers pay the copies: a Lorenz-63 batch of 4096 x 1000 is 2 ms of kernel time, 5 ms through host arrays for the
log-likelihood alone and 73 ms when all four moment arrays (786 MB) come back; from device tensors it is the kernel time.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi


def is_device_tensor(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch" and bool(getattr(x, "is_cuda", False))


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def prepare(emissions, t_emissions, opts):
    """(y [N,T,m], t [N,T] or [T], batched, numpy dtype) as contiguous device tensors; sets opts.t_shared / dt_final."""
    import torch
    y = emissions
    if y.dtype not in (torch.float32, torch.float64):
        y = y.double()
    if y.ndim == 1:
        y = y[:, None]
    batched = y.ndim == 3
    if not batched:
        y = y[None]
    N, T, _ = y.shape
    if t_emissions is None:
        t = torch.arange(T, dtype=y.dtype, device=y.device)
        opts.dt_final = 1.0
        opts.t_shared = 1
    else:
        t = torch.as_tensor(np.asarray(t_emissions) if not hasattr(t_emissions, "device") else t_emissions).to(device=y.device, dtype=y.dtype)
        if t.ndim == 3 or (t.ndim == 2 and t.shape[-1] != 1 and batched):
            t = t.reshape(N, T)
            opts.t_shared = 0
        else:
            t = t.reshape(-1)
            if t.shape[0] != T:
                raise ValueError(f"t_emissions has {t.shape[0]} time points but emissions has {T}")
            opts.t_shared = 1
    return y.contiguous(), t.contiguous(), batched, (np.float32 if y.dtype == torch.float32 else np.float64)


def _launch(fn, mdl, opts, y, *ptr_args):
    import torch
    opts.device = y.device.index if y.device.index is not None else torch.cuda.current_device()
    stream = C.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
    _ffi.check(fn(C.byref(mdl.c), C.byref(opts), y.shape[0], y.shape[1], *ptr_args, stream))


def run_device(algo: str, mdl: _ffi.ModelBlock, opts, t, y, want):
    """cdkf_<algo>_<f32|f64>_dev on device tensors (t [N,T] or [T], y [N,T,m]); returns (ll, [4 tensors or None], status) with the
    reference shapes [N,T,...] (views of native-layout buffers)."""
    import torch
    N, T, _ = y.shape
    d = mdl.state_dim
    suffix = "f32" if y.dtype == torch.float32 else "f64"
    opts.layout = _ffi.lib().cdkf_preferred_layout(C.byref(mdl.c))
    opts.layout_in = _ffi.LAYOUT_NT
    tcn = opts.layout == _ffi.LAYOUT_TCN
    kw = dict(dtype=y.dtype, device=y.device)
    ll = torch.empty(N, **kw)
    status = torch.zeros(N, dtype=torch.int32, device=y.device)
    shapes = [(T, d, N), (T, d, d, N), (T, d, N), (T, d, d, N)] if tcn else [(T, N, d), (T, N, d, d)] * 2
    outs = [torch.empty(s, **kw) if w else None for s, w in zip(shapes, want)]
    _launch(getattr(_ffi.lib(), f"cdkf_{algo}_{suffix}_dev"), mdl, opts, y, _p(t), _p(y), _p(ll), *[_p(o) for o in outs], _p(status))
    outs = [None if o is None else (o.movedim(-1, 0) if tcn else o.transpose(0, 1)) for o in outs]
    return ll, outs, status


def loglik_grad_device(mdl: _ffi.ModelBlock, opts, t, y, with_model: bool, ukf: bool = False):
    """cdkf_ekf_loglik_grad[_all]_<f32|f64>_dev on device tensors: (ll [N], grad [N, n_theta], status[, model block])."""
    import torch
    N = y.shape[0]
    suffix = "f32" if y.dtype == torch.float32 else "f64"
    opts.layout = _ffi.LAYOUT_TCN
    opts.layout_in = _ffi.LAYOUT_NT
    kw = dict(dtype=y.dtype, device=y.device)
    ll = torch.empty(N, **kw)
    grad = torch.empty(N, mdl.theta.size, **kw)
    status = torch.zeros(N, dtype=torch.int32, device=y.device)
    if with_model:
        gm = torch.empty(N, _ffi.model_grad_size(mdl.state_dim, mdl.emission_dim), **kw)
        _launch(getattr(_ffi.lib(), f"cdkf_{'ukf' if ukf else 'ekf'}_loglik_grad_all_{suffix}_dev"), mdl, opts, y, _p(t), _p(y), _p(ll), _p(grad),
                _p(gm), _p(status))
        return ll, grad, status, gm
    algo = "ukf" if ukf else "ekf"
    _launch(getattr(_ffi.lib(), f"cdkf_{algo}_loglik_grad_{suffix}_dev"), mdl, opts, y, _p(t), _p(y), _p(ll), _p(grad), _p(status))
    return ll, grad, status
