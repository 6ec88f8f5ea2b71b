"""ctypes binding of the C ABI declared in ``include/cdkf.h`` (libcdkf_hip.so).

There is deliberately NO fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CDKF_LIB_PATH: development aid (A/B timing of two builds); the default is the in-tree library
LIB_PATH = os.environ.get("CDKF_LIB_PATH") or os.path.join(_HERE, "lib", "libcdkf_hip.so")

CDKF_OK = 0
CDKF_EINVAL = -1
CDKF_EUNSUPPORTED = -2
CDKF_EHIP = -3

STATUS_NOT_PD = 1
STATUS_NAN = 2
STATUS_MAX_STEPS = 4

DRIFT_LINEAR = 0
DRIFT_LORENZ63 = 1
DRIFT_LORENZ96 = 2
DRIFT_MLP_TANH = 3

LAYOUT_NT = 0
LAYOUT_TN = 1
LAYOUT_TCN = 2
LAYOUT_SAME = -1
FLAG_UKF_SIGMA_POINTS = 1  # cdkf_opts.flags (include/cdkf.h)

ORDER = {"zeroth": 0, "first": 1, "second": 2}

_dp = C.POINTER(C.c_double)


class CdkfModel(C.Structure):
    _fields_ = [
        ("drift_kind", C.c_int32),
        ("state_dim", C.c_int32),
        ("emission_dim", C.c_int32),
        ("hidden1", C.c_int32),
        ("hidden2", C.c_int32),
        ("emission_kind", C.c_int32),
        ("n_theta", C.c_int64),
        ("theta", _dp),
        ("L", _dp),
        ("Qc", _dp),
        ("H", _dp),
        ("h_bias", _dp),
        ("R", _dp),
        ("m0", _dp),
        ("P0", _dp),
        ("input_dim", C.c_int32),
        ("reserved0", C.c_int32),
    ]


class CdkfOpts(C.Structure):
    _fields_ = [
        ("state_order", C.c_int32),
        ("num_iter", C.c_int32),
        ("t_shared", C.c_int32),
        ("device", C.c_int32),
        ("layout", C.c_int32),
        ("forecast", C.c_int32),
        ("solver", C.c_int32),
        ("adaptive", C.c_int32),
        ("max_steps", C.c_int64),
        ("dt0", C.c_double),
        ("dt_final", C.c_double),
        ("cov_rescaling", C.c_double),
        ("ukf_alpha", C.c_double),
        ("ukf_beta", C.c_double),
        ("ukf_kappa", C.c_double),
        ("rtol", C.c_double),
        ("atol", C.c_double),
        ("pid_p", C.c_double),
        ("pid_i", C.c_double),
        ("pid_d", C.c_double),
        ("layout_in", C.c_int32),
        ("flags", C.c_int32),
        ("dtmin", C.c_double),
        ("dtmax", C.c_double),
        ("inputs", C.c_void_p),
        ("pid_safety", C.c_double),
        ("pid_factormin", C.c_double),
        ("pid_factormax", C.c_double),
    ]


class CdkfError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"cdkf error {code}: {msg}")
        self.code = code


class CdkfUnsupported(CdkfError, NotImplementedError):
    """CDKF_EUNSUPPORTED: no kernel for this (drift, shape, precision, option) combination -- the library's counterpart of the
    NotImplementedError the host layer raises for what it can rule out before the call."""


# every symbol include/cdkf.h declares (tests/test_abi.py checks this list against the header)
_ALGOS = ("ekf_filter", "ukf_filter", "ekf_smoother")
SYMBOLS = (
    ["cdkf_default_opts", "cdkf_version", "cdkf_last_error", "cdkf_device_count", "cdkf_supported",
     "cdkf_preferred_layout", "cdkf_trajectories_per_wavefront", "cdkf_malloc",
     "cdkf_free", "cdkf_memcpy_h2d", "cdkf_memcpy_d2h", "cdkf_memset", "cdkf_synchronize", "cdkf_ll_sum_f64_dev",
     "cdkf_ll_sum_f32_dev", "cdkf_emission_moments_f64", "cdkf_emission_moments_f32", "cdkf_emission_moments_f64_dev",
     "cdkf_emission_moments_f32_dev", "cdkf_custom_emission_moments_f64", "cdkf_custom_emission_moments_f32",
     "cdkf_custom_emission_moments_f64_dev", "cdkf_custom_emission_moments_f32_dev",
     "cdkf_custom_emission_moments_compile", "cdkf_ekf_loglik_grad_f64", "cdkf_ekf_loglik_grad_f32",
     "cdkf_ekf_loglik_grad_f64_dev", "cdkf_ekf_loglik_grad_f32_dev", "cdkf_ukf_loglik_grad_f64", "cdkf_ukf_loglik_grad_f32",
     "cdkf_ukf_loglik_grad_f64_dev", "cdkf_ukf_loglik_grad_f32_dev", "cdkf_ukf_grad_supported", "cdkf_custom_drift_register", "cdkf_custom_drift_compile", "cdkf_custom_emission_register", "cdkf_set_kernel_source_dir",
     "cdkf_kf_smoother1_f64", "cdkf_kf_smoother1_f32", "cdkf_kf_smoother1_f64_dev", "cdkf_kf_smoother1_f32_dev",
     "cdkf_kf_smoother1_supported", "cdkf_kf_pushforward_f64", "cdkf_kf_pushforward_f32", "cdkf_grad_supported", "cdkf_grad_all_supported", "cdkf_release_workspace", "cdkf_ekf_loglik_grad_all_f64",
     "cdkf_ekf_loglik_grad_all_f32", "cdkf_ekf_loglik_grad_all_f64_dev", "cdkf_ekf_loglik_grad_all_f32_dev", "cdkf_ekf_loglik_grad_jumps_f64",
     "cdkf_ekf_loglik_grad_jumps_f32", "cdkf_ukf_loglik_grad_all_f64", "cdkf_ukf_loglik_grad_all_f32", "cdkf_ukf_loglik_grad_all_f64_dev",
     "cdkf_ukf_loglik_grad_all_f32_dev", "cdkf_ukf_grad_all_supported", "cdkf_ukf_tangent_compile", "cdkf_ekf_tangent_compile", "cdkf_debug_ukf_tangent_args", "cdkf_grad_sum_f64_dev",
     "cdkf_grad_sum_f32_dev", "cdkf_comm_preflight", "cdkf_comm_unique_id", "cdkf_comm_init_rank", "cdkf_comm_init_all", "cdkf_comm_rank", "cdkf_comm_world",
     "cdkf_ll_allreduce", "cdkf_comm_allreduce_max", "cdkf_ll_allreduce_all", "cdkf_comm_destroy", "cdkf_rdv_create",
     "cdkf_rdv_broadcast", "cdkf_rdv_allreduce", "cdkf_rdv_barrier", "cdkf_rdv_destroy", "cdkf_last_kernel", "cdkf_event_create",
     "cdkf_event_record", "cdkf_event_elapsed_ms", "cdkf_event_destroy", "cdkf_stream_create", "cdkf_stream_destroy",
     "cdkf_set_device", "cdkf_struct_sizes", "cdkf_debug_exec_prologue_check", "cdkf_debug_custom_reg_blob", "cdkf_debug_wg_args", "cdkf_rtc_cache_stats"]
    + [f"cdkf_{a}_{p}{s}" for a in _ALGOS for p in ("f64", "f32") for s in ("", "_dev")]
)

_lib: Optional[C.CDLL] = None


def _share_torch_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 and load it by FILE name, this library
    asks for the SONAME: whichever comes first decides whether the other finds it already loaded.  With torch imported first
    everything shares torch's copy; with this library first, a later ``import torch`` (bench.py, the device-tensor front-end,
    torch.distributed) would bring a second runtime into the process and neither sees the GPU any more.  So when torch is
    installed its runtime libraries are loaded here, before ours, without importing torch itself."""
    import importlib.util
    import sys
    if os.environ.get("CDKF_SYSTEM_HIP_RUNTIME"):
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    # RCCL belongs to the HIP runtime it was built with: when the process runs on torch's runtime, the collective
    # (cdkf_comm_*, loaded on first use) takes torch's copy too
    if not os.environ.get("CDKF_RCCL_PATH") and os.path.exists(os.path.join(libdir, "librccl.so")):
        os.environ["CDKF_RCCL_PATH"] = os.path.join(libdir, "librccl.so")
    if "torch" in sys.modules:
        return  # torch's runtime is already in the process
    for name in ("libamd_comgr.so", "libamdhip64.so", "libhiprtc.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib() -> C.CDLL:
    """Load libcdkf_hip.so (once).  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"HIP library {LIB_PATH} is missing; build it with `make -C cd_dynamax_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback."
        )
    _share_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    L.cdkf_last_error.restype = C.c_char_p
    L.cdkf_version.restype = C.c_int
    if hasattr(L, "cdkf_struct_sizes"):   # the structs this module mirrors have grown (versions 107, 109): never hand a library a shorter one
        mb, ob = C.c_int64(0), C.c_int64(0)
        L.cdkf_struct_sizes.restype = None
        L.cdkf_struct_sizes(C.byref(mb), C.byref(ob))
        if (mb.value, ob.value) != (C.sizeof(CdkfModel), C.sizeof(CdkfOpts)):
            raise CdkfError(f"{LIB_PATH} was built for cdkf_model / cdkf_opts of {mb.value} / {ob.value} bytes (version {L.cdkf_version()}); "
                            f"this binding's structs are {C.sizeof(CdkfModel)} / {C.sizeof(CdkfOpts)}: rebuild the library or update the binding")
    L.cdkf_device_count.restype = C.c_int
    L.cdkf_default_opts.argtypes = [C.POINTER(CdkfOpts)]
    L.cdkf_default_opts.restype = None
    L.cdkf_supported.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int, C.c_int]
    L.cdkf_supported.restype = C.c_int
    L.cdkf_preferred_layout.argtypes = [C.POINTER(CdkfModel)]
    L.cdkf_preferred_layout.restype = C.c_int
    L.cdkf_trajectories_per_wavefront.argtypes = [C.c_int64]
    L.cdkf_trajectories_per_wavefront.restype = C.c_int
    L.cdkf_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_int64]
    L.cdkf_free.argtypes = [C.c_void_p]
    L.cdkf_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.cdkf_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.cdkf_memset.argtypes = [C.c_void_p, C.c_int, C.c_int64]
    L.cdkf_synchronize.argtypes = [C.c_void_p]
    for name in ("cdkf_ll_sum_f64_dev", "cdkf_ll_sum_f32_dev"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        getattr(L, name).restype = C.c_int
    for p in ("f64", "f32"):
        f = getattr(L, f"cdkf_emission_moments_{p}")
        f.argtypes = [C.POINTER(CdkfModel), C.c_int64] + [C.c_void_p] * 4
        f.restype = C.c_int
        f = getattr(L, f"cdkf_emission_moments_{p}_dev")
        f.argtypes = [C.POINTER(CdkfModel), C.c_int64] + [C.c_void_p] * 5
        f.restype = C.c_int
        f = getattr(L, f"cdkf_custom_emission_moments_{p}")
        f.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int, C.c_int64] + [C.c_void_p] * 6
        f.restype = C.c_int
        f = getattr(L, f"cdkf_custom_emission_moments_{p}_dev")
        f.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int, C.c_int64] + [C.c_void_p] * 7
        f.restype = C.c_int
    L.cdkf_custom_emission_moments_compile.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int]
    L.cdkf_custom_emission_moments_compile.restype = C.c_int
    L.cdkf_custom_drift_register.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
    L.cdkf_custom_drift_register.restype = C.c_int
    L.cdkf_custom_drift_compile.argtypes = [C.c_int] * 6
    L.cdkf_custom_emission_register.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    L.cdkf_custom_emission_register.restype = C.c_int
    L.cdkf_custom_drift_compile.restype = C.c_int
    L.cdkf_set_kernel_source_dir.argtypes = [C.c_char_p]
    L.cdkf_set_kernel_source_dir.restype = None
    L.cdkf_set_kernel_source_dir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc").encode())
    L.cdkf_kf_smoother1_supported.argtypes = [C.POINTER(CdkfModel)]
    L.cdkf_kf_smoother1_supported.restype = C.c_int
    for p in ("f64", "f32"):
        f = getattr(L, f"cdkf_kf_pushforward_{p}")
        f.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        f.restype = C.c_int
    for p in ("f64", "f32"):
        base = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64] + [C.c_void_p] * 9
        f = getattr(L, f"cdkf_kf_smoother1_{p}")
        f.argtypes = base
        f.restype = C.c_int
        f = getattr(L, f"cdkf_kf_smoother1_{p}_dev")
        f.argtypes = base + [C.c_void_p]
        f.restype = C.c_int
    L.cdkf_grad_supported.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts)]
    L.cdkf_grad_supported.restype = C.c_int
    L.cdkf_grad_all_supported.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts)]
    L.cdkf_grad_all_supported.restype = C.c_int
    if hasattr(L, "cdkf_ukf_grad_supported"):
        L.cdkf_ukf_grad_supported.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts)]
        L.cdkf_ukf_grad_supported.restype = C.c_int
    if hasattr(L, "cdkf_release_workspace"):  # (absent from libraries older than 107: A/B runs through CDKF_LIB_PATH)
        L.cdkf_release_workspace.argtypes = []
        L.cdkf_release_workspace.restype = C.c_int
    for p in ("f64", "f32"):
        base = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64] + [C.c_void_p] * 6
        f = getattr(L, f"cdkf_ekf_loglik_grad_all_{p}")
        f.argtypes = base
        f.restype = C.c_int
        f = getattr(L, f"cdkf_ekf_loglik_grad_all_{p}_dev")
        f.argtypes = base + [C.c_void_p]
        f.restype = C.c_int
        if hasattr(L, f"cdkf_ukf_loglik_grad_all_{p}"):  # (absent from libraries older than 108)
            f = getattr(L, f"cdkf_ukf_loglik_grad_all_{p}")
            f.argtypes = base
            f.restype = C.c_int
            f = getattr(L, f"cdkf_ukf_loglik_grad_all_{p}_dev")
            f.argtypes = base + [C.c_void_p]
            f.restype = C.c_int
            L.cdkf_ukf_grad_all_supported.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts)]
            L.cdkf_ukf_grad_all_supported.restype = C.c_int
        if hasattr(L, f"cdkf_ekf_loglik_grad_jumps_{p}"):  # (absent from libraries older than 108)
            f = getattr(L, f"cdkf_ekf_loglik_grad_jumps_{p}")
            f.argtypes = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64] + [C.c_void_p] * 9
            f.restype = C.c_int
    for p in ("f64", "f32"):
        base = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64] + [C.c_void_p] * 5
        for algo in ("ekf", "ukf"):
            if not hasattr(L, f"cdkf_{algo}_loglik_grad_{p}"):
                continue  # (the unscented entry points are absent from libraries older than 107)
            f = getattr(L, f"cdkf_{algo}_loglik_grad_{p}")
            f.argtypes = base
            f.restype = C.c_int
            f = getattr(L, f"cdkf_{algo}_loglik_grad_{p}_dev")
            f.argtypes = base + [C.c_void_p]
            f.restype = C.c_int
        f = getattr(L, f"cdkf_grad_sum_{p}_dev")
        f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        f.restype = C.c_int
    for a in _ALGOS:
        for p in ("f64", "f32"):
            base = [C.POINTER(CdkfModel), C.POINTER(CdkfOpts), C.c_int64, C.c_int64] + [C.c_void_p] * 8
            f = getattr(L, f"cdkf_{a}_{p}")
            f.argtypes = base
            f.restype = C.c_int
            f = getattr(L, f"cdkf_{a}_{p}_dev")
            f.argtypes = base + [C.c_void_p]
            f.restype = C.c_int
    L.cdkf_comm_preflight.argtypes = [C.c_int]
    L.cdkf_comm_unique_id.argtypes = [C.c_void_p]
    L.cdkf_comm_init_rank.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.cdkf_comm_init_all.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    L.cdkf_comm_rank.argtypes = [C.c_void_p]
    L.cdkf_comm_world.argtypes = [C.c_void_p]
    L.cdkf_ll_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.cdkf_comm_allreduce_max.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.cdkf_ll_allreduce_all.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_int64, C.POINTER(C.c_void_p)]
    L.cdkf_comm_destroy.argtypes = [C.c_void_p]
    L.cdkf_rdv_create.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.cdkf_rdv_broadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.cdkf_rdv_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
    L.cdkf_rdv_barrier.argtypes = [C.c_void_p]
    L.cdkf_rdv_destroy.argtypes = [C.c_void_p]
    L.cdkf_last_kernel.restype = C.c_char_p
    L.cdkf_event_create.argtypes = [C.POINTER(C.c_void_p)]
    L.cdkf_event_record.argtypes = [C.c_void_p, C.c_void_p]
    L.cdkf_event_elapsed_ms.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    L.cdkf_event_destroy.argtypes = [C.c_void_p]
    L.cdkf_stream_create.argtypes = [C.POINTER(C.c_void_p)]
    L.cdkf_stream_destroy.argtypes = [C.c_void_p]
    L.cdkf_set_device.argtypes = [C.c_int]
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != CDKF_OK:
        raise (CdkfUnsupported if rc == CDKF_EUNSUPPORTED else CdkfError)(rc, lib().cdkf_last_error().decode("utf-8", "replace"))


def default_opts() -> CdkfOpts:
    o = CdkfOpts()
    lib().cdkf_default_opts(C.byref(o))
    return o


class ModelBlock:
    """Owns the double-precision host arrays a ``cdkf_model`` points to."""

    def __init__(self, drift_kind, theta, L, Qc, H, h_bias, R, m0, P0, hidden=(0, 0), emission_kind=0):
        f64 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.float64))
        self.theta, self.L, self.Qc, self.H, self.h_bias, self.R, self.m0, self.P0 = map(
            f64, (theta, L, Qc, H, h_bias, R, m0, P0))
        d = self.m0.shape[0]
        m = self.h_bias.shape[0]
        if self.L.shape != (d, d) or self.Qc.shape != (d, d) or self.P0.shape != (d, d):
            raise ValueError("L, Qc and P0 must be [state_dim, state_dim]")
        if self.H.shape != (m, d) or self.R.shape != (m, m):
            raise ValueError("H must be [emission_dim, state_dim] and R [emission_dim, emission_dim]")
        self.state_dim, self.emission_dim = d, m
        ptr = lambda a: a.ctypes.data_as(_dp)
        self.c = CdkfModel(
            drift_kind=int(drift_kind), state_dim=d, emission_dim=m, hidden1=int(hidden[0]), hidden2=int(hidden[1]),
            emission_kind=int(emission_kind), n_theta=self.theta.size, theta=ptr(self.theta), L=ptr(self.L), Qc=ptr(self.Qc), H=ptr(self.H),
            h_bias=ptr(self.h_bias), R=ptr(self.R), m0=ptr(self.m0), P0=ptr(self.P0))


class DeviceArray:
    """A device allocation owned by the library's allocator (cdkf_malloc / cdkf_free), with NumPy upload / download."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = C.c_void_p()
        check(lib().cdkf_malloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        check(lib().cdkf_memcpy_h2d(d.ptr, a.ctypes.data_as(C.c_void_p), d.nbytes))
        return d

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape, self.dtype)
        check(lib().cdkf_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().cdkf_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _vp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_host(algo: str, mdl: ModelBlock, opts: CdkfOpts, t: np.ndarray, y: np.ndarray, want: dict, dtype):
    """Call cdkf_<algo>_<f32|f64> on host (NumPy) buffers.

    t: [N,T] or [T] (opts.t_shared), y: [N,T,m] in the reference layout.  ``want``: four booleans for
    the four optional output arrays in ABI order.  Returns (ll, [4 arrays or None], status); the
    arrays have the reference shapes [N,T,...].

    The outputs are produced in the layout the library prefers for this model (cdkf_preferred_layout:
    CDKF_LAYOUT_TCN = [T,w,N] for the lane-per-trajectory kernels, CDKF_LAYOUT_TN = [T,N,w] for the
    workgroup-per-trajectory ones) and come back as transposed VIEWS of those buffers (same shapes and values as
    the reference's arrays); the inputs are uploaded untransposed (opts.layout_in = CDKF_LAYOUT_NT).
    """
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    N, T, m = y.shape
    d = mdl.state_dim
    opts.layout = lib().cdkf_preferred_layout(C.byref(mdl.c))
    opts.layout_in = LAYOUT_NT  # t and y go over as they are ([N,T], [N,T,m]): no host-side transposition of the inputs
    tcn = opts.layout == LAYOUT_TCN
    t = np.ascontiguousarray(t, dtype=dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    ll = np.empty((N,), dtype)
    status = np.zeros((N,), np.int32)
    shapes = [(T, d, N), (T, d, d, N), (T, d, N), (T, d, d, N)] if tcn else [(T, N, d), (T, N, d, d)] * 2
    outs = [np.empty(s, dtype) if w else None for s, w in zip(shapes, want)]
    fn = getattr(lib(), f"cdkf_{algo}_{suffix}")
    check(fn(C.byref(mdl.c), C.byref(opts), N, T, _vp(t), _vp(y), _vp(ll), *[_vp(o) for o in outs], _vp(status)))
    outs = [None if o is None else (np.moveaxis(o, -1, 0) if tcn else np.swapaxes(o, 0, 1)) for o in outs]  # views
    return ll, outs, status


DRIFT_CUSTOM_BASE = 1000
SOLVERS = {"dopri5": 0, "tsit5": 1, "bosh3": 2, "heun": 3, "midpoint": 4, "ralston": 5, "euler": 6}


def register_custom_drift(state_dim: int, n_theta: int, f_src: str, jac_src: Optional[str], divgrad_src: Optional[str]) -> int:
    """cdkf_custom_drift_register: returns the drift_kind (same sources -> same kind).  ``jac_src`` None: the Jacobian is derived
    from ``f_src`` by dual numbers; ``divgrad_src`` "auto": so is grad(div f)."""
    kind = lib().cdkf_custom_drift_register(int(state_dim), int(n_theta), f_src.encode(), None if jac_src is None else jac_src.encode(),
                                            None if divgrad_src is None else divgrad_src.encode())
    if kind < 0:
        check(kind)
    return kind


def register_custom_emission(state_dim: int, emission_dim: int, h_src: str, hjac_src: str) -> int:
    """cdkf_custom_emission_register: returns the emission_kind (same sources -> same kind)."""
    kind = lib().cdkf_custom_emission_register(int(state_dim), int(emission_dim), h_src.encode(), None if hjac_src is None else hjac_src.encode())
    if kind < 0:
        check(kind)
    return kind


def kf_smoother1(mdl: ModelBlock, opts: CdkfOpts, t: np.ndarray, y: np.ndarray, dtype):
    """cdkf_kf_smoother1_<f32|f64> on host buffers: returns (ll, fm, fP, sm, sP, cross [N,T-1,d,d], status) in the
    reference shapes (views of [T,N,...] buffers)."""
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    N, T, m = y.shape
    d = mdl.state_dim
    opts.layout = LAYOUT_TN
    opts.layout_in = LAYOUT_NT
    t = np.ascontiguousarray(t, dtype=dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    ll = np.empty((N,), dtype)
    status = np.zeros((N,), np.int32)
    fm, sm = (np.empty((T, N, d), dtype) for _ in range(2))
    fP, sP, cr = (np.empty((T, N, d, d), dtype) for _ in range(3))
    fn = getattr(lib(), f"cdkf_kf_smoother1_{suffix}")
    check(fn(C.byref(mdl.c), C.byref(opts), N, T, _vp(t), _vp(y), _vp(ll), _vp(fm), _vp(fP), _vp(sm), _vp(sP), _vp(cr),
             _vp(status)))
    sw = lambda a: np.swapaxes(a, 0, 1)
    return ll, sw(fm), sw(fP), sw(sm), sw(sP), sw(cr)[:, :T - 1], status


def kf_pushforward(mdl: ModelBlock, opts: CdkfOpts, t: np.ndarray, dtype):
    """cdkf_kf_pushforward_<f32|f64>: the pushed-forward (A, Q) of every observation interval, each [N, T-1, d, d]; t [N,T]."""
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    t = np.ascontiguousarray(t, dtype=dtype)
    N, T = t.shape
    d = mdl.state_dim
    opts.layout_in = LAYOUT_NT
    opts.t_shared = 0
    AQ = np.zeros((N, max(T - 1, 0), 2, d, d), dtype)
    check(getattr(lib(), f"cdkf_kf_pushforward_{suffix}")(C.byref(mdl.c), C.byref(opts), N, T, _vp(t), _vp(AQ)))
    return AQ[:, :, 0], AQ[:, :, 1]


def release_workspace() -> None:
    """Return the reverse sweeps' device workspace (forward moments + stage checkpoints, grow-only, up to CDKF_ADJ_CKPT_GB) to the
    device once its last user has finished; the next gradient call allocates again."""
    check(lib().cdkf_release_workspace())


def model_grad_size(d: int, m: int) -> int:
    return d + 2 * d * d + m * d + m + m * m


def loglik_grad(mdl: ModelBlock, opts: CdkfOpts, t: np.ndarray, y: np.ndarray, dtype, with_model: bool = False, ukf: bool = False):
    """cdkf_ekf_loglik_grad[_all]_<f32|f64> (``ukf``: cdkf_ukf_loglik_grad_*) on host buffers (t [N,T] or [T], y [N,T,m]): returns
    (ll [N], grad [N, n_theta], status) and, ``with_model``, the model block [N, d + 2 d^2 + m d + m + m^2] as a fourth item."""
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    N, T, m = y.shape
    opts.layout = LAYOUT_TCN
    opts.layout_in = LAYOUT_NT  # inputs as they are; only ll [N] and grad [N, n_theta] come back
    t = np.ascontiguousarray(t, dtype=dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    ll = np.empty((N,), dtype)
    grad = np.empty((N, mdl.theta.size), dtype)
    status = np.zeros((N,), np.int32)
    if with_model:
        gm = np.empty((N, model_grad_size(mdl.state_dim, mdl.emission_dim)), dtype)
        fn = getattr(lib(), f"cdkf_{'ukf' if ukf else 'ekf'}_loglik_grad_all_{suffix}")
        check(fn(C.byref(mdl.c), C.byref(opts), N, T, _vp(t), _vp(y), _vp(ll), _vp(grad), _vp(gm), _vp(status)))
        return ll, grad, status, gm
    fn = getattr(lib(), f"cdkf_{'ukf' if ukf else 'ekf'}_loglik_grad_{suffix}")
    check(fn(C.byref(mdl.c), C.byref(opts), N, T, _vp(t), _vp(y), _vp(ll), _vp(grad), _vp(status)))
    return ll, grad, status


def loglik_grad_jumps(mdl: ModelBlock, opts: CdkfOpts, t: np.ndarray, y: np.ndarray, jumps: np.ndarray, dtype):
    """cdkf_ekf_loglik_grad_jumps_<f32|f64> on host buffers: the reverse sweep of a model whose predicted mean takes the jump
    ``jumps[n, k]`` behind interval k (the linear front-end's dynamics bias / inputs).  Returns (ll [N], grad [N, n_theta],
    grad_model [N, ...], grad_jumps [N, T, d], grad_y [N, T, m], status)."""
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    N, T, m = y.shape
    d = mdl.state_dim
    opts.layout = LAYOUT_TCN
    opts.layout_in = LAYOUT_NT
    t = np.ascontiguousarray(t, dtype=dtype)
    y = np.ascontiguousarray(y, dtype=dtype)
    jumps = np.ascontiguousarray(np.broadcast_to(jumps, (N, T, d)), dtype=dtype)
    ll = np.empty((N,), dtype)
    grad = np.empty((N, max(1, mdl.theta.size)), dtype)
    gm = np.empty((N, model_grad_size(d, m)), dtype)
    gj, gy = np.empty((N, T, d), dtype), np.empty((N, T, m), dtype)
    status = np.zeros((N,), np.int32)
    fn = getattr(lib(), f"cdkf_ekf_loglik_grad_jumps_{suffix}")
    check(fn(C.byref(mdl.c), C.byref(opts), N, T, _vp(t), _vp(y), _vp(jumps), _vp(ll), _vp(grad), _vp(gm), _vp(gj), _vp(gy), _vp(status)))
    return ll, grad[:, :mdl.theta.size], gm, gj, gy, status


def custom_emission_moments(mdl: ModelBlock, opts: "CdkfOpts", ukf: bool, t: Optional[np.ndarray], inputs: Optional[np.ndarray], means: np.ndarray,
                            covs: Optional[np.ndarray], dtype):
    """cdkf_custom_emission_moments_<f32|f64> on host buffers: means [..., d], covs [..., d, d] or None, t [...] or None,
    inputs [..., d_u] or None (an emission given as source: the extended and the unscented versions differ)."""
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    d, m = mdl.state_dim, mdl.emission_dim
    lead = means.shape[:-1]
    mu = np.ascontiguousarray(means, dtype).reshape(-1, d)
    rows = mu.shape[0]
    P = None if covs is None else np.ascontiguousarray(covs, dtype).reshape(rows, d, d)
    tt = None if t is None else np.ascontiguousarray(np.broadcast_to(np.asarray(t, dtype), lead), dtype).reshape(rows)
    uu = None
    if inputs is not None:
        u = np.asarray(inputs, dtype)
        uu = np.ascontiguousarray(np.broadcast_to(u, lead + u.shape[-1:]), dtype).reshape(rows, u.shape[-1])
        mdl.c.input_dim = int(u.shape[-1])
    om = np.empty((rows, m), dtype)
    oc = None if P is None else np.empty((rows, m, m), dtype)
    check(getattr(lib(), f"cdkf_custom_emission_moments_{suffix}")(C.byref(mdl.c), C.byref(opts), 1 if ukf else 0, rows, _vp(tt), _vp(uu), _vp(mu),
                                                                     _vp(P), _vp(om), _vp(oc)))
    return om.reshape(lead + (m,)), None if oc is None else oc.reshape(lead + (m, m))


def emission_moments(mdl: ModelBlock, means: np.ndarray, covs: Optional[np.ndarray], dtype):
    """cdkf_emission_moments_<f32|f64> on host buffers: means [..., d], covs [..., d, d] or None."""
    dtype = np.dtype(dtype)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]
    d, m = mdl.state_dim, mdl.emission_dim
    lead = means.shape[:-1]
    mu = np.ascontiguousarray(means, dtype).reshape(-1, d)
    rows = mu.shape[0]
    P = None if covs is None else np.ascontiguousarray(covs, dtype).reshape(rows, d, d)
    om = np.empty((rows, m), dtype)
    oc = None if P is None else np.empty((rows, m, m), dtype)
    check(getattr(lib(), f"cdkf_emission_moments_{suffix}")(C.byref(mdl.c), rows, _vp(mu), _vp(P), _vp(om), _vp(oc)))
    return om.reshape(lead + (m,)), None if oc is None else oc.reshape(lead + (m, m))
