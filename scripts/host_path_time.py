"""End-to-end time of the Python host path on the C2 workload (dev helper): where does a drop-in user's time go?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import params_from
import bench
N, T = 4096, 1000
t, y = bench.make_batch(0, N, T)
P = params_from(o.lorenz63_model(3))
for rep in range(3):
    t0 = time.perf_counter(); post = cd.cdnlgssm_filter(P, y, t[..., None]); el = time.perf_counter() - t0
    t0 = time.perf_counter(); ll = cd.ContDiscreteNonlinearGaussianSSM(3, 3).marginal_log_prob(P, y, t[..., None]); el2 = time.perf_counter() - t0
    print(f"cdnlgssm_filter (4 outputs): {el*1e3:.1f} ms   marginal_log_prob: {el2*1e3:.1f} ms", flush=True)
# pieces
t0 = time.perf_counter(); yt = np.ascontiguousarray(y.transpose(1, 2, 0)); tt = np.ascontiguousarray(t.T); print(f"host transposes: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); outs = [np.empty(s) for s in ((T, 3, N), (T, 3, 3, N), (T, 3, N), (T, 3, 3, N))]; print(f"np.empty outputs: {(time.perf_counter()-t0)*1e3:.1f} ms")
d = _ffi.DeviceArray.from_numpy(yt)
t0 = time.perf_counter(); d2 = _ffi.DeviceArray.from_numpy(yt); print(f"H2D 98 MB (pageable): {(time.perf_counter()-t0)*1e3:.1f} ms")
big = _ffi.DeviceArray((T, 3, 3, N), np.float64)
t0 = time.perf_counter(); a = big.numpy(); print(f"D2H 295 MB into a fresh pageable array: {(time.perf_counter()-t0)*1e3:.1f} ms")
t0 = time.perf_counter(); a = big.numpy(); print(f"D2H again: {(time.perf_counter()-t0)*1e3:.1f} ms")
