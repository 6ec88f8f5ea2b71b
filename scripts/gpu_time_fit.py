"""Per-step cost of fit_sgd on the benchmark's Lorenz-63 batch (dev helper): device value + gradient of every leaf, device reduction,
host Adam."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import bench, cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import fit, _ffi
from helpers import params_from
N, T = 4096, 1000
t, y = bench.make_batch(7, N, T)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_fit import _l63_problem
model, params, props = _l63_problem(3)
for steps in (2, 12):
    t0 = time.perf_counter()
    out = model.fit_sgd(params, props, y, t[..., None], cd.EKFHyperParams(), optimizer=fit.Adam(1e-3), batch_size=N, num_epochs=steps)
    el = time.perf_counter() - t0
    print(f"fit_sgd {steps} epochs of one {N} x {T} minibatch: {el*1e3:.1f} ms   kernel: {_ffi.lib().cdkf_last_kernel().decode()[:48]}")
    prev = el if steps == 2 else prev
print(f"per step: {(el - prev) / 10 * 1e3:.2f} ms")
