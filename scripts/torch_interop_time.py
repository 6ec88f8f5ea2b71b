"""Timing of the library + torch in one process (dev helper)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
t0 = time.time()
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from cd_dynamax_amd import _ffi
from helpers import params_from
L = _ffi.lib(); print("lib loaded", round(time.time() - t0, 1), flush=True)
mdl = o.lorenz63_model(2); rng = np.random.default_rng(0)
t = o.irregular_times(rng, 8, 10, 0.05); y = o.simulate(mdl, t, rng); P = params_from(mdl)
host = cd.cdnlgssm_filter(P, y, t[..., None]); print("host filter", round(time.time() - t0, 1), flush=True)
import torch; print("import torch", round(time.time() - t0, 1), flush=True)
yd = torch.from_numpy(y).cuda(); td = torch.from_numpy(t[..., None]).cuda(); torch.cuda.synchronize(); print("tensors on device", round(time.time() - t0, 1), flush=True)
dev = cd.cdnlgssm_filter(P, yd, td); torch.cuda.synchronize(); print("device filter", round(time.time() - t0, 1), flush=True)
print(np.abs(dev.filtered_means.cpu().numpy() - host.filtered_means).max())

# BASELINE config 2 from device tensors: wall time per call (torch caching allocator warm), all four moment arrays / ll only
mdl = o.lorenz63_model(3)
P = params_from(mdl)
N, T = 4096, 1000
t = o.irregular_times(rng, N, T, 0.01)
y = rng.standard_normal((N, T, 3))
yd = torch.from_numpy(y).cuda(); td = torch.from_numpy(t[..., None]).cuda()
model = cd.ContDiscreteNonlinearGaussianSSM(3, 3)
for name, fn in (("filter, 4 outputs", lambda: cd.cdnlgssm_filter(P, yd, td)),
                 ("marginal_log_prob", lambda: model.marginal_log_prob(P, yd, td)),
                 ("loglik_and_grad", lambda: cd.cdnlgssm_loglik_and_grad(P, yd, td))):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize(); t1 = time.time()
    for _ in range(10):
        r = fn()
    torch.cuda.synchronize()
    print(f"C2 device-resident {name}: {(time.time() - t1) * 100:.2f} ms per call", flush=True)
