"""Reverse sweep on the lane grid against the forward-sensitivity kernel and the oracle (dev helper)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import cdkf_oracle as o
    import cd_dynamax_amd as cd
    from helpers import params_from
    rng = np.random.default_rng(77)
    mdl = o.lorenz63_model(3)
    out = {}
    for name, N, T, gap in (("a", 70, 40, 0.2), ("b", 9, 30, 0.008), ("c", 5, 4, 2.4)):  # (total spans)
        t = o.irregular_times(rng, N, T, gap)
        y = o.simulate(mdl, t, rng)
        ll, g = cd.cdnlgssm_loglik_and_grad(params_from(mdl), y, t[..., None])
        out[name + "_ll"] = ll; out[name + "_g"] = np.stack([g.sigma, g.rho, g.beta], -1)
        out[name + "_t"] = t; out[name + "_y"] = y
    from cd_dynamax_amd import _ffi
    print("kernel:", _ffi.lib().cdkf_last_kernel().decode())
    np.savez(sys.argv[2], **out)
    sys.exit(0)
import cdkf_oracle as o
res = {}
for tag, env in (("grid", {}), ("sens", {"CDKF_NO_LPE_GRAD": "1"})):
    f = f"/tmp/lpe_grad_{tag}.npz"
    subprocess.run([sys.executable, __file__, "child", f], check=True, env={**os.environ, **env})
    res[tag] = np.load(f)
mdl = o.lorenz63_model(3)
for name in "abc":
    a, b = res["grid"], res["sens"]
    scale = np.abs(b[name + "_g"]).max(axis=0, keepdims=True)
    print(name, "grid vs sens: ll", np.max(np.abs(a[name + "_ll"] - b[name + "_ll"]) / np.abs(b[name + "_ll"])),
          "grad", np.max(np.abs(a[name + "_g"] - b[name + "_g"]) / scale))
    if name != "a":
        ll_ref, g_ref = o.ekf_loglik_grad(mdl, a[name + "_t"], a[name + "_y"])
        print(name, "grid vs oracle:", np.max(np.abs(a[name + "_g"] - g_ref) / scale), " sens vs oracle:", np.max(np.abs(b[name + "_g"] - g_ref) / scale))
        if name == "c":
            print(a[name + "_g"][:2], g_ref[:2])
# repeated observation times (two observations at one instant): no predict between them, in either sweep
import cd_dynamax_amd as cd
from helpers import params_from
rng = np.random.default_rng(4)
t = o.irregular_times(rng, 5, 12, 0.3)
t[:, 6] = t[:, 5]
y = o.simulate(mdl, t, rng)
ll_ref, g_ref = o.ekf_loglik_grad(mdl, t, y)
ll, g = cd.cdnlgssm_loglik_and_grad(params_from(mdl), y, t[..., None])
gd = np.stack([g.sigma, g.rho, g.beta], -1)
print("repeated times: ll", np.max(np.abs(ll - ll_ref) / np.abs(ll_ref)), "grad", np.max(np.abs(gd - g_ref)) / np.abs(g_ref).max())
