"""Phase timing of the workgroup update (dev helper; needs a library built with -DCDKF_PHASE_PROFILE, CDKF_LIB_PATH)."""
import ctypes as C, sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")]
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from helpers import lorenz96_model, params_from
rng = np.random.default_rng(0)
mdl = lorenz96_model(40, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
N, T = 4, 60
t = o.irregular_times(rng, N, T, 0.3)
y = rng.standard_normal((N, T, mdl.m)) + 8.0
cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], cd.EKFHyperParams(state_order="first"), output_fields=[])
