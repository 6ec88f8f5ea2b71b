"""cd_dynamax_amd -- MI355X-native continuous-discrete Gaussian filtering engine (host surface).

Same public names as /root/reference/src/continuous_discrete_nonlinear_gaussian_ssm/__init__.py for the
filtering / smoothing hot path; the arithmetic lives in hand-written HIP kernels behind the C ABI of
``include/cdkf.h``.
"""
from . import fit
from ._ffi import release_workspace
from . import mcmc
from .linear import (ContDiscreteLinearGaussianSSM, KFHyperParams, ParamsCDLGSSM, ParamsCDLGSSMDynamics,
                     ParamsLGSSMEmissions, cdlgssm_filter, cdlgssm_smoother)
from .models import (ContDiscreteNonlinearGaussianSSM, cdnlgssm_emissions, cdnlgssm_filter, cdnlgssm_forecast,
                     cdnlgssm_loglik_and_grad, cdnlgssm_loglik_and_grad_all, cdnlgssm_smoother)
from .params import (ConstantStepSize, PIDController, EKFHyperParams, EnKFHyperParams, GSSMForecast, LearnableCustomDrift, LearnableCustomEmission, LearnableLinear, LearnableLorenz63, LearnableLorenz96,
                     LearnableMatrix, LearnableMLP, LearnableVector, ParameterProperties, ParamsCDNLGSSM,
                     ParamsCDNLGSSMDynamics, ParamsCDNLGSSMEmissions, ParamsLGSSMInitial, PosteriorGSSMFiltered,
                     PosteriorGSSMSmoothed, UKFHyperParams)

__all__ = [
    "ContDiscreteNonlinearGaussianSSM", "cdnlgssm_filter", "cdnlgssm_smoother", "EKFHyperParams", "UKFHyperParams",
    "EnKFHyperParams", "LearnableVector", "LearnableMatrix", "LearnableLinear", "LearnableLorenz63",
    "LearnableLorenz96", "LearnableMLP", "ParameterProperties", "ParamsLGSSMInitial", "ParamsCDNLGSSMDynamics",
    "ParamsCDNLGSSMEmissions", "ParamsCDNLGSSM", "PosteriorGSSMFiltered", "PosteriorGSSMSmoothed",
    "ContDiscreteLinearGaussianSSM", "KFHyperParams", "ParamsCDLGSSM", "ParamsCDLGSSMDynamics", "ParamsLGSSMEmissions",
    "cdlgssm_filter", "cdlgssm_smoother", "cdnlgssm_forecast", "cdnlgssm_emissions", "GSSMForecast",
    "cdnlgssm_loglik_and_grad", "cdnlgssm_loglik_and_grad_all", "LearnableCustomDrift", "LearnableCustomEmission", "PIDController", "ConstantStepSize",
    "release_workspace",
]
