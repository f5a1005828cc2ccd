"""outerbase_amd -- MI355X-native outer-product basis regression (hot path of
the R package outerbase) behind the reference's `obmod` module surface.

The arithmetic lives in libobhip.so (HIP kernels for gfx950 + C ABI declared in
include/obhip.h); this package is the host-side mirror of the reference's
Rcpp module (class and method names of src/interfaceR.cpp:661-793) plus the
obfit/obpred harness (R/fitting.R).  There is no CPU fallback.
"""
from . import _lib
from ._lib import ObhipError, device_count
from .obmod import (covf, covf_mat25, covf_mat25ang, covf_mat25pow, gethyp, getpara, hypnames,
                    listcov, loglik_gauss, loglik_gda, loglik_std, logpr_gauss, lpdf, lpdfvec,
                    outerbase, outermod, predictor, setcovfs, setknot)
from .fitting import BFGS_lpdf, BFGS_std, obfit, obpred

__all__ = [
    "ObhipError", "device_count", "covf", "covf_mat25", "covf_mat25ang", "covf_mat25pow", "gethyp",
    "getpara", "hypnames", "listcov", "loglik_gauss", "loglik_gda", "loglik_std", "logpr_gauss",
    "lpdf", "lpdfvec", "outerbase", "outermod", "predictor", "setcovfs", "setknot",
    "BFGS_lpdf", "BFGS_std", "obfit", "obpred",
]
