"""gslnls_amd -- MI355X-native nonlinear least squares behind gslnls' gsl_nls() interface.

Host-side mirror of the reference's R front end (R/nls.R, R/nls_large.R) over the C ABI
of libgslnls_hip.so.  Only what the hot path needs lives here.
"""
from .control import gsl_nls_control, gsl_nls_loss  # noqa: F401
from .nls import gsl_nls, DenseProblem  # noqa: F401
from .nls_large import gsl_nls_large, LargeProblem  # noqa: F401
from .batch import BatchProblem  # noqa: F401

__all__ = ["gsl_nls", "gsl_nls_large", "gsl_nls_control", "gsl_nls_loss", "DenseProblem", "LargeProblem", "BatchProblem"]
