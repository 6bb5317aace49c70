"""Host-side mirror of the reference's `models` package: same class names, constructor and method
signatures, attribute names and state_dict keys (reference models/__init__.py:5-17), so that
`from models.hybrid_ode_nn import HybridODENN` keeps working for the reference's train/eval/test
code.  The hot path behind these classes (ode_residual / forward / loss) runs in the HIP kernels
of libhode.so through `hode`; there is no CPU fallback for it.
"""
from .ode_core import ODECore
from .nn_residual import NNResidual
from .hybrid_ode_nn import HybridODENN
from .bayes import bayes_loss, VariationalParameters, compute_posterior_predictive

__all__ = ["ODECore", "NNResidual", "HybridODENN", "bayes_loss", "VariationalParameters",
           "compute_posterior_predictive"]
