"""Host-side mirror of the reference's `models` package (same public names as reference
models/__init__.py:5-17), so `from models.hybrid_ode_nn import HybridODENN` and
`from models import VariationalParameters` keep working for the reference's train / eval / test code.

The hot path behind these classes (HybridODENN.ode_residual / forward / loss / elbo) runs in the HIP
kernels of libhode.so through the `hode` binding; there is no CPU fallback for it.
"""
import importlib

# public name -> defining submodule
_EXPORTS = {
    "HybridODENN": "hybrid_ode_nn",
    "ODECore": "ode_core",
    "NNResidual": "nn_residual",
    "VariationalParameters": "bayes",
    "bayes_loss": "bayes",
    "compute_posterior_predictive": "bayes",
}
__all__ = sorted(_EXPORTS)

for _name, _mod in _EXPORTS.items():
    globals()[_name] = getattr(importlib.import_module(f"{__name__}.{_mod}"), _name)
del _name, _mod
