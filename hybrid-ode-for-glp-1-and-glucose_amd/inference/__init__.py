"""Host-side mirror of the reference's `inference` package for the part that sits on the hot path:
`VariationalInference` (reference inference/vi.py:19-340, BASELINE config 5).  `run_nuts` and the ArviZ I/O helpers of
reference inference/mcmc.py are a placeholder random-walk sampler plus CPU bookkeeping (out of
scope) and are not provided; unlike the reference's package, importing this one does not need arviz."""
from .vi import VariationalInference

__all__ = ["VariationalInference"]

_OUT_OF_SCOPE = ("run_nuts", "compute_ess", "posterior_summary", "save_mcmc_results", "load_mcmc_results")


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(f"inference.{name} (reference inference/mcmc.py) is outside the accelerated path and is not part of "
                             "this package; use the reference's own module for it")
    raise AttributeError(name)
