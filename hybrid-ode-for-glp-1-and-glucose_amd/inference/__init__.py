"""Host-side mirror of the reference's `inference` package for the part that sits on the hot path:
`VariationalInference` (reference inference/vi.py:19-340, BASELINE config 5).

MERGED package.  The reference's callers import more from `inference` than this mirror provides:
train/train_hybrid.py:32-33 does `from inference.vi import VariationalInference` and `from inference.mcmc import run_nuts`,
inference/__init__.py:5-6 re-exports `run_nuts, compute_ess, posterior_summary, save_/load_mcmc_results`.  MCMC is outside
the accelerated path (a placeholder random-walk sampler + ArviZ bookkeeping) and is neither copied nor
stubbed here.  Instead the package path is extended over every `inference/` directory on `sys.path`: a submodule this
directory defines (`vi`) resolves HERE because this directory comes first, one it does not define (`mcmc`) falls through to
the reference's own file when the reference root is on `sys.path` -- which its scripts arrange themselves
(train/train_hybrid.py:28 appends it before the first import).  The path is re-scanned at every lookup, so it does not matter
whether the reference root was put on `sys.path` before or after this package was first imported.

Unlike the reference's package, importing this one does not need arviz: `inference.mcmc` is only loaded when asked for
(`from inference.mcmc import ...`, `from inference import run_nuts`, `inference.run_nuts`).  When no `inference/mcmc.py` can
be found the ImportError says where it lives."""
import importlib
import importlib.abc
import os
import sys

from .vi import VariationalInference

__all__ = ["VariationalInference", "run_nuts", "compute_ess", "posterior_summary", "save_mcmc_results", "load_mcmc_results"]

_MCMC_NAMES = ("run_nuts", "compute_ess", "posterior_summary", "save_mcmc_results", "load_mcmc_results")
_HERE = os.path.dirname(os.path.abspath(__file__))


class _MergedPath(list):
    """`__path__` of the merged package: this directory first, then every other `<sys.path entry>/inference/` that is a
    regular package (pkgutil.extend_path's rule), looked up afresh each time the import system walks it."""

    def _dirs(self):
        out = [_HERE]
        for entry in sys.path:
            if not isinstance(entry, str):
                continue
            d = os.path.join(entry or os.getcwd(), "inference")
            if os.path.isfile(os.path.join(d, "__init__.py")) and os.path.abspath(d) not in (os.path.abspath(p) for p in out):
                out.append(d)
        return out

    def __iter__(self):
        return iter(self._dirs())

    def __len__(self):
        return len(self._dirs())

    def __getitem__(self, i):
        return self._dirs()[i]

    def __repr__(self):
        return f"_MergedPath({self._dirs()!r})"


__path__ = _MergedPath([_HERE])


class _MissingMcmc(importlib.abc.MetaPathFinder):
    """Last finder on sys.meta_path: reached only when no `inference/mcmc.py` exists on the merged path."""

    def find_spec(self, fullname, path=None, target=None):
        if fullname == __name__ + ".mcmc":
            raise ModuleNotFoundError(
                "inference.mcmc (run_nuts and the ArviZ helpers, reference inference/mcmc.py) is outside the accelerated path "
                "and is not shipped with this package: put the reference repository root on sys.path (its scripts do, "
                "train/train_hybrid.py:28) and the reference's own inference/mcmc.py is used; it needs arviz", name=fullname)
        return None


if not any(isinstance(f, _MissingMcmc) for f in sys.meta_path):
    sys.meta_path.append(_MissingMcmc())


def __getattr__(name):
    if name in _MCMC_NAMES:
        return getattr(importlib.import_module(__name__ + ".mcmc"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
