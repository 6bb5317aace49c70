"""Run the reference's OWN test files against this package (build container only: the reference cannot
travel to the GPU box, where the same assertions are restated in tests/test_host_surface.py and
tests/test_models_gpu.py).  north_star: "pass tests/test_ode_jacobians.py"."""
import os
import subprocess
import sys

import pytest

REF_TESTS = "/root/reference/tests"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference checkout not present")


def _run(args):
    env = dict(os.environ, PYTHONPATH=PKG)      # the reference appends its own root to sys.path: PYTHONPATH wins
    return subprocess.run([sys.executable, "-m", "pytest", "-p", "no:cacheprovider", "-q"] + args, cwd="/tmp", env=env,
                          capture_output=True, text=True, timeout=600)


def test_reference_test_ode_jacobians_passes_against_this_package():
    r = _run([os.path.join(REF_TESTS, "test_ode_jacobians.py")])
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_reference_nn_residual_gradient_test_passes_against_this_package():
    # the other tests of that file integrate (HybridODENN.loss) and therefore need the GPU: they are restated in
    # tests/test_models_gpu.py
    r = _run([os.path.join(REF_TESTS, "test_gradient_correctness.py"), "-k", "nn_residual"])
    assert r.returncode == 0 and "1 passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_import_block_of_the_reference_trainer_resolves_under_the_drop_in_path():
    """PYTHONPATH=<package dir>, then the reference's script appends its own root
    (/root/reference/train/train_hybrid.py:28) and imports `models.hybrid_ode_nn`, `models.ode_core`, `inference.vi` and
    `inference.mcmc` (:30-33); /root/reference/inference/__init__.py:5-6 imports `.vi` and `.mcmc` the same way.  `models.*`
    and `inference.vi` must resolve to this repository, `inference.mcmc` to the reference's own file (never a copy or a stub).
    arviz is not installed in the build container: executing the reference's mcmc.py then stops at ITS `import arviz` -- an
    ordinary ModuleNotFoundError that names arviz, which is the reference's own behaviour on this machine."""
    code = f"""
import sys, importlib.util
from pathlib import Path
sys.path.append(str(Path('/root/reference/train/train_hybrid.py').parent.parent))      # train_hybrid.py:28
from models.hybrid_ode_nn import HybridODENN                                            # :30
from models.ode_core import ODECore                                                     # :31
from inference.vi import VariationalInference                                           # :32
import models, inference
assert models.__file__.startswith({PKG!r}) and inference.__file__.startswith({PKG!r})
assert sys.modules['inference.vi'].__file__.startswith({PKG!r})
spec = importlib.util.find_spec('inference.mcmc')
assert spec is not None and spec.origin == '/root/reference/inference/mcmc.py', spec
try:
    from inference.mcmc import run_nuts                                                 # :33
    from inference import run_nuts as r2, compute_ess, posterior_summary, save_mcmc_results, load_mcmc_results  # __init__.py:6
    assert r2 is run_nuts and run_nuts.__module__ == 'inference.mcmc'
    print('import block ok (mcmc loaded)')
except ModuleNotFoundError as e:
    assert e.name == 'arviz', e
    print('import block ok (mcmc found; its arviz dependency is absent)')
"""
    r = subprocess.run([sys.executable, "-c", code], cwd="/tmp", env=dict(os.environ, PYTHONPATH=PKG), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "import block ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
