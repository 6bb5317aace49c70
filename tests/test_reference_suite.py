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
