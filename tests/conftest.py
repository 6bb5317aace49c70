"""pytest configuration: markers + import paths.

`-m "not gpu"` : oracle vs golden vectors, host logic, C-ABI symbol export (no GPU needed).
`-m gpu`       : parity tests proper -- HIP path (through the C-ABI) vs oracle / goldens.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with gpurun)")


# GPU suite order (the driver runs `pytest -x -m gpu`): the hot path first -- one test per BASELINE.json config
# (test_cfg1..5), then the kernel parity file, the class surface, the plain-C program -- and the "next" rows (data side)
# last, so a failure in a widening row can never hide the evidence for SURVEY section 8's (a) rows.
_FILE_ORDER = ["test_hip_parity.py", "test_models_gpu.py", "test_generic_gpu.py", "test_vi_gpu.py", "test_c_example.py", "test_bench_contract_gpu.py",
               "test_data_side_gpu.py"]


def pytest_collection_modifyitems(config, items):
    def key(pair):
        i, item = pair
        fname = os.path.basename(str(item.fspath))
        name = item.name
        if name.startswith("test_cfg") and name[8:9].isdigit():
            return (0, int(name[8]), i)
        return (1 + (_FILE_ORDER.index(fname) if fname in _FILE_ORDER else len(_FILE_ORDER)), 0, i)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build libhode.so (hipcc cross-compiles for gfx950
    without a GPU) and the oracle once, exactly as __graft_entry__.build() does.  A failing build is reported by the
    tests that need the library (the product path raises when it is missing), not hidden here."""
    import subprocess
    try:
        from hode import _build          # missing OR linked from other sources than the ones in this tree -> make
        _build.ensure(lab=True, jobs=8)  # + hode/lab/libhode_lab.so: the experiment kernels the bitwise tests compare with
    except Exception:
        pass
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libhode_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=False,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def g0():
    import numpy as np
    w = np.load(os.path.join(GOLDEN, "g0_weights_h64_l4.npz"))
    return {"nn": w["nn_flat"], "ode": w["ode"], "H": 64, "L": 4, "raw": w}


@pytest.fixture(scope="session")
def g0_small():
    import numpy as np
    w = np.load(os.path.join(GOLDEN, "g0_weights_h32_l2.npz"))
    return {"nn": w["nn_flat"], "ode": w["ode"], "H": 32, "L": 2}


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
