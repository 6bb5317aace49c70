"""hode -- Python binding of libhode.so (HIP kernels for gfx950 behind a C ABI, include/hode.h).

PyTorch is plumbing here: device memory, streams, torch.distributed.  All arithmetic of the hot
path (RHS, DP5(4) stepping, adjoint, Adam) runs in the hand-written HIP kernels; there is NO
CPU / eager fallback -- if the library or a GPU is missing the calls raise.
"""
from . import _build as build_info  # noqa: F401
from . import _capi as capi  # noqa: F401
from ._capi import (METHOD_DP54, METHOD_RK4, HodeError, adam_step, lib_path, load, mse_fwd_bwd, n_params,  # noqa: F401
                    rhs_bwd, rhs_fwd, selftest_xlane, solve_bwd, solve_fwd, version)
from . import train  # noqa: F401,E402
from . import datagen  # noqa: F401,E402
