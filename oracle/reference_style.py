"""Reference-STYLE CPU path: how the reference itself integrates, restated.  TEST INFRASTRUCTURE ONLY.

The reference's HybridODENN.forward (reference models/hybrid_ode_nn.py:184-256) is a serial Python loop over patients;
each patient is one `scipy.integrate.solve_ivp` call whose right-hand side callback
  * rounds (t, y) to fp32 torch tensors (:207-208),
  * interpolates every 2-D input linearly on the grid with `np.searchsorted(t_eval, t)` (side='left'), clamped to the
    first / last value outside the grid, and reads 1-D inputs as one constant per patient (:217-231),
  * evaluates ODECore + NNResidual under `torch.no_grad()` on the CPU (:234-235, :108-134) and hands the result back to
    SciPy's fp64 stepper as a NumPy array (:237),
and the trajectory SciPy serves at `t_eval` (dense output, steps NOT broken at the grid) is cast to fp32 (:248).
`'rk45'` -> SciPy `RK45` (the Dormand-Prince 5(4) pair), `'dopri5'` -> `DOP853` (:174-181).

This module does the same thing with this repository's own `models.ODECore` / `models.NNResidual` torch modules (plain
nn.Modules; the product's batched hot path never calls their forward).  It exists for two purposes:
  * `bench.py`'s `cpu_baseline` leg: the reference's per-patient speed on the GPU box's host cores (BASELINE.md section 4,
    SURVEY.md 8d) -- the reference's Python cannot travel to that box, this restatement can;
  * `tests/test_reference_style.py`: a second, SciPy-driven pin of the oracle and of the golden vectors.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
"""
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = os.path.join(_ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

SCIPY_METHOD = {"dopri5": "DOP853", "rk45": "RK45", "dop853": "DOP853", "radau": "Radau", "bdf": "BDF"}


def build_modules(nn_flat, ode_vec, H=64, L=4, activation="relu"):
    """(ODECore, NNResidual) on the CPU carrying the given flat MLP parameters (parameters() order) and 17 constants."""
    import torch
    from models.nn_residual import NNResidual
    from models.ode_core import ODE_PARAM_NAMES, ODECore
    core = ODECore({n: float(v) for n, v in zip(ODE_PARAM_NAMES, np.asarray(ode_vec, np.float64))})
    net = NNResidual(9, H, 6, L, activation=activation)
    flat = torch.as_tensor(np.asarray(nn_flat), dtype=torch.float32)
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(flat[off:off + p.numel()].reshape(p.shape))
            off += p.numel()
    assert off == flat.numel(), (off, flat.numel())
    return core.eval(), net.eval()


def _rhs_callback(core, net, t_eval, inputs, b):
    """The `ode_func` closure of one patient (hybrid_ode_nn.py:206-237)."""
    import torch
    two_d = {k: v for k, v in inputs.items() if v.dim() == 2}
    one_d = {k: v[b] for k, v in inputs.items() if v.dim() != 2}
    n = len(t_eval)

    def f(t, y):
        tt = torch.tensor(t, dtype=torch.float32)
        yy = torch.tensor(y, dtype=torch.float32)
        u = dict(one_d)
        if two_d:
            t32 = tt.numpy()
            i = int(np.searchsorted(t_eval, t32))
            for k, v in two_d.items():
                if i == 0:
                    u[k] = v[b, 0]
                elif i >= n:
                    u[k] = v[b, -1]
                else:
                    lo, hi = t_eval[i - 1], t_eval[i]
                    u[k] = v[b, i - 1] + ((t32 - lo) / (hi - lo)) * (v[b, i] - v[b, i - 1])
        with torch.no_grad():
            d = core(tt, yy, u if inputs else None) + net(tt, yy, yy[..., 3], u.get("tVNS", torch.tensor(0.0)))
        return d.numpy()
    return f


def solve(x0, t, inputs, nn_flat, ode_vec, H=64, L=4, solver="rk45", rtol=1e-6, atol=1e-8, activation="relu",
          modules=None, count=None):
    """Per-patient solve_ivp loop -> y[B, T, 6] fp32 (rows after a failure stay zero, :243-256).
    inputs: {'meal': [B,T] or [B], 'tVNS': ..., 'GD': ...} (numpy); t: [T] or [B,T].  count: optional dict, gets 'nfev'."""
    import torch
    from scipy.integrate import solve_ivp
    core, net = modules or build_modules(nn_flat, ode_vec, H, L, activation)
    x0 = np.asarray(x0, np.float32)
    t = np.asarray(t, np.float32)
    ins = {k: torch.as_tensor(np.asarray(v, np.float32)) for k, v in (inputs or {}).items() if v is not None}
    B, T = x0.shape[0], t.shape[-1]
    y = np.zeros((B, T, 6), np.float32)
    nfev = 0
    for b in range(B):
        te = t[b] if t.ndim == 2 else t
        sol = solve_ivp(_rhs_callback(core, net, te, ins, b), (float(te[0]), float(te[-1])), x0[b], t_eval=te,
                        method=SCIPY_METHOD.get(solver.lower(), solver), rtol=rtol, atol=atol)
        got = sol.y.T.astype(np.float32)
        y[b, :got.shape[0]] = got
        nfev += int(sol.nfev)
    if count is not None:
        count["nfev"] = nfev
    return y


def _pool_worker(args):
    import torch
    torch.set_num_threads(1)
    x0, t, inputs, nn_flat, ode_vec, H, L, solver, rtol, atol = args
    c = {}
    y = solve(x0, t, inputs, nn_flat, ode_vec, H, L, solver, rtol, atol, count=c)
    return y, c["nfev"]


def _pool_warm(_):
    import scipy.integrate  # noqa: F401
    import torch
    torch.set_num_threads(1)
    from models.nn_residual import NNResidual  # noqa: F401
    return os.getpid()


def solve_all_cores(x0, t, inputs, nn_flat, ode_vec, H=64, L=4, solver="rk45", rtol=1e-6, atol=1e-8, procs=None, timed=False):
    """The same loop spread over `procs` worker processes (patients are independent) -> (y, nfev, procs[, seconds]).
    timed=True also returns the wall time of the solves alone: workers are started and have imported torch / scipy before
    the clock starts (a resident trainer would not pay that per batch)."""
    import multiprocessing as mp
    import time
    procs = procs or (os.cpu_count() or 1)
    B = len(x0)
    procs = max(1, min(procs, B))
    cuts = np.linspace(0, B, procs + 1).astype(int)
    jobs = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = {k: (np.asarray(v)[lo:hi]) for k, v in (inputs or {}).items() if v is not None}
        jobs.append((np.asarray(x0)[lo:hi], np.asarray(t)[lo:hi] if np.asarray(t).ndim == 2 else t, part, nn_flat, ode_vec, H, L,
                     solver, rtol, atol))
    with mp.get_context("spawn").Pool(procs) as pool:
        pool.map(_pool_warm, range(4 * procs))
        t0 = time.perf_counter()
        out = pool.map(_pool_worker, jobs, chunksize=1)
        dt = time.perf_counter() - t0
    res = (np.concatenate([o[0] for o in out]), sum(o[1] for o in out), procs)
    return res + (dt,) if timed else res
