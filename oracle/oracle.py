"""ORACLE -- test infrastructure only (ctypes wrapper around oracle/hode_oracle.c).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module,
and only as the checker.  The product path (hode/ + csrc/) never imports it.

numpy in, numpy out.  dtype selects the fp32 / fp64 instantiation.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("HODE_ORACLE_LIB") or os.path.join(_HERE, "_build", "libhode_oracle.so")   # (the sanitizer build: make asan)
_lib = None

METHOD_DP54 = 0
METHOD_RK4 = 1


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    src = [os.path.join(_HERE, f) for f in ("hode_oracle.c", "hode_oracle_impl.h", "fourgi_oracle.c")]
    if (not force) and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src):
        return _SO
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.hode_oracle_version.restype = C.c_char_p
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(dtype)


def _arr(a, dtype):
    return None if a is None else np.ascontiguousarray(np.asarray(a), dtype=dtype)


def _mode(a, B, T):
    """hybrid_ode_nn.py:217-231: dim()==2 -> time varying [B,T]; dim()==1 -> constant per patient."""
    if a is None:
        return 0
    if a.ndim == 2:
        assert a.shape == (B, T), (a.shape, B, T)
        return 2
    assert a.shape == (B,), (a.shape, B)
    return 1


ACT_RELU, ACT_TANH, ACT_ELU, ACT_LEAKY_RELU = 0, 1, 2, 3


def layers(L, act=ACT_RELU):
    """The `L` argument of every entry point: hidden layers in bits 0..7, activation code in bits 8..15 (include/hode.h)."""
    return (L & 0xff) | (act << 8)


def n_params(H, L):
    L &= 0xff
    return 9 * H + H + (L - 1) * (H * H + H) + 6 * H + 6


def rhs(x, t, meal, tvns, gd, ode, nn_p, H, L, dtype=np.float32):
    x = _arr(x, dtype)
    B = x.shape[0]
    t, meal, tvns, gd = (_arr(v, dtype) for v in (t, meal, tvns, gd))
    ode, nn_p = _arr(ode, dtype), _arr(nn_p, dtype)
    assert nn_p.size == n_params(H, L)
    out = np.empty((B, 6), dtype)
    rc = getattr(lib(), f"hode_oracle_rhs_{_sfx(dtype)}")(
        C.c_int(B), _p(x), _p(t), _p(meal), _p(tvns), _p(gd), _p(ode), _p(nn_p), C.c_int(H), C.c_int(L), _p(out))
    assert rc == 0
    return out


def rhs_vjp(x, t, meal, tvns, gd, ode, nn_p, H, L, gout, dtype=np.float32):
    x = _arr(x, dtype)
    B = x.shape[0]
    t, meal, tvns, gd, gout = (_arr(v, dtype) for v in (t, meal, tvns, gd, gout))
    ode, nn_p = _arr(ode, dtype), _arr(nn_p, dtype)
    gx = np.zeros((B, 6), dtype)
    gnn = np.zeros(nn_p.size, dtype)
    gode = np.zeros(17, dtype)
    rc = getattr(lib(), f"hode_oracle_rhs_vjp_{_sfx(dtype)}")(
        C.c_int(B), _p(x), _p(t), _p(meal), _p(tvns), _p(gd), _p(ode), _p(nn_p), C.c_int(H), C.c_int(L),
        _p(gout), _p(gx), _p(gnn), _p(gode))
    assert rc == 0
    return gx, gnn, gode


class Solution:
    __slots__ = ("y", "status", "nsteps", "nfev", "tape", "max_steps", "args")


def solve(x0, t, meal, tvns, gd, ode, nn_p, H, L, method=METHOD_DP54, rtol=1e-6, atol=1e-8,
          max_steps=None, dtype=np.float32, want_tape=False):
    """Grid-broken DP5(4) / RK4 -- the integrator the HIP product implements."""
    x0 = _arr(x0, dtype)
    B = x0.shape[0]
    t = _arr(t, dtype)
    T = t.shape[-1]
    t_batched = int(t.ndim == 2)
    meal, tvns, gd = (_arr(v, dtype) for v in (meal, tvns, gd))
    ode, nn_p = _arr(ode, dtype), _arr(nn_p, dtype)
    assert nn_p.size == n_params(H, L)
    if max_steps is None:
        max_steps = 64 * (T - 1) + 64
    s = Solution()
    s.y = np.zeros((B, T, 6), dtype)
    s.status = np.zeros(B, np.int32)
    s.nsteps = np.zeros(B, np.int32)
    s.nfev = np.zeros(B, np.int32)
    s.max_steps = max_steps
    esz = getattr(lib(), f"hode_oracle_tape_entry_size_{_sfx(dtype)}")()
    s.tape = np.zeros(B * max_steps * esz, np.uint8) if want_tape else None
    rc = getattr(lib(), f"hode_oracle_solve_{_sfx(dtype)}")(
        C.c_int(B), C.c_int(T), _p(x0), _p(t), C.c_int(t_batched),
        _p(meal), C.c_int(_mode(meal, B, T)), _p(tvns), C.c_int(_mode(tvns, B, T)),
        _p(gd), C.c_int(_mode(gd, B, T)), _p(ode), _p(nn_p), C.c_int(H), C.c_int(L), C.c_int(method),
        C.c_double(rtol), C.c_double(atol), C.c_int(max_steps), _p(s.y), _p(s.status), _p(s.nsteps),
        _p(s.nfev), _p(s.tape))
    assert rc == 0
    s.args = (t, t_batched, meal, tvns, gd, ode, nn_p, H, L, method, dtype)
    return s


def solve_bwd(sol, gy, want_gnn=True, want_gode=True):
    """Discrete adjoint over the tape of `sol` (which must have been made with want_tape=True)."""
    t, t_batched, meal, tvns, gd, ode, nn_p, H, L, method, dtype = sol.args
    assert sol.tape is not None
    B, T = sol.y.shape[:2]
    gy = _arr(gy, dtype)
    assert gy.shape == (B, T, 6)
    gx0 = np.zeros((B, 6), dtype)
    gnn = np.zeros(nn_p.size, dtype) if want_gnn else None
    gode = np.zeros(17, dtype) if want_gode else None
    rc = getattr(lib(), f"hode_oracle_solve_bwd_{_sfx(dtype)}")(
        C.c_int(B), C.c_int(T), _p(t), C.c_int(t_batched),
        _p(meal), C.c_int(_mode(meal, B, T)), _p(tvns), C.c_int(_mode(tvns, B, T)),
        _p(gd), C.c_int(_mode(gd, B, T)), _p(ode), _p(nn_p), C.c_int(H), C.c_int(L), C.c_int(method),
        C.c_int(sol.max_steps), _p(sol.nsteps), _p(sol.status), _p(sol.tape), _p(gy), _p(gx0), _p(gnn), _p(gode))
    assert rc == 0
    return gx0, gnn, gode


def solve_reference_mode(x0, t, meal, tvns, gd, ode, nn_p, H, L, rtol=1e-6, atol=1e-8):
    """What HybridODENN.forward(solver='rk45') computes (SciPy RK45 + dense output, fp32 RHS)."""
    x0 = _arr(x0, np.float32)
    B = x0.shape[0]
    t = _arr(t, np.float32)
    T = t.shape[-1]
    meal, tvns, gd = (_arr(v, np.float32) for v in (meal, tvns, gd))
    ode, nn_p = _arr(ode, np.float32), _arr(nn_p, np.float32)
    y = np.zeros((B, T, 6), np.float32)
    status = np.zeros(B, np.int32)
    nsteps = np.zeros(B, np.int32)
    nfev = np.zeros(B, np.int32)
    rc = lib().hode_oracle_solve_scipy_rk45(
        C.c_int(B), C.c_int(T), _p(x0), _p(t), C.c_int(int(t.ndim == 2)),
        _p(meal), C.c_int(_mode(meal, B, T)), _p(tvns), C.c_int(_mode(tvns, B, T)),
        _p(gd), C.c_int(_mode(gd, B, T)), _p(ode), _p(nn_p), C.c_int(H), C.c_int(L),
        C.c_double(rtol), C.c_double(atol), _p(y), _p(status), _p(nsteps), _p(nfev))
    assert rc == 0
    return y, status, nsteps, nfev


def dp_tableau():
    Cc, A, Bw, E, P = np.zeros(6), np.zeros((6, 5)), np.zeros(6), np.zeros(7), np.zeros((7, 4))
    lib().hode_oracle_dp_tableau(_p(Cc), _p(A), _p(Bw), _p(E), _p(P))
    return Cc, A, Bw, E, P
