"""ctypes binding of the C ABI declared in include/hode.h.  Tensors in, tensors out (device memory
owned by torch's allocator, work enqueued on torch's current stream)."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("HODE_LIB") or os.path.join(_HERE, "libhode.so")   # HODE_LIB: development builds
_lib = None

METHOD_DP54 = 0
METHOD_RK4 = 1

ST_OK, ST_MAXSTEPS, ST_UNDERFLOW, ST_NONFINITE = 0, 1, 2, 3
_ERR = {-1: "HODE_EINVAL (bad argument)", -2: "HODE_EUNSUPPORTED (shape outside the supported range: H<=128, L<=8, activation code <= 3)",
        -3: "HODE_ELAUNCH (HIP launch failed)"}

# every symbol include/hode.h declares (tests check that the library exports all of them)
SYMBOLS = ["hode_version", "hode_nn_param_count", "hode_tape_bytes", "hode_tape_bytes_hl", "hode_rhs_fwd_f32", "hode_rhs_fwd_f64",
           "hode_rhs_bwd_f32", "hode_rhs_bwd_f64", "hode_solve_fwd_f32", "hode_solve_fwd_f64",
           "hode_solve_bwd_f32", "hode_solve_bwd_f64", "hode_adam_step_f32", "hode_mse_fwd_bwd_f32",
           "hode_selftest_xlane", "hode_4gi_default_params", "hode_4gi_generate_f64", "hode_4gi_rhs_f64",
           "hode_4gi_windows_f32", "hode_4gi_window_moments_f64"]


class HodeError(RuntimeError):
    pass


def lib_path():
    return _SO


def load():
    """Load libhode.so.  Raises (loudly) when it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise HodeError(f"{_SO} not found: build it with `python __graft_entry__.py` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the hot path")
        if not os.environ.get("HODE_LIB") and not os.environ.get("HODE_ALLOW_STALE"):
            from . import _build
            if not _build.is_current():
                raise HodeError(f"{_SO} was not built from the kernel sources next to it (binary {_build.binary_stamp()}, sources "
                                f"{_build.source_stamp()}): rebuild with `python __graft_entry__.py` (or `make -C {_build.CSRC}`); "
                                "HODE_ALLOW_STALE=1 loads it anyway")
        if os.environ.get("HODE_LIB") and not os.environ.get("HODE_ALLOW_STALE") and os.path.exists(_SO + ".srcsha"):
            # a development library that carries a source stamp (make lab writes one) must have been built from THESE sources:
            # the bitwise lab-versus-product tests would otherwise compare against last week's kernels without saying so
            from . import _build
            want = _build.lab_source_stamp()
            if want is not None and os.path.abspath(_SO) == os.path.abspath(_build.LAB_SO) and _build.binary_stamp(_SO) != want:
                raise HodeError(f"{_SO} is stale (binary {_build.binary_stamp(_SO)}, sources {want}): `make -C {_build.CSRC} lab`; "
                                "HODE_ALLOW_STALE=1 loads it anyway")
        _lib = C.CDLL(_SO)
        _lib.hode_version.restype = C.c_char_p
        _lib.hode_tape_bytes.restype = C.c_size_t
        _lib.hode_tape_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
        _lib.hode_tape_bytes_hl.restype = C.c_size_t
        _lib.hode_tape_bytes_hl.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    return _lib


def version():
    return load().hode_version().decode()


ACT_RELU, ACT_TANH, ACT_ELU, ACT_LEAKY_RELU = 0, 1, 2, 3
NN_SHARED = 1 << 16          # HODE_LAYERS_NN_SHARED (hode_solve_fwd_* only)


def layers(L, act=ACT_RELU):
    """The `L` argument of the C ABI: hidden layers in bits 0..7, activation code in bits 8..15 (include/hode.h HODE_LAYERS)."""
    return (L & 0xff) | (act << 8)


def n_params(H, L):
    L &= 0xff                                  # (L may carry an activation code)
    return 9 * H + H + (L - 1) * (H * H + H) + 6 * H + 6


def _check(rc, what):
    if rc != 0:
        raise HodeError(f"{what} failed: {_ERR.get(rc, rc)}")


def _need_gpu(t):
    if not t.is_cuda:
        raise HodeError("hode kernels need tensors on a HIP device (no CPU fallback)")


def _ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _sfx(dtype):
    if dtype == torch.float32:
        return "f32"
    if dtype == torch.float64:
        return "f64"
    raise HodeError(f"unsupported dtype {dtype}")


def _prep(t, dtype, dev):
    return None if t is None else t.to(device=dev, dtype=dtype).contiguous()


def _mode(t, B, T):
    """models/hybrid_ode_nn.py:217-231: dim()==2 time varying [B,T]; dim()==1 constant [B]."""
    if t is None:
        return 0
    if t.dim() == 2:
        if tuple(t.shape) != (B, T):
            raise HodeError(f"time-varying input must be [B,T]=({B},{T}), got {tuple(t.shape)}")
        return 2
    if tuple(t.shape) != (B,):
        raise HodeError(f"constant input must be [B]=({B},), got {tuple(t.shape)}")
    return 1


def selftest_xlane(device="cuda"):
    out = torch.zeros(64 * 12, dtype=torch.int32, device=device)
    _check(load().hode_selftest_xlane(_stream(), _ptr(out)), "hode_selftest_xlane")
    return out.view(64, 12)


def rhs_fwd(x, t, meal, tvns, gd, ode_p, nn_p, H, L):
    _need_gpu(x)
    dt, dev = x.dtype, x.device
    x = x.contiguous()
    B = x.shape[0]
    t, meal, tvns, gd = (_prep(v, dt, dev) for v in (t, meal, tvns, gd))
    ode_p, nn_p = _prep(ode_p, dt, dev), _prep(nn_p, dt, dev)
    assert nn_p.numel() == n_params(H, L) and ode_p.numel() == 17
    out = torch.empty(B, 6, dtype=dt, device=dev)
    fn = getattr(load(), f"hode_rhs_fwd_{_sfx(dt)}")
    _check(fn(_stream(), C.c_int(B), _ptr(x), _ptr(t), _ptr(meal), _ptr(tvns), _ptr(gd), _ptr(ode_p), _ptr(nn_p),
              C.c_int(H), C.c_int(L), _ptr(out)), "hode_rhs_fwd")
    return out


class Solve:
    """Result of solve_fwd: trajectories + per-trajectory diagnostics (+ tape for the adjoint)."""
    __slots__ = ("y", "status", "nsteps", "nfev", "tape", "max_steps", "ctx")


def tape_nbytes(B, max_steps, elem_size, L, H=64):
    return load().hode_tape_bytes_hl(B, max_steps, elem_size, H, L)


def solve_fwd(x0, t, meal, tvns, gd, ode_p, nn_p, H, L, method=METHOD_DP54, rtol=1e-6, atol=1e-8,
              max_steps=None, n_sets=1, want_tape=False, tape=None, nn_shared=False):
    """Forward solve.  want_tape=True records what the adjoint needs; pass `tape=` (a uint8 tensor of at least
    tape_nbytes(...) bytes, e.g. the `.tape` of an earlier solution of the same shape) to reuse the buffer
    instead of allocating several GB per call.
    nn_shared=True: `nn_p` is ONE network used by all n_sets sets of ODE constants (HODE_LAYERS_NN_SHARED; forward only)."""
    _need_gpu(x0)
    dt, dev = x0.dtype, x0.device
    x0 = x0.contiguous()
    B = x0.shape[0]
    t = _prep(t, dt, dev)
    T = t.shape[-1]
    t_batched = int(t.dim() == 2)
    if t_batched and t.shape[0] != B:
        raise HodeError("batched time grid must be [B,T]")
    meal, tvns, gd = (_prep(v, dt, dev) for v in (meal, tvns, gd))
    ode_p, nn_p = _prep(ode_p, dt, dev), _prep(nn_p, dt, dev)
    if nn_p.numel() != (1 if nn_shared else n_sets) * n_params(H, L) or ode_p.numel() != 17 * n_sets:
        raise HodeError("parameter vector size does not match (H, L, n_sets)")
    if nn_shared and (want_tape or tape is not None):
        raise HodeError("nn_shared is a forward-only option (the adjoint returns one gradient row per parameter set)")
    if max_steps is None:
        # accepted-step budget per trajectory.  With a tape every step costs 6*(L+1)*256 B of HBM (stage
        # tape), so the default is tighter there; a trajectory that needs more reports status 1.
        if method == METHOD_RK4:
            max_steps = max(T - 1, 1)
        elif want_tape or tape is not None:
            max_steps = (T - 1) + max(32, (T - 1) // 4)
        else:
            max_steps = 8 * (T - 1) + 64
    s = Solve()
    s.y = torch.empty(B, T, 6, dtype=dt, device=dev)
    s.status = torch.empty(B, dtype=torch.int32, device=dev)
    s.nsteps = torch.empty(B, dtype=torch.int32, device=dev)
    s.nfev = torch.empty(B, dtype=torch.int32, device=dev)
    s.max_steps = max_steps
    s.tape = None
    if want_tape or tape is not None:
        nbytes = load().hode_tape_bytes_hl(B, max_steps, x0.element_size(), H, L)
        if tape is not None:
            if tape.dtype != torch.uint8 or not tape.is_cuda or tape.numel() < nbytes or tape.data_ptr() % 256:
                raise HodeError(f"tape buffer must be a 256-byte aligned uint8 device tensor of >= {nbytes} bytes")
            s.tape = tape
        else:
            s.tape = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fn = getattr(load(), f"hode_solve_fwd_{_sfx(dt)}")
    rc = fn(_stream(), C.c_int(B), C.c_int(T), _ptr(x0), _ptr(t), C.c_int(t_batched),
            _ptr(meal), C.c_int(_mode(meal, B, T)), _ptr(tvns), C.c_int(_mode(tvns, B, T)),
            _ptr(gd), C.c_int(_mode(gd, B, T)), _ptr(ode_p), _ptr(nn_p), C.c_int(n_sets), C.c_int(H),
            C.c_int(L | (NN_SHARED if nn_shared else 0)),
            C.c_int(method), C.c_double(rtol), C.c_double(atol), C.c_int(max_steps), _ptr(s.y), _ptr(s.status),
            _ptr(s.nsteps), _ptr(s.nfev), _ptr(s.tape))
    _check(rc, "hode_solve_fwd")
    s.ctx = (t, t_batched, meal, tvns, gd, ode_p, nn_p, n_sets, H, L, method)
    return s


def solve_bwd(sol, gy, want_gnn=True, want_gode=False):
    """Reverse-time discrete adjoint over the tape of `sol` (made with want_tape=True).
    Returns (gx0[B,6], gnn[n_sets*P] or None, gode[n_sets*17] or None)."""
    if sol.tape is None:
        raise HodeError("solve_bwd needs a solution computed with want_tape=True")
    t, t_batched, meal, tvns, gd, ode_p, nn_p, n_sets, H, L, method = sol.ctx
    dt, dev = sol.y.dtype, sol.y.device
    B, T = sol.y.shape[:2]
    gy = gy.to(device=dev, dtype=dt).contiguous()
    if tuple(gy.shape) != (B, T, 6):
        raise HodeError("gy must be [B,T,6]")
    gx0 = torch.empty(B, 6, dtype=dt, device=dev)
    gnn = torch.zeros(nn_p.numel(), dtype=dt, device=dev) if want_gnn else None
    gode = torch.zeros(17 * n_sets, dtype=dt, device=dev) if want_gode else None
    fn = getattr(load(), f"hode_solve_bwd_{_sfx(dt)}")
    rc = fn(_stream(), C.c_int(B), C.c_int(T), _ptr(t), C.c_int(t_batched),
            _ptr(meal), C.c_int(_mode(meal, B, T)), _ptr(tvns), C.c_int(_mode(tvns, B, T)),
            _ptr(gd), C.c_int(_mode(gd, B, T)), _ptr(ode_p), _ptr(nn_p), C.c_int(n_sets), C.c_int(H), C.c_int(L),
            C.c_int(method), C.c_int(sol.max_steps), _ptr(sol.nsteps), _ptr(sol.status), _ptr(sol.tape), _ptr(gy),
            _ptr(gx0), _ptr(gnn), _ptr(gode))
    _check(rc, "hode_solve_bwd")
    return gx0, gnn, gode


def rhs_bwd(x, t, meal, tvns, gd, ode_p, nn_p, H, L, gout, want_gt=False, want_gnn=True, want_gode=False):
    _need_gpu(x)
    dt, dev = x.dtype, x.device
    x = x.contiguous()
    B = x.shape[0]
    t, meal, tvns, gd, gout = (_prep(v, dt, dev) for v in (t, meal, tvns, gd, gout))
    ode_p, nn_p = _prep(ode_p, dt, dev), _prep(nn_p, dt, dev)
    gx = torch.empty(B, 6, dtype=dt, device=dev)
    gt = torch.empty(B, dtype=dt, device=dev) if want_gt else None
    gnn = torch.zeros(nn_p.numel(), dtype=dt, device=dev) if want_gnn else None
    gode = torch.zeros(17, dtype=dt, device=dev) if want_gode else None
    fn = getattr(load(), f"hode_rhs_bwd_{_sfx(dt)}")
    _check(fn(_stream(), C.c_int(B), _ptr(x), _ptr(t), _ptr(meal), _ptr(tvns), _ptr(gd), _ptr(ode_p), _ptr(nn_p),
              C.c_int(H), C.c_int(L), _ptr(gout), _ptr(gx), _ptr(gt), _ptr(gnn), _ptr(gode)), "hode_rhs_bwd")
    return gx, gt, gnn, gode


def adam_step(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-8, step=1, max_norm=0.0, grad_scale=1.0,
              weight_decay=0.0, scratch=None):
    """In-place fused clip + Adam on flat fp32 vectors (train/train_hybrid.py:255-261)."""
    for a in (p, g, m, v):
        _need_gpu(a)
        if a.dtype != torch.float32 or not a.is_contiguous():
            raise HodeError("adam_step needs contiguous fp32 tensors")
    if scratch is None:
        scratch = torch.empty(2, dtype=torch.float32, device=p.device)
    _check(load().hode_adam_step_f32(_stream(), C.c_int64(p.numel()), _ptr(p), _ptr(g), _ptr(m), _ptr(v),
                                     C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_int(step),
                                     C.c_float(max_norm), C.c_float(grad_scale), C.c_float(weight_decay),
                                     _ptr(scratch)), "hode_adam_step")
    return scratch


def mse_fwd_bwd(y, obs, scale, loss_sum=None, want_grad=True):
    """sum((y-obs)^2) accumulated into loss_sum (fp64[1]) and gy = 2*scale*(y-obs) in one pass."""
    _need_gpu(y)
    y = y.contiguous()
    obs = obs.to(device=y.device, dtype=torch.float32).contiguous()
    if y.dtype != torch.float32 or y.shape != obs.shape:
        raise HodeError("mse_fwd_bwd needs fp32 tensors of equal shape")
    if loss_sum is None:
        loss_sum = torch.zeros(1, dtype=torch.float64, device=y.device)
    gy = torch.empty_like(y) if want_grad else None
    _check(load().hode_mse_fwd_bwd_f32(_stream(), C.c_int64(y.numel()), _ptr(y), _ptr(obs), C.c_float(scale),
                                       _ptr(loss_sum), _ptr(gy)), "hode_mse_fwd_bwd")
    return loss_sum, gy


# --------------------------------------------------------------------------------------------- data side
FOURGI_NPAR = 26
FOURGI_SCRATCH_BYTES = 98304
FOURGI_COLUMNS = ["subject_id", "time_hours", "time_minutes", "glucose_mmol_L", "insulin_pmol_L", "glp1_pmol_L",
                  "glucagon_pmol_L", "gip_pmol_L", "meal_indicator"]


def fourgi_default_params(patient_type="T2DM"):
    """The reference's parameter set (data/generate4GI.py:15-64) as a list of 26 floats (host call, no GPU needed)."""
    buf = (C.c_double * FOURGI_NPAR)()
    _check(load().hode_4gi_default_params(C.c_int(_ptype(patient_type)), buf), "hode_4gi_default_params")
    return list(buf)


def _ptype(patient_type):
    if patient_type in ("T2DM", 0):
        return 0
    if patient_type in ("HV", 1):
        return 1
    raise HodeError(f"patient_type must be 'T2DM' or 'HV', got {patient_type!r}")


def _par_host(par):
    if par is None:
        return None
    if len(par) != FOURGI_NPAR:
        raise HodeError(f"par must hold {FOURGI_NPAR} values")
    return (C.c_double * FOURGI_NPAR)(*[float(v) for v in par])


def fourgi_generate(bsl, T, interval_min, meal_time, meal_size, patient_type="T2DM", par=None, z=None, noise_cv=0.0,
                    subject0=0, rtol=1e-10, atol=1e-12, max_steps=100000, z_tcb=None):
    """K7.  bsl[B,5] (device, fp64) -> (table[B*T,9] fp64, status[B] int32).  meal_time/meal_size: [n] or [B,n].
    Noise draws: z[B,5,T] (numpy-stream order, transposed here) or z_tcb[T,5,B] (the kernel's layout, used as is)."""
    _need_gpu(bsl)
    dev = bsl.device
    bsl = bsl.to(torch.float64).contiguous().view(-1, 5)
    B = bsl.shape[0]
    mt = torch.as_tensor(meal_time, dtype=torch.float64, device=dev).contiguous()
    ms = torch.as_tensor(meal_size, dtype=torch.float64, device=dev).contiguous()
    if mt.shape != ms.shape or mt.dim() not in (1, 2) or (mt.dim() == 2 and mt.shape[0] != B):
        raise HodeError(f"meal_time / meal_size must both be [n_meals] or [B,n_meals], got {tuple(mt.shape)} / {tuple(ms.shape)}")
    n_meals = mt.shape[-1]
    if z is not None:
        # z[B,5,T] is the order FourGIModel.generate_dataset consumes numpy's stream in; the kernel wants [T,5,B]
        if tuple(z.shape) != (B, 5, T):
            raise HodeError(f"z must be [B,5,T]=({B},5,{T}), got {tuple(z.shape)}")
        z = z.to(device=dev, dtype=torch.float64).permute(2, 1, 0).contiguous()
    elif z_tcb is not None:
        if tuple(z_tcb.shape) != (T, 5, B):
            raise HodeError(f"z_tcb must be [T,5,B]=({T},5,{B}), got {tuple(z_tcb.shape)}")
        z = z_tcb.to(device=dev, dtype=torch.float64).contiguous()
    table = torch.empty(B * T, 9, dtype=torch.float64, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    _check(load().hode_4gi_generate_f64(
        _stream(), C.c_int(B), C.c_int(T), C.c_double(interval_min), C.c_int(_ptype(patient_type)), _par_host(par),
        _ptr(bsl), C.c_int(n_meals), _ptr(mt if n_meals else None), _ptr(ms if n_meals else None), C.c_int(int(mt.dim() == 2)),
        _ptr(z), C.c_double(noise_cv), C.c_int64(subject0), C.c_double(rtol), C.c_double(atol), C.c_int(max_steps),
        _ptr(table), _ptr(status)), "hode_4gi_generate_f64")
    return table, status


def fourgi_rhs(bsl, y, meal, patient_type="T2DM", par=None):
    _need_gpu(y)
    dev = y.device
    y = y.to(torch.float64).contiguous().view(-1, 8)
    B = y.shape[0]
    bsl = bsl.to(device=dev, dtype=torch.float64).contiguous().view(-1, 5)
    meal = meal.to(device=dev, dtype=torch.float64).contiguous().view(-1)
    assert bsl.shape[0] == B and meal.shape[0] == B
    d = torch.empty_like(y)
    _check(load().hode_4gi_rhs_f64(_stream(), C.c_int(B), C.c_int(_ptype(patient_type)), _par_host(par), _ptr(bsl), _ptr(y),
                                   _ptr(meal), _ptr(d)), "hode_4gi_rhs_f64")
    return d


def fourgi_window_moments(table, cols, row0, seq_len, check_bounds=True):
    """Mergeable statistics of one shard's windows -> tensor[13] = {count, mean[6], M2[6]} on the device."""
    _need_gpu(table)
    dev = table.device
    if table.dtype != torch.float64 or table.dim() != 2:
        raise HodeError("table must be a 2-D float64 tensor")
    table = table.contiguous()
    row0 = row0.to(device=dev, dtype=torch.int64).contiguous()
    N, S = row0.numel(), int(seq_len)
    if check_bounds and N and (int(row0.min()) < 0 or int(row0.max()) + S > table.shape[0]):
        raise HodeError("window outside the table")
    mom = torch.empty(13, dtype=torch.float64, device=dev)
    scratch = torch.empty(FOURGI_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    g = lambda k: C.c_int(int(cols.get(k, -1)))
    _check(load().hode_4gi_window_moments_f64(_stream(), _ptr(table), C.c_int(table.shape[1]), g("glucose"), g("insulin"),
                                              g("glucagon"), g("glp1"), g("ge"), g("ffa"), _ptr(row0), C.c_int64(N),
                                              C.c_int64(S), _ptr(mom), _ptr(scratch)), "hode_4gi_window_moments_f64")
    return mom


def combine_moments(moments):
    """moments[R,13] (one row per shard, rank order) -> (mean[6], std[6]) float64 CPU tensors: Chan's pairwise merge,
    applied in rank order so that every rank gets the same bits; std = sqrt(M2 / n) + 1e-6 (train_hybrid.py:124-127)."""
    m = torch.as_tensor(moments, dtype=torch.float64).cpu().reshape(-1, 13)
    n, mean, m2 = 0.0, torch.zeros(6, dtype=torch.float64), torch.zeros(6, dtype=torch.float64)
    for r in range(m.shape[0]):
        nb = float(m[r, 0])
        if nb == 0:
            continue
        d = m[r, 1:7] - mean
        tot = n + nb
        m2 = m2 + m[r, 7:13] + d * d * (n * nb / tot)
        mean = mean + d * (nb / tot)
        n = tot
    if n == 0:
        return torch.zeros(6, dtype=torch.float64), torch.ones(6, dtype=torch.float64)
    return mean, torch.sqrt(m2 / n) + 1e-6


def fourgi_windows(table, cols, time_div, row0, seq_len, normalize=True, mean_std=None, check_bounds=True):
    """K8.  table[rows,ncols] fp64 (device); cols: dict time/glucose/insulin/glucagon/glp1 (+ optional ge/ffa/meal/tvns)
    -> column index; row0[N] int64 (device).  -> states[N,S,6], meal[N,S], tvns[N,S], time[N,S] (fp32), mean_std[12] (fp64)."""
    _need_gpu(table)
    dev = table.device
    if table.dtype != torch.float64 or table.dim() != 2:
        raise HodeError("table must be a 2-D float64 tensor")
    table = table.contiguous()
    row0 = row0.to(device=dev, dtype=torch.int64).contiguous()
    N, S = row0.numel(), int(seq_len)
    # checked on the host BEFORE the launch (two device reductions + a sync); callers that built row0 from the table's
    # own subject ranges (GlucoseDataset) pass check_bounds=False
    if check_bounds and N and (int(row0.min()) < 0 or int(row0.max()) + S > table.shape[0]):
        raise HodeError("window outside the table")
    states = torch.empty(N, S, 6, dtype=torch.float32, device=dev)
    meal, tvns, time = (torch.empty(N, S, dtype=torch.float32, device=dev) for _ in range(3))
    if mean_std is not None:                            # statistics given (HODE_4GI_NORM_GIVEN)
        mean_std = torch.as_tensor(mean_std, dtype=torch.float64).to(dev).contiguous().clone()
        if mean_std.numel() != 12:
            raise HodeError("mean_std must hold 6 means and 6 stds")
        normalize = 2
    else:
        mean_std = torch.empty(12, dtype=torch.float64, device=dev)
        normalize = int(bool(normalize))
    scratch = torch.empty(FOURGI_SCRATCH_BYTES, dtype=torch.uint8, device=dev)
    g = lambda k: C.c_int(int(cols.get(k, -1)))
    _check(load().hode_4gi_windows_f32(
        _stream(), _ptr(table), C.c_int(table.shape[1]), g("time"), C.c_double(time_div), g("glucose"), g("insulin"),
        g("glucagon"), g("glp1"), g("ge"), g("ffa"), g("meal"), g("tvns"), _ptr(row0), C.c_int64(N), C.c_int64(S),
        C.c_int(normalize), _ptr(states), _ptr(meal), _ptr(tvns), _ptr(time), _ptr(mean_std), _ptr(scratch)),
        "hode_4gi_windows_f32")
    return states, meal, tvns, time, mean_std
