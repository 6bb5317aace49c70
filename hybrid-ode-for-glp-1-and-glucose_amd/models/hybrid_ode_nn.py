"""HybridODENN -- dx/dt = f_physio(t, x; theta) + g_NN(t, x, GLP1, tVNS; phi), integrated on MI355X.

Mirror of reference models/hybrid_ode_nn.py (class surface :22-68, ode_residual :108-134,
forward :136-261, loss :263-351, VI hooks :353-438).  Same constructor, method names, argument
meaning, output shapes/dtypes, state_dict keys and error conventions; the arithmetic of the hot
path runs in the HIP kernels of libhode.so:

    ode_residual  -> K1 (+ K5 under autograd)              csrc/hode_rhs.hip, hode_solve_bwd.hip
    forward       -> K2+K3 batched DP5(4)/RK4 solve         csrc/hode_solve_fwd.hip
                     (+ K4 reverse-time adjoint under autograd; the reference detaches here)
    loss          -> solve + fused MSE + ONE batched launch for all physics points

Compute always happens on the HIP device.  `self.device` (which may be 'cpu', as the reference's
tests pass) only decides where outputs are placed; tensors on other devices are staged over.
Without a GPU or without libhode.so these methods raise -- there is no CPU fallback.
"""
import logging
from typing import Dict, List, Optional, Tuple, Union

import torch
import torch.nn as nn

import hode

from .bayes import VariationalParameters, bayes_loss
from .nn_residual import NNResidual
from .ode_core import ODE_PARAM_NAMES, ODECore

logger = logging.getLogger(__name__)

# reference hybrid_ode_nn.py:174-181 maps these names onto SciPy methods ('dopri5' silently means
# DOP853 there).  Every adaptive name runs the DP5(4) kernel here: all of them converge to the same
# solution, and DP5(4) is what 'dopri5'/'rk45' name.  'rk4' = fixed step, one step per interval.
_SOLVERS = {"dopri5": hode.METHOD_DP54, "rk45": hode.METHOD_DP54, "dop853": hode.METHOD_DP54,
            "radau": hode.METHOD_DP54, "bdf": hode.METHOD_DP54, "rk4": hode.METHOD_RK4}


def _compute_device() -> torch.device:
    if not torch.cuda.is_available():
        raise hode.HodeError("HybridODENN needs a HIP device: the solve/adjoint path has no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


class _RhsFn(torch.autograd.Function):
    """ode_residual on the device: K1 forward, K5 backward (grads w.r.t. x, t, MLP and ODE constants)."""

    @staticmethod
    def forward(ctx, x, t, nn_flat, ode_vec, meal, tvns, gd, H, L):
        out = hode.rhs_fwd(x, t, meal, tvns, gd, ode_vec, nn_flat, H, L)
        ctx.save_for_backward(x, t, nn_flat, ode_vec, meal if meal is not None else x.new_empty(0),
                              tvns if tvns is not None else x.new_empty(0), gd if gd is not None else x.new_empty(0))
        ctx.cfg = (H, L, meal is not None, tvns is not None, gd is not None)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        x, t, nn_flat, ode_vec, meal, tvns, gd = ctx.saved_tensors
        H, L, has_m, has_v, has_g = ctx.cfg
        need = ctx.needs_input_grad
        gx, gt, gnn, gode = hode.rhs_bwd(x, t, meal if has_m else None, tvns if has_v else None,
                                         gd if has_g else None, ode_vec, nn_flat, H, L, gout.contiguous(),
                                         want_gt=need[1], want_gnn=need[2], want_gode=need[3])
        return gx, gt, gnn, gode, None, None, None, None, None


# Stage-tape budget of one autograd solve.  The adjoint reads 6*(L*256 + 32) B per accepted step (1.9 MB per trajectory
# for the 4x64 network at T = 241): 4 096 patients need 7.3 GiB, BASELINE config 5's 8 192 patients x 16 VI samples per
# GPU would need 233 GiB.  Above the budget the forward runs WITHOUT a tape and the backward re-integrates the batch in
# chunks that fit (forward-with-tape + adjoint per chunk, one tape buffer re-used): bounded memory for 1.3x the time.
TAPE_BUDGET_BYTES = 64 << 30


_budget_cache: Dict[Tuple[int, int], list] = {}
_span_cache: Dict[torch.device, torch.Tensor] = {}


def _short_span(dev):
    """[0, 0.1] on the device (the physics term's finite-difference step, hybrid_ode_nn.py:318) -- made once."""
    if dev not in _span_cache:
        _span_cache[dev] = torch.tensor([0.0, 0.1], dtype=torch.float32, device=dev)
    return _span_cache[dev]


def _tape_budget(dev, need: int = 0):
    """min(TAPE_BUDGET_BYTES, 80 % of what the allocator can still hand out on this device) -- no synchronisation.
    Asking the allocator costs ~0.25 ms of host time (memory_stats walks a dictionary), as much as a whole small-batch
    solve: a caller that passes the bytes it `need`s gets the last answer back while that is at least 4x the need."""
    d = torch.device(dev)
    key = (d.index if d.index is not None else torch.cuda.current_device(), TAPE_BUDGET_BYTES)
    hit = _budget_cache.get(key)
    if need > 0 and hit is not None and 4 * need <= hit[0] and hit[1] > 0:
        hit[1] -= 1                               # an answer is good for 64 uses: free memory moves (other tensors, a second model)
        return hit[0]
    free, _ = torch.cuda.mem_get_info(dev)
    reusable = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)     # cached blocks torch can re-use
    _budget_cache[key] = [max(1, min(TAPE_BUDGET_BYTES, int(0.8 * (free + reusable)))), 64]
    return _budget_cache[key][0]


# Accepted-step budget of a solve that records a tape: every step costs 6*(L*256 + 32) + 36 B of tape, so the default is
# T-1 (one step per interval, what the benchmark regime takes) plus a margin.  `TAPE_STEP_MARGIN` / the `tape_steps`
# attribute of a model change it; a trajectory that STILL needs more steps (stiff or z-scored data) is not lost: it is
# integrated again with the no-grad budget in a second, small launch (_solve_taped), so a solve under autograd returns
# the same status as the same solve under torch.no_grad().
TAPE_STEP_MARGIN = (32, 4)        # steps = (T-1) + max(margin[0], (T-1) // margin[1])


# A taped solve whose WHOLE tape at the no-grad step budget (8 (T-1) + 64 steps) stays below this is given that budget straight
# away: no trajectory can run out of tape steps, so there is nothing to retry and -- what matters at the reference's batch sizes
# (32 windows x 61 points: 110 MB) -- no host synchronisation between the forward solve and the adjoint: the host queues the whole
# optimisation step ahead of the device instead of waiting ~0.35 ms for the forward in every step (tools/prof_class_step.py).
SMALL_TAPE_BYTES = 2 << 30


def _small_tape_steps(n_traj, T, method, elem, L, H, override):
    """The no-grad step budget when the tape of `n_traj` trajectories at that budget is small (SMALL_TAPE_BYTES), else None."""
    if override is not None or method == hode.METHOD_RK4:
        return None
    big = _eval_steps(T, method)
    return big if n_traj * hode.capi.tape_nbytes(1, big, elem, L, H) <= SMALL_TAPE_BYTES else None


class _FlatNN(torch.autograd.Function):
    """The flat parameter vector the kernels take, WITHOUT the torch.cat of ten tensors per step: the ten parameters are views
    of one buffer (HybridODENN._nn_flat_store), forward hands that buffer out, backward cuts the gradient into ten views."""

    @staticmethod
    def forward(ctx, flat, *params):
        ctx.shapes = [p.shape for p in params]
        return flat.view(-1)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        outs, off = [], 0
        for shp in ctx.shapes:
            n = shp.numel()
            outs.append(g[off:off + n].view(shp))
            off += n
        return (None, *outs)


def _tape_steps(T, method, override=None):
    if method == hode.METHOD_RK4:
        return max(T - 1, 1)
    if override is not None:
        return max(int(override), 1)
    return (T - 1) + max(TAPE_STEP_MARGIN[0], (T - 1) // TAPE_STEP_MARGIN[1])


def _eval_steps(T, method):
    return max(T - 1, 1) if method == hode.METHOD_RK4 else 8 * (T - 1) + 64


class _Taped:
    """A forward solve with tape + the trajectories that ran out of tape steps and were integrated again with the
    no-grad budget (`extras`: [(indices, set id, solve or None, args)]).  y / status / nsteps / nfev are the merged results.
    An extra whose tape fitted the budget keeps it (solve != None); the others were integrated without a tape and are
    re-integrated chunk by chunk in backward(), one tape buffer re-used -- the retries are bounded by the same budget as
    everything else (4 096 retried trajectories at 12.6 MB each would otherwise ask for 51 GB on top of the tape)."""

    def __init__(self, sol, extras, worst=None):
        self.sol, self.extras = sol, extras
        self.worst = worst                   # max status of the main launch when the host already knows it (0: all fine)
        self.y, self.tape = sol.y, sol.tape
        self.status, self.nsteps, self.nfev = sol.status, sol.nsteps, sol.nfev
        if extras:
            # merged bookkeeping lives in COPIES: the adjoint of the main launch walks ITS tape with ITS step counts
            self.status, self.nsteps, self.nfev = sol.status.clone(), sol.nsteps.clone(), sol.nfev.clone()
            for idx, _, s2, _ in extras:
                self.y[idx], self.status[idx], self.nsteps[idx], self.nfev[idx] = s2.y, s2.status, s2.nsteps, s2.nfev
        self.n_retried = sum(int(e[0].numel()) for e in extras)

    def backward(self, gy, want_gnn=True, want_gode=False):
        gy = gy.contiguous()
        subs = []
        if self.extras:
            gy = gy.clone()
            for idx, *_ in self.extras:
                subs.append(gy[idx].contiguous())
                gy[idx] = 0                     # the truncated copy in the main launch contributes nothing
        gx0, gnn, gode = hode.solve_bwd(self.sol, gy, want_gnn=want_gnn, want_gode=want_gode)
        for (idx, set_id, sol2, lazy), g2 in zip(self.extras, subs):
            if sol2.tape is not None:
                g0, gn, go = hode.solve_bwd(sol2, g2, want_gnn=want_gnn, want_gode=want_gode)
                self._add(gx0, gnn, gode, idx, set_id, g0, gn, go)
                continue
            solve, cap = lazy                    # re-integrate with a tape, `cap` trajectories at a time, one buffer
            tape = None
            for lo in range(0, idx.numel(), cap):
                sl = slice(lo, min(lo + cap, idx.numel()))
                sp = solve(sl, tape)
                tape = sp.tape
                g0, gn, go = hode.solve_bwd(sp, g2[sl].contiguous(), want_gnn=want_gnn, want_gode=want_gode)
                self._add(gx0, gnn, gode, idx[sl], set_id, g0, gn, go)
        return gx0, gnn, gode

    @staticmethod
    def _add(gx0, gnn, gode, idx, set_id, g0, gn, go):
        gx0[idx] = g0
        if gn is not None:
            P = gn.numel()
            gnn[P * set_id:P * (set_id + 1)] += gn
        if go is not None:
            gode[17 * set_id:17 * (set_id + 1)] += go


def _solve_taped(x0, t, meal, tvns, gd, ode_vec, nn_flat, H, L, method, rtol, atol, n_sets, steps, tape=None):
    """Forward solve with a tape of `steps` accepted steps per trajectory; status-1 trajectories (budget exhausted) are
    solved again, set by set, with the inference budget and their rows replace the truncated ones.  One host
    synchronisation (the failure count); no second launch in the common case.  The retries' tapes count against the tape
    budget: what does not fit is integrated without a tape now and re-integrated piecewise in the backward (_Taped)."""
    sol = hode.solve_fwd(x0, t, meal, tvns, gd, ode_vec, nn_flat, H, L, method=method, rtol=rtol, atol=atol, n_sets=n_sets,
                         want_tape=tape is None, tape=tape, max_steps=steps)
    extras = []
    big = _eval_steps(t.shape[-1], method)
    worst = None
    if big > steps:
        worst = int(sol.status.max())        # the one host synchronisation: 0 in the common case, nothing else to look at
        bad = torch.nonzero(sol.status == 1).flatten() if worst else sol.status[:0]
        if bad.numel():
            per_set, P = x0.shape[0] // n_sets, nn_flat.numel() // n_sets
            cut = lambda v, i: None if v is None else v[i].contiguous()          # noqa: E731
            per_traj = hode.capi.tape_nbytes(1, big, x0.element_size(), L, H)
            left = _tape_budget(x0.device)
            for set_id in torch.unique(bad // per_set).tolist():
                idx = bad[(bad // per_set) == set_id]
                xs, ts = x0[idx].contiguous(), (t[idx].contiguous() if t.dim() == 2 else t)
                ms, vs, gs = cut(meal, idx), cut(tvns, idx), cut(gd, idx)
                o1, n1 = ode_vec[17 * set_id:17 * (set_id + 1)], nn_flat[P * set_id:P * (set_id + 1)]

                def solve(sl=None, tp=None, want=True, xs=xs, ts=ts, ms=ms, vs=vs, gs=gs, o1=o1, n1=n1):
                    c2 = lambda v: None if v is None else (v if sl is None else v[sl].contiguous())   # noqa: E731
                    return hode.solve_fwd(c2(xs), c2(ts) if ts.dim() == 2 else ts, c2(ms), c2(vs), c2(gs), o1, n1, H, L, method=method,
                                          rtol=rtol, atol=atol, n_sets=1, want_tape=want and tp is None, tape=tp if want else None,
                                          max_steps=big)
                fits = idx.numel() * per_traj <= left
                s2 = solve(want=fits)
                if fits:
                    left -= idx.numel() * per_traj
                    extras.append((idx, set_id, s2, None))
                else:
                    extras.append((idx, set_id, s2, (solve, max(1, _tape_budget(x0.device) // (2 * per_traj)))))
    return _Taped(sol, extras, None if extras else worst)


class _SolveFn(torch.autograd.Function):
    """forward solve (K2+K3); backward = reverse-time discrete adjoint (K4)."""

    @staticmethod
    def forward(ctx, x0, nn_flat, ode_vec, t, meal, tvns, gd, H, L, method, rtol, atol, n_sets, info, tape_steps=None):
        need_tape = any(ctx.needs_input_grad[:3])
        B, T = x0.shape[0], t.shape[-1]
        steps = (_small_tape_steps(B, T, method, x0.element_size(), L, H, tape_steps) if need_tape else None) or _tape_steps(T, method, tape_steps)
        per_traj = hode.capi.tape_nbytes(1, steps, x0.element_size(), L, H)
        budget = _tape_budget(x0.device, B * per_traj) if need_tape else 0
        ctx.chunked = need_tape and B * per_traj > budget
        ctx.sol = None
        if need_tape and not ctx.chunked:
            sol = ctx.sol = _solve_taped(x0, t, meal, tvns, gd, ode_vec, nn_flat, H, L, method, rtol, atol, n_sets, steps)
            info["n_budget_retries"] = sol.n_retried
            if sol.worst is not None:
                info["worst_status"] = sol.worst
        else:
            # above the tape budget: no tape now, the backward re-integrates chunk by chunk (same trajectories: a
            # trajectory's result does not depend on its step budget unless it runs out, and then it is retried)
            sol = hode.solve_fwd(x0, t, meal, tvns, gd, ode_vec, nn_flat, H, L, method=method, rtol=rtol, atol=atol,
                                 n_sets=n_sets)
        if ctx.chunked:
            ctx.args = (x0, nn_flat, ode_vec, t, meal, tvns, gd, H, L, method, rtol, atol, n_sets, steps,
                        per_traj)
        info["status"], info["nsteps"], info["nfev"] = sol.status, sol.nsteps, sol.nfev
        return sol.y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        need = ctx.needs_input_grad
        if not ctx.chunked:
            gx0, gnn, gode = ctx.sol.backward(gy, want_gnn=need[1], want_gode=need[2])
            ctx.sol = None
            return (gx0 if need[0] else None, gnn, gode) + (None,) * 12
        x0, nn_flat, ode_vec, t, meal, tvns, gd, H, L, method, rtol, atol, n_sets, steps, per_traj = ctx.args
        ctx.args = None
        cap = max(1, _tape_budget(x0.device) // per_traj)      # what fits NOW (other tapes may have been freed or made)
        gy = gy.contiguous()
        B, P = x0.shape[0], nn_flat.numel() // n_sets
        G = B // n_sets                                   # trajectories per parameter set (contiguous groups)
        gx0 = torch.empty_like(x0)
        gnn = torch.zeros_like(nn_flat) if need[1] else None
        gode = torch.zeros_like(ode_vec) if need[2] else None
        # chunk = m whole parameter sets when a set fits the budget, else a slice of one set
        pieces = [(s0 * G + a, (s1 - 1) * G + b, s0, s1) for s0, s1, a, b in _pieces(n_sets, G, cap)]
        tape = None
        cut = lambda v, lo, hi: None if v is None else v[lo:hi]          # noqa: E731  ([B,T] and [B] inputs alike)
        for lo, hi, s0, s1 in pieces:
            sol = _solve_taped(x0[lo:hi], t[lo:hi] if t.dim() == 2 else t, cut(meal, lo, hi), cut(tvns, lo, hi),
                               cut(gd, lo, hi), ode_vec[17 * s0:17 * s1], nn_flat[P * s0:P * s1], H, L, method, rtol, atol,
                               s1 - s0, steps, tape=tape)
            tape = sol.tape                                              # largest chunk first: later ones fit
            g0, gn, go = sol.backward(gy[lo:hi], want_gnn=need[1], want_gode=need[2])
            gx0[lo:hi] = g0
            if gn is not None:
                gnn[P * s0:P * s1] += gn
            if go is not None:
                gode[17 * s0:17 * s1] += go
        return (gx0 if need[0] else None, gnn, gode) + (None,) * 12


def _allreduce_sum(tensors, group):
    """One all-reduce(sum) of several device tensors (flattened into one fp64 buffer: 16 x 13 527 values for config 5).
    RCCL for an "nccl" group; staged through the host for gloo (rehearsal on one GPU, CPU tests)."""
    import torch.distributed as dist
    flat = torch.cat([v.reshape(-1).double() for v in tensors])
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(flat, group=group)
    else:
        host = flat.cpu()
        dist.all_reduce(host, group=group)
        flat = host.to(flat.device)
    off = 0
    for v in tensors:
        v.copy_(flat[off:off + v.numel()].reshape(v.shape).to(v.dtype))
        off += v.numel()


def _pieces(n_sets, G, cap):
    """Split n_sets x G trajectories into pieces of at most `cap`: whole parameter sets when one fits, slices of a
    set otherwise.  -> [(first set, last set + 1, lo, hi)] with [lo, hi) the patient range inside a set; largest first."""
    if cap >= G:
        m = min(n_sets, cap // G)
        return [(s, min(s + m, n_sets), 0, G) for s in range(0, n_sets, m)]
    return [(s, s + 1, a, min(a + cap, G)) for s in range(n_sets) for a in range(0, G, cap)]


def _gauss_lik_core(need, x0, nn_flat, ode_vec, t, meal, tvns, gd, obs, H, L, method, rtol, atol, S, info, group=None, want_y=False,
                    tape_steps=None, scale=1.0, ss=None):
    """Body of _GaussLikFn.forward (also the data term of _TrainLossFn): sum of squares into `ss` (fp64[1]), gradients of
    scale * sum of squares with respect to (x0, nn_flat, ode_vec) where need[i], trajectories when want_y."""
    grads = any(need[:3])
    B, T = x0.shape[0], t.shape[-1]
    P = nn_flat.numel() // S
    steps = (_small_tape_steps(S * B, T, method, x0.element_size(), L, H, tape_steps) if grads else None) or _tape_steps(T, method, tape_steps)
    per_traj = hode.capi.tape_nbytes(1, steps, x0.element_size(), L, H)
    cap = max(1, _tape_budget(x0.device, S * B * per_traj) // per_traj) if grads else S * B
    if ss is None:
        ss = torch.zeros(1, dtype=torch.float64, device=x0.device)
    pieces = _pieces(S, B, cap)
    one_piece = len(pieces) == 1
    gx0 = torch.zeros_like(x0) if need[0] else None
    gnn = torch.zeros_like(nn_flat) if need[1] and not one_piece else None
    gode = torch.zeros_like(ode_vec) if need[2] and not one_piece else None
    tape, stat, nst, nfe, ys, retried, worst = None, [], [], [], [], 0, 0
    for s0, s1, lo, hi in pieces:
        m = s1 - s0
        rep = lambda v: None if v is None else (v[lo:hi].repeat(m, *([1] * (v.dim() - 1))) if m > 1 else v[lo:hi])  # noqa: E731
        if grads:
            sol = _solve_taped(rep(x0), t if t.dim() == 1 else rep(t), rep(meal), rep(tvns), rep(gd), ode_vec[17 * s0:17 * s1],
                               nn_flat[P * s0:P * s1], H, L, method, rtol, atol, m, steps, tape=tape)
            tape = sol.tape
            retried += sol.n_retried
            worst = None if (worst is None or sol.worst is None) else max(worst, sol.worst)
        else:
            sol = hode.solve_fwd(rep(x0), t if t.dim() == 1 else rep(t), rep(meal), rep(tvns), rep(gd), ode_vec[17 * s0:17 * s1],
                                 nn_flat[P * s0:P * s1], H, L, method=method, rtol=rtol, atol=atol, n_sets=m)
        _, gy = hode.mse_fwd_bwd(sol.y, rep(obs), scale, loss_sum=ss, want_grad=grads)
        if grads:
            g0, gn, go = sol.backward(gy, want_gnn=need[1], want_gode=need[2])
            if gx0 is not None:
                gx0[lo:hi] += g0.view(m, hi - lo, 6).sum(0)
            if one_piece:
                gnn, gode = gn, go                 # (the whole batch in one launch: the adjoint's buffers ARE the result)
            else:
                if gn is not None:
                    gnn[P * s0:P * s1] += gn
                if go is not None:
                    gode[17 * s0:17 * s1] += go
        stat.append(sol.status), nst.append(sol.nsteps), nfe.append(sol.nfev)
        if want_y:
            ys.append(sol.y)
    one = len(stat) == 1
    info["status"], info["nsteps"], info["nfev"] = (stat[0], nst[0], nfe[0]) if one else (torch.cat(stat), torch.cat(nst), torch.cat(nfe))
    info["n_budget_retries"] = retried
    if grads and worst is not None:
        info["worst_status"] = worst          # the host has already looked (one synchronisation per piece): 0 = nothing failed
    if group is not None:
        # patients sharded over the ranks, the SAME S draws everywhere: one all-reduce(sum) of
        # [per-set MLP grads | per-set ODE grads | sum of squares] makes value and gradient global on every rank
        _allreduce_sum([v for v in (gnn, gode, ss) if v is not None], None if group is True else group)
    y = None
    if want_y:                                     # the trajectories themselves (S = 1), e.g. for the physics points
        y = ys[0] if len(ys) == 1 else torch.cat(ys)
    return ss, y, (gx0, gnn, gode)


class _GaussLikFn(torch.autograd.Function):
    """sum_{s,b,k,c} (y_s[b,k,c] - obs[b,k,c])^2 over S parameter sets x B patients, in fp64 -- the data term of the ELBO
    (reference inference/vi.py:60-118) -- with its gradient computed IN THE SAME PASS: per piece of the batch, forward
    solve with tape -> fused residual / cotangent (mse kernel) -> adjoint.  Because the dependence of this term on y is
    known, nothing has to be kept for a later backward: no separate tape-less forward above the tape budget (BASELINE
    config 5: 0.58 -> 0.44 s per step), no S-fold copies of the inputs, and y itself lives only piece by piece."""

    @staticmethod
    def forward(ctx, x0, nn_flat, ode_vec, t, meal, tvns, gd, obs, H, L, method, rtol, atol, S, info, group=None,
                want_y=False, tape_steps=None):
        ss, y, ctx.grads = _gauss_lik_core(ctx.needs_input_grad, x0, nn_flat, ode_vec, t, meal, tvns, gd, obs, H, L, method, rtol, atol,
                                           S, info, group, want_y, tape_steps)
        if want_y:
            ctx.mark_non_differentiable(y)
            return ss[0], y
        return ss[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, *_):
        gx0, gnn, gode = ctx.grads
        ctx.grads = None
        sc = lambda v: None if v is None else (v * g.to(v.dtype))          # noqa: E731
        return (sc(gx0), sc(gnn), sc(gode)) + (None,) * 15


_coef_cache: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


class _TrainLossFn(torch.autograd.Function):
    """loss() of reference models/hybrid_ode_nn.py:263-351 as ONE autograd node: total = data + lambda1 * physics + lambda2 * reg
    with the gradient of all three terms with respect to the flat parameter vector computed in the same pass.

        data     forward solve with tape -> residual / cotangent (mse kernel) -> adjoint           (_gauss_lik_core)
        physics  states at the sampled grid indices -> ONE batched two-point solve (the finite-difference target, no gradient:
                 the reference's solve is detached) -> f(t, x, u) (K1) -> residual against the target -> VJP (K5)
        reg      lambda2 * sum ||W||^2 as a masked dot product, gradient added in place

    About thirty launches where the op-by-op graph of the same arithmetic issues a hundred and ten -- at the reference's batch
    (32 windows x 61 points) the optimisation step is bound by the number of launches on the host AND on the device
    (tools/prof_class_step.py).  Same terms, same reference quirks (n = min(20, len(time_points)), the (m / n) factor of indices
    that fall outside the grid, lambda2 applied twice); the three components come back beside the total."""

    @staticmethod
    def forward(ctx, nn_flat, ode_vec, x0, t, meal, tvns, gd, obs, idx_d, n_draw, lam1, lam2, wmask, H, L, info, tape_steps):
        need = (False, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        dev = x0.device
        B, T = x0.shape[0], t.shape[-1]
        n_el = obs.numel()
        acc = torch.zeros(3, dtype=torch.float64, device=dev)              # sum of squares (data) | sum of squares (physics) | sum ||W||^2
        # ---- data: d/d theta of mean((y - obs)^2); the 1 / n_el rides in the cotangent
        _, y, (_, gnn, gode) = _gauss_lik_core(need, x0, nn_flat, ode_vec, t, meal, tvns, gd, obs, H, L, hode.METHOD_DP54, 1e-6, 1e-8, 1,
                                               info, None, True, tape_steps, scale=1.0 / n_el, ss=acc[0:1])
        m = 0 if idx_d is None else int(idx_d.numel())
        M = B * m
        if m > 0:
            # ---- physics: mean over the m sampled indices of MSE((x(0.1) - x) / 0.1, f(t, x, u)), times m / n_draw
            state = torch.index_select(y, 1, idx_d).reshape(M, 6)                         # patient-major: [b][index]
            t_true = (torch.index_select(t, 1, idx_d).reshape(M) if t.dim() == 2 else torch.index_select(t, 0, idx_d).repeat(B))
            pick = lambda v: None if v is None else (torch.index_select(v, 1, idx_d).reshape(M) if v.dim() == 2   # noqa: E731
                                                     else v.reshape(B, 1).expand(B, m).reshape(M))
            ms, vs, gs = pick(meal), pick(tvns), pick(gd)
            nxt = hode.solve_fwd(state, _short_span(dev), ms, vs, gs, ode_vec, nn_flat, H, L, rtol=1e-6, atol=1e-8).y
            fd = nxt[:, 1, :] - state
            fd.div_(0.1)
            f = hode.rhs_fwd(state, t_true, ms, vs, gs, ode_vec, nn_flat, H, L)
            c_phys = (m / n_draw) / (M * 6)
            _, gout = hode.mse_fwd_bwd(f, fd, lam1 * c_phys, loss_sum=acc[1:2], want_grad=True)
            _, _, gnn_p, gode_p = hode.rhs_bwd(state, t_true, ms, vs, gs, ode_vec, nn_flat, H, L, gout, want_gnn=need[1], want_gode=need[2])
            if gnn is not None:
                gnn.add_(gnn_p)
            if gode is not None:
                gode.add_(gode_p)
        else:
            c_phys = 0.0
        if lam2 > 0:
            # ---- reg: nn_residual.regularization_loss(l2_weight=lambda2) = lambda2 * sum ||W||^2, and the total takes lambda2 * reg
            acc[2:3].copy_(torch.dot(nn_flat * wmask, nn_flat).reshape(1))
            if gnn is not None:
                gnn.addcmul_(wmask, nn_flat, value=2.0 * lam2 * lam2)
        key = (dev, n_el, c_phys, lam1, lam2)
        co = _coef_cache.get(key)
        if co is None:
            if len(_coef_cache) > 64:
                _coef_cache.clear()
            co = _coef_cache[key] = (torch.tensor([1.0 / n_el, c_phys, lam2], dtype=torch.float64, device=dev),
                                     torch.tensor([1.0, lam1, lam2], dtype=torch.float64, device=dev))
        comps = acc * co[0]                                                # data | physics | reg, as last_loss_components reports them
        total = torch.dot(comps, co[1]).float()
        ctx.grads = (gnn, gode)
        ctx.mark_non_differentiable(comps, y)
        return total, comps, y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g, *_):
        gnn, gode = ctx.grads
        ctx.grads = None
        sc = lambda v: None if v is None else v.mul_(g.to(v.dtype))        # noqa: E731  (the buffers are this node's own)
        return (sc(gnn), sc(gode)) + (None,) * 15


class HybridODENN(nn.Module):
    def __init__(self, ode_params: Optional[Dict[str, float]] = None, nn_hidden: int = 64, nn_layers: int = 4,
                 use_variational: bool = False, prior_params: Optional[Dict[str, Dict[str, float]]] = None,
                 device: Optional[Union[torch.device, str]] = None):
        super().__init__()
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        self.use_variational = use_variational
        self.ode_core = ODECore(ode_params).to(self.device)
        self.nn_residual = NNResidual(input_dim=9, hidden_dim=nn_hidden, output_dim=6, n_layers=nn_layers).to(self.device)
        self.n_states = 6
        self.state_names = ["Glucose", "Insulin", "Glucagon", "GLP1", "GE", "FFA"]
        # True  (default): gradients flow through the solve by the adjoint kernel (north_star).
        # False: the solve is detached exactly like the reference's SciPy round trip.
        self.adjoint = True
        # elbo() / loss(): compute the Gaussian (MSE) data term and its gradient in one pass over the batch -- no tape
        # kept for a later backward.  False: the solve goes through autograd like any other op (_SolveFn).
        self.fused_likelihood = True
        # accepted-step budget per trajectory of solves that record a tape (None: T-1 + max(32, (T-1)//4)); trajectories
        # that need more are retried with the no-grad budget, see _solve_taped
        self.tape_steps: Optional[int] = None
        self.last_solve_info: Dict[str, torch.Tensor] = {}
        self.variational_params = None
        if use_variational:
            self._setup_variational_inference(prior_params)

    # ------------------------------------------------------------------ VI bookkeeping
    def _setup_variational_inference(self, prior_params):
        """Latent = 8 ODE constants + every MLP tensor (hybrid_ode_nn.py:70-106)."""
        shapes = {}
        for name, buf in self.ode_core.named_buffers():
            if name in ("a_GI", "k_I", "rho", "E_max", "EC_50", "V_max", "K_m", "k_L"):
                shapes[f"ode_{name}"] = buf.shape
        for name, p in self.nn_residual.named_parameters():
            shapes[f"nn_{name.replace('.', '_')}"] = p.shape
        means, stds = {}, {}
        for name in shapes:
            if prior_params and name in prior_params:
                means[name] = prior_params[name].get("mean", 0.0)
                stds[name] = prior_params[name].get("std", 1.0)
        self.variational_params = VariationalParameters(shapes, means, stds).to(self.device)

    def get_variational_params(self) -> Tuple[torch.Tensor, torch.Tensor]:
        if not self.use_variational:
            raise ValueError("Model was not initialized with variational inference")
        return self.variational_params.get_flattened_params()

    def sample_posterior(self, n_samples: int = 1) -> List[Dict[str, torch.Tensor]]:
        if not self.use_variational:
            raise ValueError("Model was not initialized with variational inference")
        return self.variational_params.sample(n_samples)

    # ------------------------------------------------------------------ plumbing
    def _check_supported(self):
        if not self.nn_residual.hip_supported():
            raise NotImplementedError("the HIP path computes an MLP 9 -> (<=128) x (1..8) -> 6 with relu / tanh / elu / leaky_relu and "
                                      "without dropout (include/hode.h); other NNResidual configurations are outside the hot path")

    def _params_on(self, dev, params: Optional[Dict[str, torch.Tensor]] = None):
        """(nn_flat, ode_vec) on the compute device; `params` optionally overrides named entries
        (`ode_<buf>` / `nn_<name with . -> _>`, hybrid_ode_nn.py:403-420)."""
        if params is None:
            fast = self._params_fast(dev)
            if fast is not None:
                return fast
        pieces = []
        for name, p in self.nn_residual.named_parameters():
            v = None if params is None else params.get(f"nn_{name.replace('.', '_')}")
            pieces.append((p if v is None else v.to(p.dtype)).reshape(-1).to(dev))
        nn_flat = torch.cat(pieces).float()
        ode = []
        for n in ODE_PARAM_NAMES:
            v = None if params is None else params.get(f"ode_{n}")
            ode.append(torch.as_tensor(getattr(self.ode_core, n) if v is None else v).reshape(()).float().to(dev))
        return nn_flat, torch.stack(ode)

    def _nn_flat_store(self, dev):
        """One fp32 buffer on `dev` that the ten parameters of nn_residual are VIEWS of (parameters() order = the kernels' layout),
        or None when they cannot alias it (model kept on another device / in another dtype).  Checked by address at every
        use and rebuilt when something re-homed the parameters (.to(), .double(), deepcopy, a replaced nn_residual);
        load_state_dict, optimizers and clip_grad_norm_ write through the views."""
        params = list(self.nn_residual.parameters())
        st = self.__dict__.get("_flat_nn")
        if st is not None:
            flat, offs = st
            if (len(offs) == len(params) and flat.device == dev and
                    all(p.dtype == torch.float32 and p.device == dev and p.data_ptr() == flat.data_ptr() + 4 * o and p.is_contiguous()
                        for p, o in zip(params, offs))):
                return flat, params
        if not params or any(p.dtype != torch.float32 or p.device != dev for p in params):
            self.__dict__["_flat_nn"] = None
            return None
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
            offs, off = [], 0
            for q in params:
                q.data = flat[off:off + q.numel()].view(q.shape)
                offs.append(off)
                off += q.numel()
            # weights 1, biases 0: the L2 term of loss() is one masked dot product over the flat vector
            mask = torch.cat([torch.full((q.numel(),), 1.0 if q.dim() == 2 else 0.0, device=dev) for q in params])
        self.__dict__["_flat_nn"] = (flat, offs)
        self.__dict__["_flat_wmask"] = mask
        return flat, params

    def _params_fast(self, dev):
        """(nn_flat, ode_vec) without per-step gathers: the parameters alias one flat buffer (_nn_flat_store) and the 17 mechanistic
        constants are stacked once per change (keyed by the identity and version counter of every buffer: setattr, in-place
        updates and load_state_dict all show).  ~0.02 ms of host time instead of 0.16 (tools/prof_class_step.py)."""
        st = self._nn_flat_store(dev)
        if st is None:
            return None
        flat, params = st
        bufs = [getattr(self.ode_core, n) for n in ODE_PARAM_NAMES]
        if any((not torch.is_tensor(b)) or b.requires_grad for b in bufs):
            return None
        key = tuple((id(b), b._version) for b in bufs)
        oc = self.__dict__.get("_ode_cache")
        if oc is None or oc[0] != key or oc[1].device != dev:
            oc = (key, torch.stack([b.detach().reshape(()).float().to(dev) for b in bufs]))
            self.__dict__["_ode_cache"] = oc
        nn_flat = _FlatNN.apply(flat, *params) if torch.is_grad_enabled() and any(p.requires_grad for p in params) else flat
        return nn_flat, oc[1]

    @staticmethod
    def _input(u, key, dev, n):
        """External input -> (tensor on dev or None).  0-dim / len-1 are broadcast to [n]."""
        if not u or key not in u or u[key] is None:
            return None
        v = torch.as_tensor(u[key]).to(dev, torch.float32)
        if v.dim() == 0 or (v.dim() == 1 and v.shape[0] == 1 and n != 1):
            v = v.reshape(1).expand(n)
        return v.contiguous()

    # ------------------------------------------------------------------ RHS
    def ode_residual(self, t: torch.Tensor, state: torch.Tensor,
                     external_inputs: Optional[Dict[str, torch.Tensor]] = None, _flat=None) -> torch.Tensor:
        """f(t, x, u) = ODECore + NNResidual(t, x, x[...,3], tVNS) on the device (K1; K5 for grads).
        `_flat`: (nn_flat, ode_vec) the caller has already gathered (loss() builds them once per step)."""
        self._check_supported()
        dev = _compute_device()
        single = state.dim() == 1
        x = (state.unsqueeze(0) if single else state).to(dev, torch.float32)
        n = x.shape[0]
        tt = torch.as_tensor(t).to(dev, torch.float32)
        tt = tt.reshape(1).expand(n) if tt.numel() == 1 else tt.reshape(n)
        meal, tvns, gd = (self._input(external_inputs, k, dev, n) for k in ("meal", "tVNS", "GD"))
        nn_flat, ode_vec = self._params_on(dev) if _flat is None else _flat
        nl = self.nn_residual
        out = _RhsFn.apply(x.contiguous(), tt.contiguous(), nn_flat, ode_vec, meal, tvns, gd, nl.hidden_dim, nl.hip_layers)
        out = out.to(state.device)
        return out.squeeze(0) if single else out

    # ------------------------------------------------------------------ solve
    @staticmethod
    def _prep_inputs(initial_state, t_span, external_inputs, dev):
        """(x0[B,6], t[T] or [B,T], {meal, tVNS, GD}) on the compute device in the layouts the C ABI takes."""
        x0 = initial_state.to(dev, torch.float32)
        B = x0.shape[0]
        t = torch.as_tensor(t_span).to(dev, torch.float32)
        if t.dim() == 2 and t.shape[0] != B:
            t = t.reshape(-1) if t.shape[0] == 1 else t
        t = t.contiguous()
        u = external_inputs or {}
        ins = {}
        for key in ("meal", "tVNS", "GD"):
            v = u.get(key)
            if v is None:
                ins[key] = None
                continue
            v = torch.as_tensor(v).to(dev, torch.float32)
            if v.dim() == 2:
                ins[key] = v.contiguous()                       # time-varying on the grid
            else:
                # dim()==1 (or 0): constant per patient, the reference reads values[b] (hybrid_ode_nn.py:230-231)
                v = v.reshape(-1)
                if v.numel() != 1 and v.numel() < B:
                    raise IndexError(f"external input {key!r} has {v.numel()} entries for a batch of {B}")
                ins[key] = (v.expand(B) if v.numel() == 1 else v[:B]).contiguous()
        return x0.contiguous(), t, ins

    def _solve(self, initial_state, t_span, external_inputs, solver, rtol, atol, params=None, n_sets=1,
               nn_flat=None, ode_vec=None, differentiable=None, nn_shared=False):
        self._check_supported()
        dev = _compute_device()
        x0, t, ins = self._prep_inputs(initial_state, t_span, external_inputs, dev)
        if nn_flat is None:
            nn_flat, ode_vec = self._params_on(dev, params)
        method = _SOLVERS.get(str(solver).lower())
        if method is None:
            raise ValueError(f"unknown solver {solver!r}; known: {sorted(_SOLVERS)}")
        diff = (self.adjoint if differentiable is None else differentiable) and not nn_shared
        info = {}
        nl = self.nn_residual
        if diff and torch.is_grad_enabled():
            y = _SolveFn.apply(x0.contiguous(), nn_flat, ode_vec, t, ins["meal"], ins["tVNS"], ins["GD"],
                               nl.hidden_dim, nl.hip_layers, method, float(rtol), float(atol), n_sets, info, self.tape_steps)
        else:
            with torch.no_grad():
                sol = hode.solve_fwd(x0.contiguous(), t, ins["meal"], ins["tVNS"], ins["GD"], ode_vec.detach(),
                                     nn_flat.detach(), nl.hidden_dim, nl.hip_layers, method=method, rtol=float(rtol),
                                     atol=float(atol), n_sets=n_sets, nn_shared=nn_shared)
            y = sol.y
            info = {"status": sol.status, "nsteps": sol.nsteps, "nfev": sol.nfev}
        self.last_solve_info = info
        return y

    def _ship_indices(self, idx, dev):
        """The <= 20 physics indices to the device through a small ring of PINNED buffers, asynchronously: a copy from pageable
        memory makes the host wait for the stream (0.36 ms per step once nothing else stalls the pipeline, tools/prof_class_step.py)."""
        if dev.type != "cuda" or idx.numel() > 32:
            return idx.to(dev)
        ring = self.__dict__.get("_idx_ring")
        if ring is None:
            ring = self.__dict__["_idx_ring"] = {"bufs": [torch.empty(32, dtype=torch.int64).pin_memory() for _ in range(4)],
                                                 "evs": [None] * 4, "i": 0}
        k = ring["i"]
        ring["i"] = (k + 1) % 4
        if ring["evs"][k] is not None:
            ring["evs"][k].synchronize()              # four steps old: complete
        buf = ring["bufs"][k][:idx.numel()]
        buf.copy_(idx)
        out = buf.to(dev, non_blocking=True)
        ring["evs"][k] = torch.cuda.Event()
        ring["evs"][k].record()
        return out

    def _flush_deferred_warning(self, wait=False):
        """Look at the worst statuses that have arrived in pinned memory since (all of them when wait=True): each is written into
        the solve's own info dict as `worst_status`; a non-zero one logs the reference's warnings now."""
        q = self.__dict__.get("_pending_fail")
        while q:
            host, ev, info = q[0]
            if not wait and len(q) < 4 and not ev.query():
                break
            ev.synchronize()
            q.pop(0)
            info["worst_status"] = int(host[0])
            if info["worst_status"] != 0:
                self._warn_failures(info)

    def _warn_failures(self, info, defer=False):
        """Never raise on an integration failure: log and keep the zero rows (hybrid_ode_nn.py:243-256).
        defer=True (the training path, loss()): when the host does not know yet whether anything failed, it does not wait for
        the device to find out -- the worst status travels to pinned host memory behind the solve and is looked at when a later
        step arrives here, or when solve_failures() asks: the same warnings, a step or two late, no pipeline stall."""
        if self.__dict__.get("_pending_fail"):
            self._flush_deferred_warning()
        if defer and logger.isEnabledFor(logging.WARNING) and "status" in info and "worst_status" not in info and info["status"].is_cuda:
            slots = self.__dict__.get("_fail_host")
            if slots is None:
                slots = self.__dict__["_fail_host"] = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(4)]
                self.__dict__["_fail_i"] = 0
            k = self.__dict__["_fail_i"]
            self.__dict__["_fail_i"] = (k + 1) % 4
            slots[k].copy_(info["status"].max().reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self.__dict__.setdefault("_pending_fail", []).append((slots[k], ev, info))
            return
        if logger.isEnabledFor(logging.WARNING) and "status" in info and info.get("worst_status", 1) != 0:
            bad = torch.nonzero(info["status"]).flatten()
            st = info["status"][bad].tolist()
            info.setdefault("worst_status", max(st) if st else 0)        # (the host has looked now)
            if st:
                msgs = {1: "step budget exhausted", 2: "Required step size is less than spacing between numbers.",
                        3: "non-finite state"}
                for b, s in list(zip(bad.tolist(), st))[:8]:
                    logger.warning(f"ODE solver failed for batch {b}: {msgs.get(s, s)}")

    def solve_failures(self) -> int:
        """Number of trajectories of the last solve / loss / elbo that did not reach the end of their grid (their rows
        from the failure on are zero, hybrid_ode_nn.py:243-256).  Synchronises with the device."""
        self._flush_deferred_warning(wait=True)
        st = self.last_solve_info.get("status")
        return 0 if st is None else int((st != 0).sum())

    def forward(self, initial_state: torch.Tensor, t_span: torch.Tensor,
                external_inputs: Optional[Dict[str, torch.Tensor]] = None, solver: str = "dopri5",
                rtol: float = 1e-6, atol: float = 1e-8) -> torch.Tensor:
        """Trajectories [B, T, 6] (or [T, 6] for a (6,) initial state) at the grid `t_span`
        ([T] shared or [B, T]); 2-D inputs are piecewise linear on the grid, 1-D constant per patient."""
        single = initial_state.dim() == 1
        x0 = initial_state.unsqueeze(0) if single else initial_state
        y = self._solve(x0, t_span, external_inputs, solver, rtol, atol)
        self._warn_failures(self.last_solve_info)
        y = y.to(self.device)
        return y.squeeze(0) if single else y

    def forward_with_params(self, params: Union[torch.Tensor, Dict[str, torch.Tensor]], *args, **kwargs) -> torch.Tensor:
        """forward() with named parameter overrides.  The reference swaps buffers/weights in, runs, and
        restores (hybrid_ode_nn.py:381-438); here the overrides go straight into the parameter vectors
        of the launch, the module is never mutated.  A flat tensor is a logged no-op, as there."""
        if isinstance(params, torch.Tensor):
            logger.warning("Flattened parameter vector not fully implemented")
            return self.forward(*args, **kwargs)
        names = ["initial_state", "t_span", "external_inputs", "solver", "rtol", "atol"]
        kw = dict(zip(names, args))
        kw.update(kwargs)
        single = kw["initial_state"].dim() == 1
        x0 = kw["initial_state"].unsqueeze(0) if single else kw["initial_state"]
        y = self._solve(x0, kw["t_span"], kw.get("external_inputs"), kw.get("solver", "dopri5"),
                        kw.get("rtol", 1e-6), kw.get("atol", 1e-8), params=params)
        self._warn_failures(self.last_solve_info)
        y = y.to(self.device)
        return y.squeeze(0) if single else y

    def forward_param_sets(self, param_sets: List[Dict[str, torch.Tensor]], initial_state: torch.Tensor,
                           t_span: torch.Tensor, external_inputs: Optional[Dict[str, torch.Tensor]] = None,
                           solver: str = "dopri5", rtol: float = 1e-6, atol: float = 1e-8) -> torch.Tensor:
        """S parameter sets x B patients in ONE launch -> [S, B, T, 6] (VI samples inference/vi.py:88-100,
        Sobol sets plots/plot_all.py:171-196, posterior predictive bayes.py:198-206)."""
        dev = _compute_device()
        S = len(param_sets)
        single = initial_state.dim() == 1
        x0 = initial_state.unsqueeze(0) if single else initial_state
        B = x0.shape[0]
        flat = [self._params_on(dev, p) for p in param_sets]
        nn_flat = torch.cat([f[0] for f in flat])
        ode_vec = torch.cat([f[1] for f in flat])
        rep = lambda v: None if v is None else torch.as_tensor(v).repeat(*([S] + [1] * (torch.as_tensor(v).dim() - 1)))  # noqa: E731
        t = torch.as_tensor(t_span)
        u = {k: rep(v) for k, v in (external_inputs or {}).items() if torch.as_tensor(v).dim() >= 1 and torch.as_tensor(v).numel() > 1}
        for k, v in (external_inputs or {}).items():
            if k not in u:
                u[k] = v
        y = self._solve(x0.repeat(S, 1), rep(t) if t.dim() == 2 else t, u, solver, rtol, atol, n_sets=S,
                        nn_flat=nn_flat, ode_vec=ode_vec)
        self._warn_failures(self.last_solve_info)
        y = y.reshape(S, B, y.shape[1], 6).to(self.device)
        return y[:, 0] if single else y

    def forward_ode_sets(self, ode_sets: Dict[str, torch.Tensor], initial_state: torch.Tensor, t_span: torch.Tensor,
                         external_inputs: Optional[Dict[str, torch.Tensor]] = None, solver: str = "dopri5",
                         rtol: float = 1e-6, atol: float = 1e-8) -> torch.Tensor:
        """S sets of MECHANISTIC constants x B patients through the model's one network, in ONE launch -> [S, B, T, 6]
        ([S, T, 6] for a (6,) initial state).  `ode_sets`: constant name (an ODECore buffer, with or without the `ode_`
        prefix of forward_with_params) -> S values; constants not named keep the model's value.

        The batched form of the Sobol study of the reference's plots/plot_all.py:139-196: there, 16 384 Saltelli samples of
        seven constants are written into `model.ode_core` one at a time (`setattr(..., torch.tensor(value))`, :179-181, i.e.
        rounded to fp32) and `model.forward` integrates one patient per sample ("~5-10 minutes", README.md:248).  Here the
        samples ride in the kernel's parameter-set dimension with the network shared (HODE_LAYERS_NN_SHARED).  Forward only,
        never differentiated (the study only reads trajectories)."""
        dev = _compute_device()
        cols = {}
        for name, vals in ode_sets.items():
            key = name[4:] if name.startswith("ode_") else name
            if key not in ODE_PARAM_NAMES:
                raise ValueError(f"unknown mechanistic constant {name!r}; known: {list(ODE_PARAM_NAMES)}")
            cols[ODE_PARAM_NAMES.index(key)] = torch.as_tensor(vals).reshape(-1).to(dev, torch.float32)
        if not cols:
            raise ValueError("forward_ode_sets needs at least one constant to vary")
        S = {int(v.numel()) for v in cols.values()}
        if len(S) != 1:
            raise ValueError("every constant needs the same number of samples")
        S = S.pop()
        nn_flat, ode_vec = self._params_on(dev)
        ode_mat = ode_vec.reshape(1, -1).repeat(S, 1)
        for j, v in cols.items():
            ode_mat[:, j] = v
        single = initial_state.dim() == 1
        x0 = initial_state.unsqueeze(0) if single else initial_state
        B = x0.shape[0]
        rep = lambda v: torch.as_tensor(v).repeat(*([S] + [1] * (torch.as_tensor(v).dim() - 1)))  # noqa: E731
        t = torch.as_tensor(t_span)
        u = {k: (rep(v) if torch.as_tensor(v).dim() >= 1 and torch.as_tensor(v).numel() > 1 else v)
             for k, v in (external_inputs or {}).items()}
        with torch.no_grad():
            y = self._solve(x0.repeat(S, 1), rep(t) if t.dim() == 2 else t, u, solver, rtol, atol, n_sets=S,
                            nn_flat=nn_flat, ode_vec=ode_mat.reshape(-1), nn_shared=True)
        self._warn_failures(self.last_solve_info)
        y = y.reshape(S, B, y.shape[1], 6).to(self.device)
        return y[:, 0] if single else y

    # ------------------------------------------------------------------ ELBO (BASELINE config 5)
    def elbo(self, batch: Dict[str, torch.Tensor], n_samples: int = 16, noise_sigma: float = 0.1,
             solver: str = "dopri5", rtol: float = 1e-6, atol: float = 1e-8, group=None, return_components: bool = False):
        """Monte-Carlo ELBO of reference inference/vi.py:60-118 (`VariationalInference.elbo`):
             E_q[log p(obs | theta)] - KL[q || p],   theta_s = mu + eps_s * exp(log_sigma),  s = 1..S,
        one parameter draw shared by the whole batch per sample (vi.py:88-100).  All S x B trajectories are
        ONE launch (the S draws ride in the kernel's parameter-set dimension) and -- unlike the reference,
        whose likelihood term is detached -- the reparameterised gradient reaches mu / log_sigma
        through the adjoint kernel (per-set MLP and ODE-constant gradients).  KL and the likelihood sum are
        accumulated in fp64.
        group: one process per GPU, `batch` = this rank's shard of the patients, torch's RNG seeded identically on
        every rank (same draws): the data term and its gradient are summed over the group with ONE all-reduce, the KL
        term is parameter-only and computed redundantly -- every rank returns the ELBO of the whole cohort and ends the
        backward with identical gradients.
        return_components: also return {'elbo', 'kl', 'log_likelihood'} (what VariationalInference.elbo hands back)."""
        if not self.use_variational:
            raise ValueError("Model was not initialized with variational inference")
        self._check_supported()
        dev = _compute_device()
        x0, obs, tp = batch["initial_state"], batch["observations"], batch["time_points"]
        u = batch.get("external_inputs", None) or {}
        B, S = x0.shape[0], int(n_samples)
        draws = self.variational_params.sample(S)                      # reparameterised, differentiable
        flat = [self._params_on(dev, d) for d in draws]
        nn_flat = torch.cat([f[0] for f in flat])
        ode_vec = torch.cat([f[1] for f in flat])

        n_obs = obs.numel()
        if group is not None:
            if not (self.fused_likelihood and self.adjoint):
                raise ValueError("elbo(group=...) needs the fused likelihood route (fused_likelihood and adjoint both True)")
            cnt = torch.tensor([float(n_obs)], dtype=torch.float64, device=dev)
            _allreduce_sum([cnt], None if group is True else group)
            n_obs = float(cnt)
        log_norm = 0.5 * n_obs * torch.log(torch.tensor(2 * torch.pi * noise_sigma ** 2, dtype=torch.float64, device=dev))
        kl = self.variational_params.kl_divergence().double().to(dev)
        method = _SOLVERS.get(str(solver).lower())
        if method is None:
            raise ValueError(f"unknown solver {solver!r}; known: {sorted(_SOLVERS)}")
        if self.fused_likelihood and self.adjoint and (torch.is_grad_enabled() or group is not None):
            # data term and its gradient in one pass over the S x B trajectories (_GaussLikFn)
            xs, tt, ins = self._prep_inputs(x0, tp, u, dev)
            info, nl = {}, self.nn_residual
            ss = _GaussLikFn.apply(xs, nn_flat, ode_vec, tt, ins["meal"], ins["tVNS"], ins["GD"],
                                   obs.to(dev, torch.float32).contiguous(), nl.hidden_dim, nl.hip_layers, method, float(rtol),
                                   float(atol), S, info, group, False, self.tape_steps)
            self.last_solve_info = info
            self._warn_failures(info)
            log_lik = -0.5 * ss / (noise_sigma ** 2 * S) - log_norm
            return self._elbo_out(log_lik, kl, return_components)

        def rep(v):
            v = torch.as_tensor(v)
            return v.repeat(S, *([1] * (v.dim() - 1))) if v.dim() >= 1 and v.shape[0] == B else v
        tt = torch.as_tensor(tp)
        y = self._solve(x0.repeat(S, 1), rep(tt) if tt.dim() == 2 else tt, {k: rep(v) for k, v in u.items()}, solver,
                        rtol, atol, n_sets=S, nn_flat=nn_flat, ode_vec=ode_vec)
        self._warn_failures(self.last_solve_info)
        resid = (obs.to(dev, torch.float32).repeat(S, 1, 1) - y).double() / noise_sigma
        log_lik = -0.5 * resid.pow(2).sum() / S - log_norm
        return self._elbo_out(log_lik, kl, return_components)

    def _elbo_out(self, log_lik, kl, return_components):
        elbo = (log_lik - kl).to(self.device)
        if not return_components:
            return elbo
        return elbo, {"elbo": elbo, "kl": kl.to(self.device), "log_likelihood": log_lik.to(self.device)}

    # ------------------------------------------------------------------ loss
    def loss(self, batch: Dict[str, torch.Tensor], lambda1: float = 1.0, lambda2: float = 1.0,
             use_physics_loss: bool = True) -> torch.Tensor:
        """total = data + lambda1 * physics + lambda2 * reg  (hybrid_ode_nn.py:263-351).

        data    = MSE(predictions, observations); differentiable through the adjoint (self.adjoint).
        physics = mean over <= 20 sampled grid indices of MSE((x(0.1) - x)/0.1, f(t, x)): the reference
                  runs B short solves per index in a Python loop -- here ALL indices x patients are one
                  batched launch (inputs frozen at the sampled grid value, local time restarted at 0).
        reg     = nn_residual.regularization_loss(l2_weight=lambda2), i.e. lambda2^2 * sum ||W||^2 overall.
        Reference quirks kept for loss-value parity: n = min(20, len(time_points)) and
        randperm(len(time_points)) use len() of the tensor, which is B for a batched [B,T] grid."""
        x0 = batch["initial_state"]
        obs = batch["observations"]
        tp = batch["time_points"]
        u = batch.get("external_inputs", None)
        dev = _compute_device()
        self._check_supported()
        nn_flat, ode_vec = self._params_on(dev)        # gathered ONCE per step: the solve, the physics solve and f() share them
        idx = None
        if use_physics_loss and lambda1 > 0:
            # the physics indices are drawn and shipped to the device BEFORE the solve is queued: a host-to-device copy
            # from pageable memory waits for the stream, and here the stream is still empty.  Same draw as the reference,
            # whose randperm follows its solve: nothing on this path consumes torch's host generator in between
            n = min(20, len(tp))
            idx = torch.randperm(len(tp))[:n]                         # global RNG, same draw as the reference
            idx = idx[idx < tp.shape[-1]]                              # (the reference raises IndexError here)
            idx_d = self._ship_indices(idx, dev)

        fused_all = (self.fused_likelihood and self.adjoint and torch.is_grad_enabled() and nn_flat.requires_grad
                     and self.__dict__.get("_flat_nn") is not None and not (lambda2 > 0 and self.use_variational))
        if fused_all:
            # the whole loss and its gradient as one autograd node (_TrainLossFn)
            xs, tt, ins = self._prep_inputs(x0, tp, u, dev)
            info, nl = {}, self.nn_residual
            total, comps, pred = _TrainLossFn.apply(nn_flat, ode_vec, xs, tt, ins["meal"], ins["tVNS"], ins["GD"],
                                                    obs.to(dev, torch.float32).contiguous(), idx_d if idx is not None else None,
                                                    (min(20, len(tp)) if idx is not None else 1), float(lambda1) if idx is not None else 0.0,
                                                    float(lambda2), self.__dict__["_flat_wmask"], nl.hidden_dim, nl.hip_layers, info,
                                                    self.tape_steps)
            self.last_solve_info = info
            self._warn_failures(info, defer=True)
            self.last_loss_components = (comps[0], comps[1], comps[2])
            if logger.isEnabledFor(logging.DEBUG):
                logger.debug(f"Loss components - Data: {float(comps[0]):.4f}, Physics: {float(comps[1]):.4f}, Reg: {float(comps[2]):.4f}")
            return total.to(self.device)

        if self.fused_likelihood and self.adjoint and torch.is_grad_enabled():
            # solve + MSE + adjoint piece by piece in one pass (_GaussLikFn): no tape is held until backward() and a
            # cohort whose tape exceeds the budget needs no second forward.  Defaults like reference :291.
            xs, tt, ins = self._prep_inputs(x0, tp, u, dev)
            info, nl = {}, self.nn_residual
            ss, pred = _GaussLikFn.apply(xs, nn_flat, ode_vec, tt, ins["meal"], ins["tVNS"], ins["GD"],
                                         obs.to(dev, torch.float32).contiguous(), nl.hidden_dim, nl.hip_layers, hode.METHOD_DP54,
                                         1e-6, 1e-8, 1, info, None, True, self.tape_steps)
            self.last_solve_info = info
            self._warn_failures(info, defer=True)
            data_loss = (ss / obs.numel()).float()
        else:
            pred = self._solve(x0, tp, u, "dopri5", 1e-6, 1e-8,      # defaults, like reference :291
                               nn_flat=nn_flat, ode_vec=ode_vec)
            self._warn_failures(self.last_solve_info)
            data_loss = torch.nn.functional.mse_loss(pred, obs.to(dev, torch.float32))

        physics_loss = torch.zeros((), device=dev)
        if idx is not None:
            if idx.numel() > 0:
                B, m = pred.shape[0], idx.numel()
                state = pred.detach()[:, idx_d, :].transpose(0, 1).reshape(m * B, 6).contiguous()   # [m*B, 6]
                tpd = tp.to(dev, torch.float32)
                t_true = (tpd[:, idx_d].transpose(0, 1) if tpd.dim() == 2 else tpd[idx_d].unsqueeze(1).expand(m, B)).reshape(m * B)
                ext = {}
                for key, v in (u or {}).items():
                    v = torch.as_tensor(v).to(dev, torch.float32)
                    ext[key] = (v[:, idx_d].transpose(0, 1) if v.dim() == 2 else v.reshape(1, -1).expand(m, B)).reshape(m * B).contiguous()
                with torch.no_grad():                                  # FD target carries no gradient (reference: detached solve)
                    nxt = self._solve(state, _short_span(dev), ext, "dopri5", 1e-6, 1e-8, differentiable=False,
                                      nn_flat=nn_flat, ode_vec=ode_vec)[:, 1, :]
                    fd = (nxt - state) / 0.1
                f = self.ode_residual(t_true, state.requires_grad_(True), ext, _flat=(nn_flat, ode_vec))
                # mean over indices of per-index MSE == MSE over the stacked [m*B, 6] block
                physics_loss = torch.nn.functional.mse_loss(fd, f) * (m / n)

        reg_loss = torch.zeros((), device=dev)
        if lambda2 > 0:
            if self.use_variational:
                reg_loss = bayes_loss(self, obs, noise_sigma=1.0, n_samples=5)
            elif self.__dict__.get("_flat_nn") is not None and nn_flat.numel() == self.__dict__["_flat_wmask"].numel():
                # nn_residual.regularization_loss(l2_weight=lambda2) = lambda2 * sum ||W_l||^2 as ONE masked dot product over the flat
                # vector (the per-layer pow / sum / mul / add chain is twenty launches forward and as many backward)
                reg_loss = lambda2 * torch.dot(nn_flat * self.__dict__["_flat_wmask"], nn_flat)
            else:
                reg_loss = self.nn_residual.regularization_loss(l2_weight=lambda2)
                reg_loss = reg_loss.to(dev) if torch.is_tensor(reg_loss) else torch.tensor(float(reg_loss), device=dev)

        total = data_loss + lambda1 * physics_loss + lambda2 * reg_loss
        if logger.isEnabledFor(logging.DEBUG):
            logger.debug(f"Loss components - Data: {float(data_loss):.4f}, Physics: {float(physics_loss):.4f}, "
                         f"Reg: {float(reg_loss):.4f}")
        self.last_loss_components = (data_loss.detach(), physics_loss.detach(), reg_loss.detach())
        return total.to(self.device)
