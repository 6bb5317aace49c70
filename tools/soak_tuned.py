#!/usr/bin/env python3
"""Soak test of the tuned path (4 x 64 and smaller: solve_fwd_kernel incl. its many-short-trajectories instantiation, the
wave-specialised adjoint with its step headers gathered one step ahead and the slow injection path): random shapes, batches, grids
(repeated times, batched grids), parameter sets, NaN states and step budgets for N seconds.  Per case: forward with tape + adjoint
twice -> finite, run-to-run bitwise; forward == the same trajectories in launches of <= 300; a quarter of the cases also against the fp64 ORACLE
(RK4 cases: held to 10 x the oracle's own fp32-vs-fp64 difference, 5e-3 at least -- it hunts gross errors; tolerances are the test suite's job).  Run under `timeout`.   python tools/soak_tuned.py 120"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, hode, bench
dev = torch.device("cuda")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end, n, n64 = time.time() + budget, 0, 0


def relnorm(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


while time.time() < t_end:
    H = int(rng.choice([7, 16, 33, 64, 64, 64])); L = int(rng.integers(1, 5)); T = int(rng.choice([2, 5, 13, 31, 61]))
    n_sets = int(rng.choice([1, 1, 1, 2, 3])); per = int(rng.integers(1, 40) if rng.random() < 0.3 else rng.integers(40, 1500))
    B = per * n_sets
    rk4 = bool(rng.random() < 0.3); gode = bool(rng.random() < 0.5)
    method = hode.METHOD_RK4 if rk4 else hode.METHOD_DP54
    rtol = float(rng.choice([1e-4, 1e-6, 1e-8]))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    P = hode.n_params(H, L)
    nn = (torch.randn(n_sets * P, generator=g) * (0.7 * (2.0 / (2 * H)) ** 0.5)).to(dev)
    ode = bench.ODE_DEFAULT.repeat(n_sets).to(dev)
    x0, t, meal, tv = bench.synth_cohort(B, int(rng.integers(1000)))
    x0 = (x0 * (0.5 + torch.rand(B, 6, generator=g))).to(dev)
    nan_row = int(rng.integers(B)) if rng.random() < 0.2 else -1
    if nan_row >= 0: x0[nan_row] = float("nan")
    t = t[:T].clone() * float(rng.choice([1.0, 3.0]))
    if T > 4 and rng.random() < 0.3:                       # repeated grid times: at the start, in the middle
        t[1] = t[0]
        k = int(rng.integers(2, T - 1)); t[k + 1:] -= (t[k + 1] - t[k]).clone(); 
    if rng.random() < 0.2: t = t.repeat(B, 1) * (1.0 + 0.1 * torch.rand(B, 1, generator=g))
    t, meal, tv = t.contiguous().to(dev), meal[:, :T].contiguous().to(dev), tv[:, :T].contiguous().to(dev)
    kw = dict(method=method, rtol=rtol, atol=rtol * 1e-2, n_sets=n_sets)
    if rk4:
        ms = T - 1
    else:
        probe = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L, max_steps=400, **kw)
        ms = max(int(np.percentile(probe.nsteps.cpu().numpy(), float(rng.choice([70, 100])))) + int(rng.integers(0, 3)), 1)
    sol = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L, want_tape=True, max_steps=ms, **kw)
    gy = torch.randn(B, T, 6, device=dev, generator=torch.Generator(dev).manual_seed(n)) / (B * T)
    a = hode.solve_bwd(sol, gy, want_gode=gode)
    b = hode.solve_bwd(sol, gy, want_gode=gode)
    ok_rows = sol.status == 0
    assert torch.isfinite(sol.y[ok_rows]).all()
    for u, v in zip(a, b):
        if u is not None:
            assert torch.equal(u.view(torch.int32), v.view(torch.int32)), ("not reproducible", H, L, T, B, n_sets, rk4, int((sol.status != 0).sum()))   # (bit patterns: NaN == NaN)
    ymax = float(sol.y.nan_to_num(posinf=0.0, neginf=0.0).abs().max())
    # (a trajectory that blows up passes through huge FINITE states before it fails: their records are on the tape, their products
    #  overflow fp32 in the SHARED parameter gradient -- in the oracle as well; only moderate batches are held to a finite gradient)
    tame = ok_rows & (sol.y.nan_to_num(posinf=3e38, neginf=3e38).abs().amax((1, 2)) < 1e12)
    assert torch.isfinite(a[0][tame]).all() and (ymax > 1e12 or torch.isfinite(a[1]).all()), ("not finite", H, L, T, B, n_sets, rk4, int((~ok_rows).sum()), ymax)
    lo = int(rng.integers(0, max(per - 5, 1))); sl = slice(lo, lo + min(300, per - lo))
    tt = t[sl].contiguous() if t.dim() == 2 else t
    small = hode.solve_fwd(x0[sl].contiguous(), tt, meal[sl].contiguous(), tv[sl].contiguous(), None, ode[:17], nn[:P], H, L,
                           method=method, rtol=rtol, atol=rtol * 1e-2, max_steps=ms)
    if not (torch.equal(small.y.nan_to_num(), sol.y[sl].nan_to_num()) and torch.equal(small.status, sol.status[sl])):
        dy = (small.y.nan_to_num() - sol.y[sl].nan_to_num()).abs()
        bad = torch.nonzero(dy.amax((1, 2)) > 0).flatten()
        print("forward differs", dict(H=H, L=L, T=T, B=B, n_sets=n_sets, rk4=rk4, rtol=rtol, ms=ms, tdim=t.dim(), nan_row=nan_row, lo=lo, nbad=int(bad.numel()),
              first_bad=bad[:5].tolist(), maxdiff=float(dy.max()), st_small=small.status[bad[:5]].tolist(), st_big=sol.status[sl][bad[:5]].tolist(),
              t=t.flatten()[:T].tolist()), flush=True)
        raise SystemExit(1)
    tag = ""
    # (at loose tolerances the gradient follows the step sequence; a trajectory that wanders off to 1e3 is ill-conditioned -- there the
    #  kernel, the fp32 oracle and the fp64 oracle differ from each other by 1e-3..1e-2: tools/adjoint_conditioning.py)
    if rng.random() < 0.4 and nan_row < 0 and rk4 and ymax < 500.0:        # (adaptive: the discrete gradient follows the step sequence, 1e-4..1e-3 between ANY two arithmetics)
        # values: the first trajectories of set 0 on their own through the kernels, against the ORACLE in fp64 -- held to five times what
        # the oracle's own fp32 run differs from it by (a randomly initialised network over a long horizon can be ill-conditioned: ReLU
        # switches flip with the last bit), 1e-3 at least
        from oracle import oracle as O
        nb = min(per, 12)
        f64 = lambda v: v.detach().cpu().numpy().astype(np.float64)      # noqa: E731
        ts_ = t[:nb].contiguous() if t.dim() == 2 else t
        ks = hode.solve_fwd(x0[:nb].contiguous(), ts_, meal[:nb].contiguous(), tv[:nb].contiguous(), None, ode[:17], nn[:P], H, L,
                            method=method, rtol=rtol, atol=rtol * 1e-2, want_tape=True, max_steps=ms)
        if bool((ks.status == 0).all()):
            kg = hode.solve_bwd(ks, gy[:nb].contiguous(), want_gode=False)
            res = {}
            for dt in (np.float32, np.float64):
                r = O.solve(f64(x0[:nb]), f64(ts_), f64(meal[:nb]), f64(tv[:nb]), None, f64(ode[:17]), f64(nn[:P]), H, L, method=method,
                            rtol=rtol, atol=rtol * 1e-2, dtype=dt, want_tape=True, max_steps=ms)
                res[dt] = O.solve_bwd(r, f64(gy[:nb]).astype(dt), want_gode=False) if int(r.status.max()) == 0 else None
            if res[np.float32] is not None and res[np.float64] is not None:
                rn = lambda u, v: float(np.linalg.norm(np.asarray(u, np.float64) - v) / (np.linalg.norm(v) + 1e-300))      # noqa: E731
                e32 = max(rn(res[np.float32][0], res[np.float64][0]), rn(res[np.float32][1], res[np.float64][1]))
                ek = max(rn(f64(kg[0]), res[np.float64][0]), rn(f64(kg[1]), res[np.float64][1]))
                if not ek < max(10 * e32, 5e-3):
                    # a trajectory ON a ReLU kink?  (one unit within the last bits of zero: the kernel's fp32 arithmetic may land on the
                    # other side of it than the oracle's -- ~1 % of a small batch's gradient.)  Then the fp64 oracle's own gradient
                    # jumps under a 1e-6 / 1e-5 perturbation of the initial states.
                    def o64(xs):
                        r = O.solve(xs, f64(ts_), f64(meal[:nb]), f64(tv[:nb]), None, f64(ode[:17]), f64(nn[:P]), H, L, method=method, rtol=rtol,
                                    atol=rtol * 1e-2, dtype=np.float64, want_tape=True, max_steps=ms)
                        return O.solve_bwd(r, f64(gy[:nb]), want_gode=False)[1]
                    jump = max(rn(o64(f64(x0[:nb]) * (1 + e)), res[np.float64][1]) for e in (1e-6, -1e-6, 1e-5, -1e-5))
                    if jump > 1e-3:
                        tag = f" (on a ReLU kink: the fp64 gradient moves {jump:.1e} under a 1e-5 perturbation; kernel {ek:.1e})"
                        ek = 0.0
                    else:
                        torch.save(dict(x0=x0[:nb].cpu(), t=ts_.cpu(), meal=meal[:nb].cpu(), tv=tv[:nb].cpu(), ode=ode[:17].cpu(), nn=nn[:P].cpu(),
                                        gy=gy[:nb].cpu(), H=H, L=L, method=method, rtol=rtol, ms=ms, kgx0=kg[0].cpu(), kgnn=kg[1].cpu(),
                                        rgx0=res[np.float64][0], rgnn=res[np.float64][1], ky=ks.y.cpu()), os.path.join(ROOT, "gpurun_out", "soak_fail.pt"))
                assert ek < max(10 * e32, 5e-3), ("adjoint vs fp64 oracle", ek, e32, H, L, T, nb, rk4, rtol)
                n64 += 1
                tag = tag or f" vs fp64 oracle: {ek:.1e} (oracle fp32: {e32:.1e})"
    n += 1
    print(f"case {n}: H={H} L={L} T={T} B={B} sets={n_sets} {'RK4' if rk4 else 'DP54 rtol=%g' % rtol} max_steps={ms} failed={int((~ok_rows).sum())} max|y|={ymax:.1e} gode={gode} ok{tag}", flush=True)
torch.cuda.synchronize()
print(f"{n} cases ({n64} checked against the fp64 oracle), no hang, no mismatch")
