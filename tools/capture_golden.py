#!/usr/bin/env python3
"""Capture golden input/output vectors from the reference implementation.

Runs ONLY in the build container (needs /root/reference, read-only).  It imports the
reference's `models` package, evaluates the hot path on small seeded inputs and writes
*data* (inputs + expected outputs, no source text) to tests/golden/*.npz|json.

Fixtures:
  G0  weights        seeded state_dict with a non-zero MLP output layer
  G1  ODECore.forward            models/ode_core.py:81-166      (fp32 and .double())
  G2  NNResidual.forward         models/nn_residual.py:100-151
  G3  HybridODENN.ode_residual   models/hybrid_ode_nn.py:108-134 (batched and (6,)/0-dim path)
  G4  HybridODENN.forward        models/hybrid_ode_nn.py:136-261 at rk45 tight / rk45 default /
      DOP853 default tolerances, plus an fp64 interval-by-interval converged solve of the
      reference's own `.double().ode_residual`
  G5  HybridODENN.loss           models/hybrid_ode_nn.py:263-351 (values + NN gradients)
  G6  VariationalParameters      models/bayes.py:65-175 (KL, one sample)
  G7  finite differences of the reference forward w.r.t. a few weights / x0 entries
      (adjoint spot-check, SURVEY 8c "(ii)")

Usage:  python tools/capture_golden.py            (takes a few minutes, single thread)
"""
import copy
import json
import os
import sys
import time

import numpy as np
import torch
from scipy.integrate import solve_ivp

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)

from models.hybrid_ode_nn import HybridODENN  # noqa: E402
from models.ode_core import ODECore  # noqa: E402
from models.nn_residual import NNResidual  # noqa: E402
from models.bayes import VariationalParameters  # noqa: E402

torch.set_num_threads(1)
CPU = torch.device("cpu")

ODE_NAMES = ["a_GI", "k_I", "rho", "G_b", "I_b", "E_max", "EC_50", "Glu_b", "V_max", "K_m",
             "k_L", "k_GE0", "IGD_50", "g", "p_7", "p_8", "p_9"]


def seeded_model(hidden=64, layers=4, seed=0):
    """G0 recipe: seed, default init, then output layer ~ N(0, 0.01)."""
    torch.manual_seed(seed)
    m = HybridODENN(nn_hidden=hidden, nn_layers=layers, device=CPU)
    with torch.no_grad():
        m.nn_residual.network[-1].weight.normal_(0, 0.01)
        m.nn_residual.network[-1].bias.normal_(0, 0.01)
    return m


def flat_nn(m):
    return torch.cat([p.detach().reshape(-1) for p in m.nn_residual.parameters()]).numpy()


def ode_vec(m):
    return np.array([float(getattr(m.ode_core, n)) for n in ODE_NAMES], dtype=np.float32)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path)/1024:.1f} KiB")


# ----------------------------------------------------------------------------- G0
def g0():
    m = seeded_model()
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    save("g0_weights_h64_l4.npz", nn_flat=flat_nn(m), ode=ode_vec(m),
         **{k.replace(".", "__"): v for k, v in sd.items()})
    m2 = seeded_model(hidden=32, layers=2, seed=1)
    save("g0_weights_h32_l2.npz", nn_flat=flat_nn(m2), ode=ode_vec(m2))
    with open(os.path.join(OUT, "g0_state_dict_keys.json"), "w") as f:
        json.dump({"keys": list(m.state_dict().keys()),
                   "shapes": [list(v.shape) for v in m.state_dict().values()],
                   "n_params": sum(p.numel() for p in m.parameters())}, f, indent=1)
    return m, m2


# ----------------------------------------------------------------------------- G1-G3
def rhs_inputs():
    g = torch.Generator().manual_seed(123)
    base = torch.tensor([5., 60., 80., 10., 0., 1.])
    physio = base * (1 + 0.05 * torch.randn(40, 6, generator=g))
    zs = torch.randn(16, 6, generator=g)
    special = torch.tensor([[20., 500., 200., 100., 2., 5.], [2., 10., 10., 5., 0., .1],
                            [5., 100., 50., 20., 0., 1.], [8., 150., 40., 30., .5, 1.2],
                            [5., 100., 50., 20., 0., 1.], [0., 0., 0., 0., 0., 0.],
                            [-1., -2., 3., -4., .5, -.5], [50., 600., 800., 100., 0., 10.]])
    x = torch.cat([physio, zs, special])          # 64 states
    n = x.shape[0]
    t = torch.rand(n, generator=g) * 20
    meal = torch.rand(n, generator=g) * 10
    meal[::3] = 0
    tvns = (torch.rand(n, generator=g) > .5).float()
    gd = torch.tensor([0., 500., 1000., 250.])[torch.arange(n) % 4]
    return x, t, meal, tvns, gd


def g123(m, m2):
    x, t, meal, tvns, gd = rhs_inputs()
    out = dict(x=x.numpy(), t=t.numpy(), meal=meal.numpy(), tvns=tvns.numpy(), gd=gd.numpy())
    md = copy.deepcopy(m).double()
    with torch.no_grad():
        for tag, ext in (("nogd", {"meal": meal, "tVNS": tvns}),
                         ("gd", {"meal": meal, "tVNS": tvns, "GD": gd}),
                         ("none", None)):
            out[f"ode_f32_{tag}"] = m.ode_core(t, x, ext).numpy()
            extd = None if ext is None else {k: v.double() for k, v in ext.items()}
            out[f"ode_f64_{tag}"] = md.ode_core(t.double(), x.double(), extd).numpy()
            out[f"rhs_f32_{tag}"] = m.ode_residual(t, x, ext).numpy()
            out[f"rhs_f64_{tag}"] = md.ode_residual(t.double(), x.double(), extd).numpy()
        out["nn_f32"] = m.nn_residual(t, x, x[:, 3], tvns).numpy()
        out["nn_f64"] = md.nn_residual(t.double(), x.double(), x[:, 3].double(), tvns.double()).numpy()
        out["nn_f32_h32l2"] = m2.nn_residual(t, x, x[:, 3], tvns).numpy()
        out["rhs_f32_h32l2"] = m2.ode_residual(t, x, {"meal": meal, "tVNS": tvns}).numpy()
        # the (6,)/0-dim path used by ode_func (hybrid_ode_nn.py:206-237)
        single = [m.ode_residual(t[i], x[i], {"meal": meal[i], "tVNS": tvns[i]}).numpy()
                  for i in range(8)]
        out["rhs_f32_single8"] = np.stack(single)
    # VJP goldens for K5: d(sum_i w_i f_i)/d(x, nn params) by reference autograd
    gw = torch.Generator().manual_seed(7)
    w = torch.randn(x.shape[0], 6, generator=gw)
    xr = x.clone().requires_grad_(True)
    m.zero_grad()
    f = m.ode_residual(t, xr, {"meal": meal, "tVNS": tvns})
    (f * w).sum().backward()
    out["vjp_w"] = w.numpy()
    out["vjp_gx"] = xr.grad.numpy()
    out["vjp_gnn"] = torch.cat([p.grad.reshape(-1) for p in m.nn_residual.parameters()]).numpy()
    m.zero_grad()
    save("g123_rhs.npz", **out)


# ----------------------------------------------------------------------------- G-act
def g_act():
    """The activations NNResidual offers besides ReLU (models/nn_residual.py:50-56: tanh, elu, leaky_relu(0.1)).  HybridODENN never
    passes one (hybrid_ode_nn.py:57-58), so they are reached by replacing `model.nn_residual`; what is captured is what such a
    model computes: NNResidual.forward, ode_residual (fp32 and .double()), the autograd VJP, and a converged rk45 trajectory."""
    x, t, meal, tvns, gd = rhs_inputs()
    for act in ("tanh", "elu", "leaky_relu"):
        torch.manual_seed(11)
        m = HybridODENN(nn_hidden=16, nn_layers=3, device=CPU)
        m.nn_residual = NNResidual(input_dim=9, hidden_dim=16, output_dim=6, n_layers=3, activation=act)
        with torch.no_grad():
            m.nn_residual.network[-1].weight.normal_(0, 0.05)
            m.nn_residual.network[-1].bias.normal_(0, 0.01)
            for lin in [l for l in m.nn_residual.network[:-1] if isinstance(l, torch.nn.Linear)]:
                lin.bias.normal_(0, 0.3)                      # pre-activations on both sides of zero (ELU / leaky slopes, tanh's knee)
        md = copy.deepcopy(m).double()
        out = dict(x=x.numpy(), t=t.numpy(), meal=meal.numpy(), tvns=tvns.numpy(), nn_flat=flat_nn(m), ode=ode_vec(m))
        with torch.no_grad():
            out["nn_f32"] = m.nn_residual(t, x, x[:, 3], tvns).numpy()
            out["nn_f64"] = md.nn_residual(t.double(), x.double(), x[:, 3].double(), tvns.double()).numpy()
            out["rhs_f32"] = m.ode_residual(t, x, {"meal": meal, "tVNS": tvns}).numpy()
            out["rhs_f64"] = md.ode_residual(t.double(), x.double(), {"meal": meal.double(), "tVNS": tvns.double()}).numpy()
        gw = torch.Generator().manual_seed(7)
        w = torch.randn(x.shape[0], 6, generator=gw).double()
        xr = x.double().clone().requires_grad_(True)
        md.zero_grad()
        (md.ode_residual(t.double(), xr, {"meal": meal.double(), "tVNS": tvns.double()}) * w).sum().backward()
        out["vjp_w"], out["vjp_gx_f64"] = w.numpy(), xr.grad.numpy()
        out["vjp_gnn_f64"] = torch.cat([p.grad.reshape(-1) for p in md.nn_residual.parameters()]).numpy()
        x0, tt, ml, tv = cohort(4, 37, 3.0, "pulses", 21)               # 5-min grid over 3 h, unit meals at indices 6 and 30
        with torch.no_grad():
            out["traj_x0"], out["traj_t"], out["traj_meal"], out["traj_tvns"] = x0.numpy(), tt.numpy(), ml.numpy(), tv.numpy()
            out["traj_y_rk45_tight"] = m.forward(x0, tt, {"meal": ml, "tVNS": tv}, solver="rk45", rtol=1e-10, atol=1e-12).numpy()
        save(f"g_act_{act}.npz", **out)


# ----------------------------------------------------------------------------- G4
def cohort(B, T, t_end, kind, seed):
    g = torch.Generator().manual_seed(seed)
    base = torch.tensor([5., 60., 80., 10., 0., 1.])
    x0 = base * (1 + 0.05 * torch.randn(B, 6, generator=g))
    t = torch.linspace(0, t_end, T)
    meal = torch.zeros(B, T)
    tvns = torch.zeros(B, T)
    if kind == "pulses":
        meal[:, 6] = 1.0
        meal[:, 30] = 1.0
    elif kind == "const":
        meal = torch.rand(B, generator=g) * 0.5      # dim()==1: constant per patient
        tvns = (torch.rand(B, generator=g) > .5).float()
    elif kind == "rand":
        meal = torch.rand(B, T, generator=g) * 2
        tvns = (torch.rand(B, T, generator=g) > .7).float()
    return x0, t, meal, tvns


def converged_f64(m, x0, t, meal, tvns):
    """SURVEY F7 recipe: fp64 solve of the reference's own .double() RHS, broken at grid points
    (forcing is piecewise linear there), rtol 1e-12."""
    md = copy.deepcopy(m).double()
    B, T = x0.shape[0], t.shape[-1]
    tt = t.double().numpy()
    out = np.zeros((B, T, 6))
    for b in range(B):
        tb = tt[b] if tt.ndim == 2 else tt
        y = x0[b].double().numpy().copy()
        out[b, 0] = y
        for k in range(T - 1):
            t0, t1 = float(tb[k]), float(tb[k + 1])

            def inp(v, tau):
                if v.dim() == 2:
                    a = (tau - t0) / (t1 - t0)
                    return float(v[b, k]) + a * (float(v[b, k + 1]) - float(v[b, k]))
                return float(v[b])

            def f(tau, yy):
                with torch.no_grad():
                    ext = {"meal": torch.tensor(inp(meal, tau), dtype=torch.float64),
                           "tVNS": torch.tensor(inp(tvns, tau), dtype=torch.float64)}
                    return md.ode_residual(torch.tensor(tau, dtype=torch.float64),
                                           torch.tensor(yy, dtype=torch.float64), ext).numpy()
            sol = solve_ivp(f, (t0, t1), y, method="DOP853", rtol=1e-12, atol=1e-14)
            y = sol.y[:, -1]
            out[b, k + 1] = y
    return out


def g4(m, m2):
    cases = [("t61_zero", 8, 61, 5.0, "zero", 11), ("t61_pulses", 8, 61, 5.0, "pulses", 12),
             ("t241_pulses", 8, 241, 20.0, "pulses", 13), ("t241_zero", 4, 241, 20.0, "zero", 14),
             ("t61_const", 8, 61, 5.0, "const", 15), ("t61_rand", 8, 61, 5.0, "rand", 16)]
    for name, B, T, t_end, kind, seed in cases:
        t0 = time.time()
        x0, t, meal, tvns = cohort(B, T, t_end, kind, seed)
        ext = {"meal": meal, "tVNS": tvns}
        with torch.no_grad():
            tight = m.forward(x0, t, ext, solver="rk45", rtol=1e-10, atol=1e-12).numpy()
            dflt = m.forward(x0, t, ext, solver="rk45", rtol=1e-6, atol=1e-8).numpy()
            dop = m.forward(x0[:2], t, {k: v[:2] for k, v in ext.items()}).numpy()  # 'dopri5'->DOP853
        conv = converged_f64(m, x0[:4], t, meal[:4], tvns[:4])
        save(f"g4_{name}.npz", x0=x0.numpy(), t=t.numpy(), meal=meal.numpy(), tvns=tvns.numpy(),
             y_rk45_tight=tight, y_rk45_default=dflt, y_dop853_default_first2=dop,
             y_f64_converged_first4=conv)
        print(f"    {name}: {time.time()-t0:.1f}s  tight-vs-conv max rel "
              f"{np.max(np.abs(tight[:4]-conv)/(np.abs(conv)+1e-3)):.2e}  default-vs-conv "
              f"{np.max(np.abs(dflt[:4]-conv)/(np.abs(conv)+1e-3)):.2e}")
    # batched (per-patient) time grids + the small (32,2) network + z-scored states
    g = torch.Generator().manual_seed(21)
    B, T = 4, 21
    x0 = torch.randn(B, 6, generator=g)
    t = torch.stack([torch.linspace(0, 1 + b, T) for b in range(B)])
    meal = torch.rand(B, T, generator=g) * 10
    tvns = torch.rand(B, T, generator=g)
    with torch.no_grad():
        tight = m2.forward(x0, t, {"meal": meal, "tVNS": tvns}, solver="rk45", rtol=1e-10, atol=1e-12).numpy()
    conv = converged_f64(m2, x0, t, meal, tvns)
    save("g4_batched_t_h32l2.npz", x0=x0.numpy(), t=t.numpy(), meal=meal.numpy(), tvns=tvns.numpy(),
         y_rk45_tight=tight, y_f64_converged_first4=conv)
    # data/4gi_dataset.csv -> GlucoseDataset-style first window per subject
    # (train/train_hybrid.py:72-139: columns, ge=0, ffa=1, z-score over all windows, hours)
    import pandas as pd
    df = pd.read_csv(os.path.join(REF, "data", "4gi_dataset.csv"))
    df["ge"] = 0.0
    df["ffa"] = 1.0
    cols = ["glucose_mmol_L", "insulin_pmol_L", "glucagon_pmol_L", "glp1_pmol_L", "ge", "ffa"]
    wins, meals, times = [], [], []
    for sid, sub in df.groupby("subject_id"):
        sub = sub.iloc[:61]
        wins.append(sub[cols].values)
        meals.append(sub["meal_indicator"].values.astype(np.float32))
        times.append((sub["time_minutes"].values / 60.0).astype(np.float32))
    allst = np.concatenate(wins)
    mean, std = allst.mean(0), allst.std(0) + 1e-6
    obs = torch.tensor(np.stack([(w - mean) / std for w in wins]), dtype=torch.float32)
    meal = torch.tensor(np.stack(meals))
    t = torch.tensor(times[0])
    tvns = torch.zeros_like(meal)
    with torch.no_grad():
        tight = m.forward(obs[:, 0], t, {"meal": meal, "tVNS": tvns}, solver="rk45", rtol=1e-10, atol=1e-12).numpy()
    conv = converged_f64(m, obs[:4, 0], t, meal[:4], tvns[:4])
    save("g4_4gi_csv.npz", x0=obs[:, 0].numpy(), t=t.numpy(), meal=meal.numpy(), tvns=tvns.numpy(),
         observations=obs.numpy(), y_rk45_tight=tight, y_f64_converged_first4=conv)


# ----------------------------------------------------------------------------- G5
def g5():
    res = {}
    # tests/test_gradient_correctness.py:65-114 batch
    torch.manual_seed(0); np.random.seed(0)
    model = HybridODENN(nn_hidden=32, nn_layers=2, use_variational=False, device="cpu")
    B, T = 2, 5
    batch = {"initial_state": torch.randn(B, 6), "observations": torch.randn(B, T, 6),
             "time_points": torch.linspace(0, 1, T).unsqueeze(0).expand(B, -1),
             "external_inputs": {"meal": torch.rand(B, T) * 10, "tVNS": torch.rand(B, T)}}
    rng = torch.get_rng_state()
    perm = torch.randperm(len(batch["time_points"]))
    torch.set_rng_state(rng)
    loss = model.loss(batch, lambda1=1.0, lambda2=0.1, use_physics_loss=True)
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.nn_residual.parameters()]).numpy()
    with torch.no_grad():
        pred = model.forward(batch["initial_state"], batch["time_points"], batch["external_inputs"])
        data = torch.nn.functional.mse_loss(pred, batch["observations"]).item()
        reg = model.nn_residual.regularization_loss(l2_weight=0.1).item()
    save("g5_loss_b2_t5.npz", nn_flat=flat_nn(model), ode=ode_vec(model),
         x0=batch["initial_state"].numpy(), obs=batch["observations"].numpy(),
         t=batch["time_points"].contiguous().numpy(), meal=batch["external_inputs"]["meal"].numpy(),
         tvns=batch["external_inputs"]["tVNS"].numpy(), perm=perm.numpy(), total=np.float64(loss.item()),
         data=np.float64(data), reg=np.float64(reg), lambda1=1.0, lambda2=0.1, grads=grads,
         pred_dop853_default=pred.numpy())
    res["b2_t5"] = dict(total=loss.item(), data=data, reg=reg)
    # same but with a non-zero output layer so that the physics term is non-trivial
    model = seeded_model(hidden=32, layers=2, seed=3)
    torch.manual_seed(5)
    B, T = 3, 10
    batch = {"initial_state": torch.randn(B, 6), "observations": torch.randn(B, T, 6),
             "time_points": torch.linspace(0, 1, T),
             "external_inputs": {"meal": torch.rand(B, T), "tVNS": torch.zeros(B, T)}}
    rng = torch.get_rng_state()
    perm = torch.randperm(len(batch["time_points"]))
    torch.set_rng_state(rng)
    loss = model.loss(batch, lambda1=0.5, lambda2=0.1)
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.nn_residual.parameters()]).numpy()
    with torch.no_grad():
        pred = model.forward(batch["initial_state"], batch["time_points"], batch["external_inputs"])
        data = torch.nn.functional.mse_loss(pred, batch["observations"]).item()
        reg = model.nn_residual.regularization_loss(l2_weight=0.1).item()
    save("g5_loss_b3_t10_shared.npz", nn_flat=flat_nn(model), ode=ode_vec(model),
         x0=batch["initial_state"].numpy(), obs=batch["observations"].numpy(),
         t=batch["time_points"].numpy(), meal=batch["external_inputs"]["meal"].numpy(),
         tvns=batch["external_inputs"]["tVNS"].numpy(), perm=perm.numpy(), total=np.float64(loss.item()),
         data=np.float64(data), reg=np.float64(reg), lambda1=0.5, lambda2=0.1, grads=grads,
         pred_dop853_default=pred.numpy())
    res["b3_t10"] = dict(total=loss.item(), data=data, reg=reg)
    print("   ", res)


def g5_tight():
    """G5 at CONVERGED tolerances.  loss() calls self.forward(...) with its defaults twice
    (models/hybrid_ode_nn.py:291,320): 'dopri5' = DOP853 at 1e-6 / 1e-8, itself ~1e-2 away from the converged solution.  Here
    the defaults are overridden IN THE CAPTURE PROCESS -- an instance attribute `forward` bound over the reference's own method
    with solver='rk45', rtol=1e-10, atol=1e-12 -- so that the captured loss, its three components and its gradients are those
    of the converged trajectories: a fixture that can see a wrong m/n factor or a missing term at 1e-5."""
    import functools
    for tag, seed_model, seed_batch, B, T, batched, lam1, lam2, meal_scale in [
            ("b2_t5", None, 0, 2, 5, True, 1.0, 0.1, 10.0), ("b3_t10_shared", 3, 5, 3, 10, False, 0.5, 0.1, 1.0),
            ("b4_t25_shared", 7, 11, 4, 25, False, 2.0, 0.3, 1.0)]:          # T >= 20: n = 20 sampled indices of 25, m / n matters
        if seed_model is None:
            torch.manual_seed(0); np.random.seed(0)
            model = HybridODENN(nn_hidden=32, nn_layers=2, use_variational=False, device="cpu")
            with torch.no_grad():                      # the default zero output layer makes the MLP invisible: give it one
                model.nn_residual.network[-1].weight.normal_(0, 0.01)
                model.nn_residual.network[-1].bias.normal_(0, 0.01)
        else:
            model = seeded_model(hidden=32, layers=2, seed=seed_model)
        torch.manual_seed(seed_batch)
        tp = torch.linspace(0, 1, T)
        batch = {"initial_state": torch.randn(B, 6), "observations": torch.randn(B, T, 6),
                 "time_points": tp.unsqueeze(0).expand(B, -1) if batched else tp,
                 "external_inputs": {"meal": torch.rand(B, T) * meal_scale, "tVNS": torch.rand(B, T) if batched else torch.zeros(B, T)}}
        model.forward = functools.partial(model.forward, solver="rk45", rtol=1e-10, atol=1e-12)
        rng = torch.get_rng_state()
        perm = torch.randperm(len(batch["time_points"]))
        torch.set_rng_state(rng)
        # the physics component is only logged by the reference: record every F.mse_loss the call makes (the first is the data
        # term, the others the per-index physics terms, models/hybrid_ode_nn.py:294,330) instead of differencing fp32 totals
        terms = []
        orig_mse = torch.nn.functional.mse_loss

        def recording_mse(*a, **k):
            v = orig_mse(*a, **k)
            terms.append(float(v.detach().double()))
            return v
        torch.nn.functional.mse_loss = recording_mse
        try:
            loss = model.loss(batch, lambda1=lam1, lambda2=lam2, use_physics_loss=True)
        finally:
            torch.nn.functional.mse_loss = orig_mse
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in model.nn_residual.parameters()]).numpy()
        with torch.no_grad():
            pred = model.forward(batch["initial_state"], batch["time_points"], batch["external_inputs"])
            data = torch.nn.functional.mse_loss(pred, batch["observations"]).item()
            reg = model.nn_residual.regularization_loss(l2_weight=lam2).item()
        n_phys = min(20, len(batch["time_points"]))
        assert len(terms) == 1 + n_phys and abs(terms[0] - data) < 1e-6 * abs(data), (len(terms), terms[0], data)
        physics = sum(terms[1:]) / n_phys
        assert abs(data + lam1 * physics + lam2 * reg - loss.item()) < 2e-6 * abs(loss.item())
        save(f"g5t_loss_{tag}.npz", nn_flat=flat_nn(model), ode=ode_vec(model), x0=batch["initial_state"].numpy(),
             obs=batch["observations"].numpy(), t=batch["time_points"].contiguous().numpy(),
             meal=batch["external_inputs"]["meal"].numpy(), tvns=batch["external_inputs"]["tVNS"].numpy(), perm=perm.numpy(),
             total=np.float64(loss.item()), data=np.float64(data), reg=np.float64(reg), physics=np.float64(physics),
             lambda1=lam1, lambda2=lam2, grads=grads, pred_rk45_tight=pred.numpy(), physics_terms=np.array(terms[1:]))
        print("   ", tag, dict(total=loss.item(), data=data, physics=physics, reg=reg, gradnorm=float(np.linalg.norm(grads))))


# ----------------------------------------------------------------------------- G6
def g6():
    torch.manual_seed(0)
    model = HybridODENN(nn_hidden=16, nn_layers=2, use_variational=True, device=CPU)
    vp = model.variational_params
    names = list(vp.param_shapes.keys())
    kl0 = vp.kl_divergence().item()
    torch.manual_seed(1)
    with torch.no_grad():
        for n in names:
            vp.means[n].add_(0.05 * torch.randn_like(vp.means[n]))
            vp.log_stds[n].add_(0.1 * torch.randn_like(vp.log_stds[n]))
    kl1 = vp.kl_divergence().item()
    mu, ls = vp.get_flattened_params()
    torch.manual_seed(2)
    samp = vp.sample(1)[0]
    torch.manual_seed(2)
    eps = {n: torch.randn_like(vp.means[n]) for n in names}
    arrs = {"mu_flat": mu.detach().numpy(), "log_sigma_flat": ls.detach().numpy(),
            "kl_init": np.float64(kl0), "kl_perturbed": np.float64(kl1)}
    for n in names:
        arrs["mean__" + n] = vp.means[n].detach().numpy()
        arrs["logstd__" + n] = vp.log_stds[n].detach().numpy()
        arrs["sample__" + n] = samp[n].detach().numpy()
        arrs["eps__" + n] = eps[n].numpy()
    save("g6_vi.npz", **arrs)
    with open(os.path.join(OUT, "g6_vi_names.json"), "w") as f:
        json.dump({"names": names, "sorted": sorted(names), "latent_dims": int(mu.numel())}, f, indent=1)
    print(f"    KL init {kl0:.4f} perturbed {kl1:.4f} latent {mu.numel()}")


# ----------------------------------------------------------------------------- G7
def g7(m):
    """Central finite differences of the *reference* forward (rk45 @ 1e-10/1e-12) of the scalar
    L = sum(c * y) w.r.t. a few NN weights and x0 entries.  fp32 output quantisation limits the
    accuracy to ~1e-3..1e-4 relative."""
    B, T = 2, 21
    x0, t, meal, tvns = cohort(B, T, 2.0, "rand", 31)
    g = torch.Generator().manual_seed(32)
    c = torch.randn(B, T, 6, generator=g)
    ext = {"meal": meal, "tVNS": tvns}

    def L(model, x):
        with torch.no_grad():
            y = model.forward(x, t, ext, solver="rk45", rtol=1e-10, atol=1e-12)
        return float((y.double() * c.double()).sum())

    params = list(m.nn_residual.parameters())
    offs = np.cumsum([0] + [p.numel() for p in params])
    picks = [(0, 5), (0, 100), (1, 3), (2, 70), (2, 2000), (4, 999), (6, 1234), (7, 10), (8, 17), (8, 300), (9, 2)]
    fd_idx, fd_val = [], []
    for pi, ei in picks:
        p = params[pi]
        flat = p.data.view(-1)
        old = flat[ei].item()
        eps = max(1e-3, 2e-2 * abs(old))
        flat[ei] = old + eps; lp = L(m, x0)
        flat[ei] = old - eps; lm = L(m, x0)
        flat[ei] = old
        fd_idx.append(int(offs[pi] + ei)); fd_val.append((lp - lm) / (2 * eps))
    fdx = np.zeros((B, 6))
    for b in range(B):
        for i in range(6):
            eps = max(1e-3, 1e-2 * abs(x0[b, i].item()))
            xp = x0.clone(); xp[b, i] += eps
            xm = x0.clone(); xm[b, i] -= eps
            fdx[b, i] = (L(m, xp) - L(m, xm)) / (float(xp[b, i]) - float(xm[b, i]))
    save("g7_fd_reference.npz", x0=x0.numpy(), t=t.numpy(), meal=meal.numpy(), tvns=tvns.numpy(),
         c=c.numpy(), fd_param_index=np.array(fd_idx), fd_param_grad=np.array(fd_val), fd_x0_grad=fdx)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    if len(sys.argv) > 1 and sys.argv[1] == "g5t":          # only the tight-tolerance loss fixtures
        print("G5 tight"); g5_tight()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gact":         # only the non-ReLU activation fixtures
        print("G-act"); g_act()
        sys.exit(0)
    print("G0"); m, m2 = g0()
    print("G1-G3"); g123(m, m2)
    print("G5"); g5()
    print("G5 tight"); g5_tight()
    print("G-act"); g_act()
    print("G6"); g6()
    print("G7"); g7(m)
    print("G4"); g4(m, m2)
    print(f"done in {time.time()-t0:.0f}s")
