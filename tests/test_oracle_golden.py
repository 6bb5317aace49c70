"""Pin the oracle (oracle/hode_oracle.c) against vectors captured from the reference itself
(tools/capture_golden.py, run in the build container; tests/golden/*.npz).  CPU only.

Tolerances:
  RHS fp64      1e-12 relative  (vs reference `.double()` evaluation)
  RHS fp32      5e-6            (summation order of the 64-wide dot products differs from BLAS)
  trajectories  fp64 oracle vs fp64-converged reference solve: 1e-6 (BASELINE bar is 1e-5)
                fp32 oracle vs converged: 1e-4 (BASELINE bar is 1e-3)
"""
import os

import numpy as np
import pytest

from oracle import oracle as O


def rel(a, b, floor=1e-3):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / (np.abs(b) + floor)))


def test_tableau_matches_scipy():
    """scipy/integrate/_ivp/rk.py:377-401 constants (RK45.C/A/B/E/P)."""
    from scipy.integrate._ivp.rk import RK45
    C, A, B, E, P = O.dp_tableau()
    assert np.array_equal(C, RK45.C)
    assert np.array_equal(A, RK45.A[:, :5])
    assert np.array_equal(B, RK45.B)
    assert np.array_equal(E, RK45.E)
    assert np.array_equal(P, RK45.P)


@pytest.mark.parametrize("tag", ["nogd", "gd", "none"])
def test_rhs_vs_reference(golden_dir, g0, tag):
    """G1+G3: ode_core.py:81-166 + nn_residual.py:100-151 + hybrid_ode_nn.py:108-134."""
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    meal, tvns = (None, None) if tag == "none" else (r["meal"], r["tvns"])
    gd = r["gd"] if tag == "gd" else None
    o64 = O.rhs(r["x"], r["t"], meal, tvns, gd, g0["ode"], g0["nn"], 64, 4, np.float64)
    o32 = O.rhs(r["x"], r["t"], meal, tvns, gd, g0["ode"], g0["nn"], 64, 4, np.float32)
    assert rel(o64, r[f"rhs_f64_{tag}"]) < 1e-12
    assert rel(o32, r[f"rhs_f32_{tag}"]) < 5e-6
    # mechanistic part alone: zero the MLP
    z = np.zeros_like(g0["nn"])
    m64 = O.rhs(r["x"], r["t"], meal, tvns, gd, g0["ode"], z, 64, 4, np.float64)
    assert rel(m64, r[f"ode_f64_{tag}"]) < 1e-12
    m32 = O.rhs(r["x"], r["t"], meal, tvns, gd, g0["ode"], z, 64, 4, np.float32)
    assert rel(m32, r[f"ode_f32_{tag}"]) < 5e-6


def test_nn_residual_vs_reference(golden_dir, g0, g0_small):
    """G2: NNResidual.forward alone = rhs - mechanistic."""
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    z = np.zeros_like(g0["nn"])
    args = (r["x"], r["t"], r["meal"], r["tvns"], None, g0["ode"])
    nn64 = O.rhs(*args, g0["nn"], 64, 4, np.float64) - O.rhs(*args, z, 64, 4, np.float64)
    assert np.max(np.abs(nn64 - r["nn_f64"])) < 1e-12
    # (32,2) network: rhs on the same inputs
    o = O.rhs(r["x"], r["t"], r["meal"], r["tvns"], None, g0_small["ode"], g0_small["nn"], 32, 2, np.float32)
    assert rel(o, r["rhs_f32_h32l2"]) < 5e-6
    # the (6,)/0-dim path used by ode_func (hybrid_ode_nn.py:206-237) gives the batched values
    o8 = O.rhs(r["x"][:8], r["t"][:8], r["meal"][:8], r["tvns"][:8], None, g0["ode"], g0["nn"], 64, 4, np.float32)
    assert rel(o8, r["rhs_f32_single8"]) < 5e-6


def test_rhs_vjp_vs_reference_autograd(golden_dir, g0):
    """K5 oracle: VJP of ode_residual vs torch autograd of the reference (hybrid_ode_nn.py:318-330 path)."""
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    gx, gnn, _ = O.rhs_vjp(r["x"], r["t"], r["meal"], r["tvns"], None, g0["ode"], g0["nn"], 64, 4,
                           r["vjp_w"], np.float64)
    assert rel(gx, r["vjp_gx"]) < 2e-6          # golden is fp32 autograd
    assert np.max(np.abs(gnn - r["vjp_gnn"])) / np.max(np.abs(r["vjp_gnn"])) < 2e-6


def test_rhs_vjp_gode_finite_difference(golden_dir, g0):
    """d/d(ode constants): analytic partials vs central differences of the oracle RHS (fp64)."""
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    sel = slice(0, 40)     # physiological states, away from the Hill / Michaelis poles
    x, t, meal, tvns, gd = r["x"][sel], r["t"][sel], r["meal"][sel], r["tvns"][sel], r["gd"][sel] + 10.0
    w = r["vjp_w"][sel].astype(np.float64)
    ode = g0["ode"].astype(np.float64)
    _, _, gode = O.rhs_vjp(x, t, meal, tvns, gd, ode, g0["nn"], 64, 4, w, np.float64)
    for i in range(17):
        e = 1e-6 * max(1.0, abs(ode[i]))
        p, m = ode.copy(), ode.copy()
        p[i] += e
        m[i] -= e
        fp = O.rhs(x, t, meal, tvns, gd, p, g0["nn"], 64, 4, np.float64)
        fm = O.rhs(x, t, meal, tvns, gd, m, g0["nn"], 64, 4, np.float64)
        fd = float(((fp - fm) * w).sum() / (2 * e))
        assert abs(fd - gode[i]) <= 1e-6 * max(1.0, abs(fd)), (i, fd, gode[i])


CASES = ["t61_zero", "t61_pulses", "t241_pulses", "t241_zero", "t61_const", "t61_rand", "4gi_csv"]


@pytest.mark.parametrize("name", CASES)
def test_grid_integrator_vs_converged_reference(golden_dir, g0, name):
    """G4: the grid-broken DP5(4) (what the HIP kernel implements) against (a) the reference's
    forward(solver='rk45', rtol=1e-10, atol=1e-12) and (b) an fp64 converged solve of the
    reference's own `.double().ode_residual`."""
    g = np.load(os.path.join(golden_dir, f"g4_{name}.npz"))
    args = (g["x0"], g["t"], g["meal"], g["tvns"], None, g0["ode"], g0["nn"], 64, 4)
    conv = g["y_f64_converged_first4"]
    s64 = O.solve(*args, rtol=1e-10, atol=1e-12, dtype=np.float64)
    assert s64.status.max() == 0
    assert rel(s64.y[:4], conv) < 1e-6
    assert rel(s64.y, g["y_rk45_tight"]) < 5e-5            # golden: fp32 RHS + fp32-rounded output
    # default tolerance (rtol 1e-6): already converged to 1e-6 because steps break at the kinks
    s64d = O.solve(*args, rtol=1e-6, atol=1e-8, dtype=np.float64)
    assert rel(s64d.y[:4], conv) < 1e-6
    s32 = O.solve(*args, rtol=1e-6, atol=1e-8, dtype=np.float32)
    assert s32.status.max() == 0
    assert rel(s32.y[:4], conv) < 1e-4
    assert s32.nsteps.min() >= g["t"].shape[-1] - 1          # >= 1 step per grid interval


def test_grid_integrator_batched_time_small_net(golden_dir, g0_small):
    g = np.load(os.path.join(golden_dir, "g4_batched_t_h32l2.npz"))
    s = O.solve(g["x0"], g["t"], g["meal"], g["tvns"], None, g0_small["ode"], g0_small["nn"], 32, 2,
                rtol=1e-10, atol=1e-12, dtype=np.float64)
    assert rel(s.y, g["y_f64_converged_first4"]) < 1e-6
    assert rel(s.y, g["y_rk45_tight"]) < 5e-6


@pytest.mark.parametrize("name", ["t61_zero", "t241_zero", "t61_const"])
def test_reference_mode_default_tolerance(golden_dir, g0, name):
    """SciPy-RK45 restatement (whole-span steps, dense output, fp32 RHS inside fp64 stepping) vs
    the reference at its DEFAULT tolerances.  Only meaningful without meal kinks: with kinks the
    reference's own answer moves by 1e-2 with the last bit of the RHS."""
    g = np.load(os.path.join(golden_dir, f"g4_{name}.npz"))
    y, st, ns, nf = O.solve_reference_mode(g["x0"], g["t"], g["meal"], g["tvns"], None, g0["ode"], g0["nn"],
                                           64, 4, 1e-6, 1e-8)
    assert st.max() == 0
    assert rel(y, g["y_rk45_default"]) < 2e-5


@pytest.mark.parametrize("name", CASES)
def test_reference_mode_tight_tolerance(golden_dir, g0, name):
    g = np.load(os.path.join(golden_dir, f"g4_{name}.npz"))
    y, st, _, _ = O.solve_reference_mode(g["x0"], g["t"], g["meal"], g["tvns"], None, g0["ode"], g0["nn"],
                                         64, 4, 1e-10, 1e-12)
    assert st.max() == 0
    assert rel(y, g["y_rk45_tight"]) < 5e-5    # fp32-RHS rounding noise (z-scored 4GI states: 2e-5)


def test_rk4_fixed_step_config1(golden_dir, g0):
    """BASELINE config 1: fixed-step RK4, pure ODECore (MLP zeroed), fp64, 32 patients.  One RK4
    step per 5-min interval is 4th order: agrees with the converged DP5(4) to ~1e-6."""
    rng = np.random.default_rng(0)
    B, T = 32, 241
    x0 = np.array([5, 60, 80, 10, 0, 1.0]) * (1 + 0.05 * rng.standard_normal((B, 6)))
    t = np.arange(T) * (5.0 / 60.0)
    meal = np.zeros((B, T))
    for b in range(B):
        meal[b, rng.choice(np.arange(6, 235), 4, replace=False)] = 1.0
    z = np.zeros_like(g0["nn"])
    rk4 = O.solve(x0, t, meal, None, None, g0["ode"], z, 64, 4, method=O.METHOD_RK4, dtype=np.float64)
    dp = O.solve(x0, t, meal, None, None, g0["ode"], z, 64, 4, rtol=1e-11, atol=1e-13, dtype=np.float64)
    assert rk4.status.max() == 0 and (rk4.nsteps == T - 1).all() and (rk4.nfev == 4 * (T - 1)).all()
    assert rel(rk4.y, dp.y) < 1e-5


def _loss_and_grad(g0, x0, t, meal, tvns, c, dtype, rtol, atol, nn=None, method=O.METHOD_DP54):
    nn = g0["nn"] if nn is None else nn
    s = O.solve(x0, t, meal, tvns, None, g0["ode"], nn, g0["H"], g0["L"], method=method, rtol=rtol, atol=atol,
                dtype=dtype, want_tape=True)
    L = float((s.y.astype(np.float64) * c).sum())
    gx0, gnn, gode = O.solve_bwd(s, c)
    return L, gx0, gnn, gode, s


def test_adjoint_vs_finite_differences_of_oracle(golden_dir, g0):
    """Discrete adjoint (fp64) vs central differences of the oracle forward, fixed RK4 steps so the
    step sequence cannot move, plus DP5(4) at tight tolerance."""
    g = np.load(os.path.join(golden_dir, "g7_fd_reference.npz"))
    x0, t, meal, tvns, c = (g[k].astype(np.float64) for k in ("x0", "t", "meal", "tvns", "c"))
    for method, rtol, atol, tol in ((O.METHOD_RK4, 0, 0, 2e-6), (O.METHOD_DP54, 1e-11, 1e-13, 2e-5)):
        L, gx0, gnn, gode, s = _loss_and_grad(g0, x0, t, meal, tvns, c, np.float64, rtol, atol, method=method)
        nn = g0["nn"].astype(np.float64)
        for idx in list(g["fd_param_index"]) + [13509, 13504]:
            e = 1e-5
            p, m = nn.copy(), nn.copy()
            p[idx] += e
            m[idx] -= e
            Lp = _loss_and_grad(g0, x0, t, meal, tvns, c, np.float64, rtol, atol, nn=p, method=method)[0]
            Lm = _loss_and_grad(g0, x0, t, meal, tvns, c, np.float64, rtol, atol, nn=m, method=method)[0]
            fd = (Lp - Lm) / (2 * e)
            assert abs(fd - gnn[idx]) <= tol * max(1.0, abs(fd)), (method, idx, fd, gnn[idx])
        for b in range(x0.shape[0]):
            for i in range(6):
                e = 1e-6 * max(1.0, abs(x0[b, i]))
                xp, xm = x0.copy(), x0.copy()
                xp[b, i] += e
                xm[b, i] -= e
                Lp = _loss_and_grad(g0, xp, t, meal, tvns, c, np.float64, rtol, atol, method=method)[0]
                Lm = _loss_and_grad(g0, xm, t, meal, tvns, c, np.float64, rtol, atol, method=method)[0]
                fd = (Lp - Lm) / (2 * e)
                assert abs(fd - gx0[b, i]) <= tol * max(1.0, abs(fd)), (method, b, i, fd, gx0[b, i])
        ode = g0["ode"].astype(np.float64)
        for i in (0, 1, 5, 8, 10, 11, 14):
            e = 1e-6 * max(1e-2, abs(ode[i]))
            gp = dict(g0, ode=ode.copy())
            gm = dict(g0, ode=ode.copy())
            gp["ode"][i] += e
            gm["ode"][i] -= e
            Lp = _loss_and_grad(gp, x0, t, meal, tvns, c, np.float64, rtol, atol, method=method)[0]
            Lm = _loss_and_grad(gm, x0, t, meal, tvns, c, np.float64, rtol, atol, method=method)[0]
            fd = (Lp - Lm) / (2 * e)
            assert abs(fd - gode[i]) <= 10 * tol * max(1.0, abs(fd)), (method, i, fd, gode[i])


def test_adjoint_vs_finite_differences_of_reference(golden_dir, g0):
    """G7: central differences of the REFERENCE forward (rk45 @ 1e-10/1e-12, fp32
    outputs) for 11 weights and all x0 entries.  fp32 output quantisation (6e-8*|y| per output, eps>=1e-3)
    puts ~0.03 of absolute noise on each FD value, so the bar is 5e-3 of the gradient scale (8.07)."""
    g = np.load(os.path.join(golden_dir, "g7_fd_reference.npz"))
    _, gx0, gnn, _, _ = _loss_and_grad(g0, g["x0"], g["t"], g["meal"], g["tvns"], g["c"].astype(np.float64),
                                       np.float64, 1e-10, 1e-12)
    scale_p = np.max(np.abs(g["fd_param_grad"]))
    assert np.max(np.abs(gnn[g["fd_param_index"]] - g["fd_param_grad"])) < 5e-3 * scale_p
    scale_x = np.max(np.abs(g["fd_x0_grad"]))
    assert np.max(np.abs(gx0 - g["fd_x0_grad"])) < 5e-3 * scale_x


def test_failed_trajectory_reports_status_not_exception(g0):
    """hybrid_ode_nn.py:243-256: never raise; rows after the failure stay zero."""
    x0 = np.array([[5, 60, 80, 10, 0, 1.0]])
    t = np.linspace(0, 20, 241)
    s = O.solve(x0, t, None, None, None, g0["ode"], g0["nn"], 64, 4, max_steps=10, dtype=np.float32)
    assert s.status[0] == 1 and s.nsteps[0] == 10
    assert np.all(s.y[0, 11:] == 0) and np.all(s.y[0, :10, 0] != 0)


def test_adjoint_rows_that_copy_x0_and_rows_of_a_failed_trajectory(g0):
    """Edge cases of the cotangent injection, against central differences of the oracle forward (RK4: the step sequence
    cannot move).  (i) the grid STARTS with repeated times: rows 1..k are copies of x0 and their cotangents belong to
    gx0; (ii) every interval has zero length (no step at all); (iii) a trajectory that runs out of steps right after a
    step that closed its interval: that interval's end row (and the zero-length copies behind it) were still written."""
    rng = np.random.default_rng(5)
    x0 = np.array([[5, 60, 80, 10, 0, 1.0], [6, 50, 70, 12, 0.1, 0.9]])
    nn, ode = g0["nn"].astype(np.float64), g0["ode"].astype(np.float64)

    def run(x, t, c, max_steps=None):
        s = O.solve(x, t, meal, None, None, ode, nn, 64, 4, method=O.METHOD_RK4, dtype=np.float64, want_tape=True,
                    max_steps=max_steps)
        return float((s.y * c).sum()), s

    for t, max_steps in ((np.array([0.0, 0.0, 0.0, 0.1, 0.2, 0.2, 0.35]), None),      # (i)
                         (np.zeros(4), None),                                         # (ii)
                         (np.array([0.0, 0.0, 0.1, 0.2, 0.2, 0.2, 0.3, 0.4]), 2)):    # (iii): budget ends after interval 2
        T = len(t)
        meal = rng.random((2, T))
        c = rng.standard_normal((2, T, 6))
        L, s = run(x0, t, c, max_steps)
        if max_steps is not None:
            assert (s.status == 1).all() and (s.nsteps == 2).all()
            assert (s.y[:, 6:] == 0).all() and (s.y[:, 5] != 0).any()      # rows 3 (closing row), 4, 5 (copies) written
        gx0, gnn, _ = O.solve_bwd(s, c)
        for b in range(2):
            for i in range(6):
                e = 1e-6 * max(1.0, abs(x0[b, i]))
                xp, xm = x0.copy(), x0.copy()
                xp[b, i] += e
                xm[b, i] -= e
                fd = (run(xp, t, c, max_steps)[0] - run(xm, t, c, max_steps)[0]) / (2 * e)
                assert abs(fd - gx0[b, i]) <= 2e-6 * max(1.0, abs(fd)), (t, b, i, fd, gx0[b, i])


# ---------------------------------------------------------------------------- activations other than ReLU
@pytest.mark.parametrize("name,code", [("tanh", O.ACT_TANH), ("elu", O.ACT_ELU), ("leaky_relu", O.ACT_LEAKY_RELU)])
def test_oracle_activations_vs_reference(golden_dir, name, code):
    """NNResidual's other activations (models/nn_residual.py:50-56; reached by replacing `model.nn_residual`, HybridODENN itself
    only builds ReLU): the oracle's RHS, its VJP and its converged trajectories against values captured from the reference
    (tools/capture_golden.py g_act: a 16 x 3 network with biases on both sides of zero)."""
    g = np.load(os.path.join(golden_dir, f"g_act_{name}.npz"))
    L = O.layers(3, code)
    f64 = O.rhs(g["x"], g["t"], g["meal"], g["tvns"], None, g["ode"], g["nn_flat"], 16, L, dtype=np.float64)
    assert rel(f64, g["rhs_f64"]) < 1e-12
    f32 = O.rhs(g["x"], g["t"], g["meal"], g["tvns"], None, g["ode"], g["nn_flat"], 16, L, dtype=np.float32)
    assert rel(f32, g["rhs_f32"].astype(np.float64)) < 2e-5
    gx, gnn, _ = O.rhs_vjp(g["x"], g["t"], g["meal"], g["tvns"], None, g["ode"], g["nn_flat"], 16, L, g["vjp_w"], dtype=np.float64)
    assert rel(gx, g["vjp_gx_f64"]) < 1e-10
    assert float(np.linalg.norm(gnn - g["vjp_gnn_f64"]) / np.linalg.norm(g["vjp_gnn_f64"])) < 1e-12
    s = O.solve(g["traj_x0"], g["traj_t"], g["traj_meal"], g["traj_tvns"], None, g["ode"], g["nn_flat"], 16, L, rtol=1e-10, atol=1e-12,
                dtype=np.float64)
    assert int(s.status.max()) == 0 and rel(s.y, g["traj_y_rk45_tight"].astype(np.float64)) < 2e-5      # (the reference's RHS is fp32)
    # the activation really is a different function
    relu = O.rhs(g["x"], g["t"], g["meal"], g["tvns"], None, g["ode"], g["nn_flat"], 16, 3, dtype=np.float64)
    assert rel(relu, g["rhs_f64"]) > 1e-3


def test_oracle_under_address_and_ub_sanitizers(golden_dir):
    """`make -C oracle asan` (gcc -fsanitize=address,undefined): the checker itself is checked -- RHS + VJP, an adaptive solve with
    a tape, the adjoint, a failing trajectory, the SciPy-mode solve and the 4GI generator run through the instrumented build in a
    child process (libasan must be the first library of the process) and must come back clean and with the golden values."""
    import shutil
    import subprocess
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(here, "oracle"), "-s", "asan"], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not found")
    code = f"""
import os, sys, numpy as np
sys.path.insert(0, {here!r})
from oracle import oracle as O
from oracle import fourgi as F
g = np.load(os.path.join({golden_dir!r}, "g4_t61_pulses.npz")); w = np.load(os.path.join({golden_dir!r}, "g0_weights_h64_l4.npz"))
for dt in (np.float32, np.float64):
    f = O.rhs(g["x0"], np.zeros(len(g["x0"])), g["meal"][:, 0], g["tvns"][:, 0], None, w["ode"], w["nn_flat"], 64, 4, dtype=dt)
    O.rhs_vjp(g["x0"], np.zeros(len(g["x0"])), g["meal"][:, 0], g["tvns"][:, 0], None, w["ode"], w["nn_flat"], 64, 4, np.ones_like(f), dtype=dt)
    s = O.solve(g["x0"], g["t"], g["meal"], g["tvns"], None, w["ode"], w["nn_flat"], 64, 4, dtype=dt, want_tape=True, max_steps=200)
    assert int(s.status.max()) == 0
    gx, gn, go = O.solve_bwd(s, np.ones_like(s.y))
    assert np.isfinite(gn).all() and np.isfinite(go).all()
ref = O.solve(g["x0"], g["t"], g["meal"], g["tvns"], None, w["ode"], w["nn_flat"], 64, 4, rtol=1e-10, atol=1e-12, dtype=np.float64)
assert np.max(np.abs(ref.y - g["y_rk45_tight"]) / (np.abs(g["y_rk45_tight"]) + 1e-3)) < 2e-5
bad = g["x0"].copy(); bad[1, 3] = -50.0
with np.errstate(all="ignore"):
    sb = O.solve(bad, g["t"], g["meal"], g["tvns"], None, w["ode"], w["nn_flat"], 64, 4, dtype=np.float32, want_tape=True, max_steps=8)
    O.solve_bwd(sb, np.ones_like(sb.y))
assert sb.status[1] != 0
O.solve_reference_mode(g["x0"][:2], g["t"], g["meal"][:2], g["tvns"][:2], None, w["ode"], w["nn_flat"], 64, 4)
F.simulate(F.BASELINE[None] * np.ones((3, 1)), 13, 5.0, [0.2], [60.0])
print("sanitized oracle ok")
"""
    env = dict(os.environ, LD_PRELOAD=asan, HODE_ORACLE_LIB=os.path.join(here, "oracle", "_build", "libhode_oracle_asan.so"),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
