"""The generic network path (csrc/hode_generic.hip): MLP shapes outside the register-resident envelope, up to 8 x 128 --
in particular the reference's largest config, nn_hidden 128 / nn_layers 5 (configs/ablation_no_physics.yaml:11-12) --
against the oracle (HODE_MAXH 128, HODE_MAXL 8): K1, K5, K2+K3 and K4, fp32 and fp64, through the C ABI and the class."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle as O  # noqa: E402  (checker only)

SHAPES = [(128, 5), (96, 6), (64, 5), (100, 2), (128, 8), (7, 5), (100, 1),        # (100, 1): no hidden matrix, wider than a wave
          # widths that are not multiples of 16 INSIDE the register-resident envelope (H <= 64, L <= 4): the bounded weight
          # gathers of the row-block order (mlp_hidden_blk) and of the rotating output layer (out_rot_fill)
          (40, 3), (7, 2), (33, 4)]


def rel(a, b, floor=1e-3):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / (np.abs(b) + floor)))


def relnorm(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))


def dev(a, dtype):
    return None if a is None else torch.as_tensor(np.asarray(a), dtype=dtype, device="cuda")


@pytest.fixture(scope="module")
def hode():
    import hode as h
    h.load()
    return h


def net(H, L, seed=0):
    """xavier-like hidden layers (gain 0.5 so that ReLU units live and die), small non-zero output layer."""
    rng = np.random.default_rng(seed + 1000 * H + L)
    parts = []
    dims = [(H, 9)] + [(H, H)] * (L - 1) + [(6, H)]
    for i, (o, n) in enumerate(dims):
        std = 0.02 if i == len(dims) - 1 else 0.5 * (2.0 / (o + n)) ** 0.5
        parts += [rng.standard_normal((o, n)) * std, rng.standard_normal(o) * (0.01 if i == len(dims) - 1 else 0.05)]
    p = np.concatenate([a.reshape(-1) for a in parts]).astype(np.float32).astype(np.float64)
    assert p.size == O.n_params(H, L)
    return p


def cohort(golden_dir, B=6):
    g = np.load(os.path.join(golden_dir, "g4_t61_pulses.npz"))
    sel = np.arange(0, 61, 4)
    return g["x0"][:B].astype(np.float64), g["t"][sel].astype(np.float64), g["meal"][:B, sel].astype(np.float64), g["tvns"][:B, sel].astype(np.float64)


@pytest.mark.parametrize("H,L", SHAPES)
def test_rhs_fwd_bwd_generic_vs_oracle(hode, golden_dir, g0, H, L):
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    x, t, meal, tv, gd = (r[k].astype(np.float64) for k in ("x", "t", "meal", "tvns", "gd"))
    nn, ode = net(H, L), g0["ode"].astype(np.float64)
    w = np.random.default_rng(1).standard_normal(x.shape)
    for dt, tol_f, tol_g in ((torch.float64, 1e-12, 1e-10), (torch.float32, 5e-5, 1e-4)):      # fp32: states up to 500, gain-0.5 layers
        npdt = np.float64 if dt == torch.float64 else np.float32
        want = O.rhs(x, t, meal, tv, gd, ode, nn, H, L, dtype=np.float64)
        got = hode.rhs_fwd(dev(x, dt), dev(t, dt), dev(meal, dt), dev(tv, dt), dev(gd, dt), dev(ode, dt), dev(nn, dt), H, L)
        assert rel(got.cpu().numpy(), want) < tol_f, (H, L, dt)
        rgx, rgnn, rgode = O.rhs_vjp(x, t, meal, tv, gd, ode, nn, H, L, w, dtype=np.float64)
        e = 1e-6                                                # d/dt by central differences of the oracle (t only enters the MLP)
        rgt = ((O.rhs(x, t + e, meal, tv, gd, ode, nn, H, L, dtype=np.float64) - O.rhs(x, t - e, meal, tv, gd, ode, nn, H, L, dtype=np.float64))
               / (2 * e) * w).sum(1)
        gx, gt, gnn, gode = hode.rhs_bwd(dev(x, dt), dev(t, dt), dev(meal, dt), dev(tv, dt), dev(gd, dt), dev(ode, dt), dev(nn, dt), H, L,
                                         dev(w, dt), want_gt=True, want_gnn=True, want_gode=True)
        assert relnorm(gx.cpu().numpy(), rgx) < tol_g and relnorm(gnn.cpu().numpy(), rgnn) < tol_g, (H, L, npdt)
        assert relnorm(gt.cpu().numpy(), rgt) < 1e-4 and relnorm(gode.cpu().numpy(), rgode) < 10 * tol_g


@pytest.mark.parametrize("H,L", SHAPES)
def test_solve_and_adjoint_generic_fp64_vs_oracle(hode, golden_dir, g0, H, L):
    x0, t, meal, tv = cohort(golden_dir)
    nn, ode = net(H, L), g0["ode"].astype(np.float64)
    c = np.random.default_rng(2).standard_normal((x0.shape[0], len(t), 6))
    dt = torch.float64
    for method in (O.METHOD_DP54, O.METHOD_RK4):
        ms = 1500 if method == O.METHOD_DP54 else None         # same step budget on both sides (the lively test networks take 20-30 steps per interval)
        ref = O.solve(x0, t, meal, tv, None, ode, nn, H, L, method=method, rtol=1e-8, atol=1e-10, dtype=np.float64, want_tape=True,
                      max_steps=ms)
        s = hode.solve_fwd(dev(x0, dt), dev(t, dt), dev(meal, dt), dev(tv, dt), None, dev(ode, dt), dev(nn, dt), H, L, method=method,
                           rtol=1e-8, atol=1e-10, want_tape=True, max_steps=ms)
        assert int(s.status.max()) == 0 and np.array_equal(s.nsteps.cpu().numpy(), ref.nsteps), (H, L, method)
        # same algorithm, same dtype, different summation order inside a layer: the adaptive controller (hundreds of steps at
        # 1e-8 for these lively networks) turns last-bit differences into tolerance-sized ones; RK4 has no controller
        ty, tg = (2e-7, 5e-5) if method == O.METHOD_DP54 else (1e-11, 1e-9)     # (a ReLU unit near zero may switch between the two runs)
        assert np.array_equal(s.nfev.cpu().numpy(), ref.nfev) and rel(s.y.cpu().numpy(), ref.y) < ty
        rx, rnn, rode = O.solve_bwd(ref, c)
        gx0, gnn, gode = hode.solve_bwd(s, dev(c, dt), want_gode=True)
        assert relnorm(gx0.cpu().numpy(), rx) < tg and relnorm(gnn.cpu().numpy(), rnn) < tg, (H, L, method)
        assert relnorm(gode.cpu().numpy(), rode) < 10 * tg
        # without a tape: same trajectories
        s2 = hode.solve_fwd(dev(x0, dt), dev(t, dt), dev(meal, dt), dev(tv, dt), None, dev(ode, dt), dev(nn, dt), H, L, method=method,
                            rtol=1e-8, atol=1e-10, max_steps=ms)
        assert torch.equal(s2.y, s.y)


def test_reference_ablation_shape_fp32_forward_adjoint_and_sets(hode, golden_dir, g0):
    """nn_hidden 128, nn_layers 5 at the reference's batch (32 windows of 61 points): fp32 at the default tolerances against
    the fp64 oracle at tight ones (bars 1e-3 / 1e-4), two parameter sets, batched time grid, failure -> status."""
    H, L = 128, 5
    g = np.load(os.path.join(golden_dir, "g4_t61_pulses.npz"))
    x0 = np.concatenate([g["x0"]] * 4) * (1 + 0.01 * np.arange(32)[:, None])
    meal, tv = np.concatenate([g["meal"]] * 4), np.concatenate([g["tvns"]] * 4)
    t = g["t"].astype(np.float64)
    nn, ode = net(H, L), g0["ode"].astype(np.float64)
    f32 = torch.float32
    s = hode.solve_fwd(dev(x0, f32), dev(t, f32), dev(meal, f32), dev(tv, f32), None, dev(ode, f32), dev(nn, f32), H, L, want_tape=True,
                       max_steps=1500)
    ref = O.solve(x0, t, meal, tv, None, ode, nn, H, L, rtol=1e-10, atol=1e-12, dtype=np.float64, want_tape=True)
    assert int(s.status.max()) == 0 and rel(s.y.cpu().numpy(), ref.y) < 1e-4
    bare = O.solve(x0[:4], t, meal[:4], tv[:4], None, ode, np.zeros_like(nn), H, L, rtol=1e-8, atol=1e-10, dtype=np.float64)
    assert rel(bare.y, ref.y[:4]) > 1e-2                                  # the network matters in this test
    c = np.random.default_rng(3).standard_normal(ref.y.shape)
    _, rnn, _ = O.solve_bwd(ref, c)
    gx0, gnn, _ = hode.solve_bwd(s, dev(c, f32))
    # fp32 steps at 1e-6 through a 5 x 128 network with gain-0.5 layers (activations of O(10), units switching) against fp64 at
    # 1e-10: 1.1e-3 measured; the 1e-4 bar of north_star is the 4 x 64 reference network's (tests/test_hip_parity.py)
    assert relnorm(gnn.cpu().numpy(), rnn) < 5e-3
    # a realistically initialised network of the same shape (xavier gain 0.1, as NNResidual initialises) meets it
    rng = np.random.default_rng(9)
    parts = []
    for i, (o_, n_) in enumerate([(H, 9)] + [(H, H)] * (L - 1) + [(6, H)]):
        std = 0.01 if i == L else 0.1 * (2.0 / (o_ + n_)) ** 0.5
        parts += [rng.standard_normal((o_, n_)) * std, np.zeros(o_) if i < L else rng.standard_normal(o_) * 0.01]
    nn_r = np.concatenate([a.reshape(-1) for a in parts]).astype(np.float32).astype(np.float64)
    s_r = hode.solve_fwd(dev(x0, f32), dev(t, f32), dev(meal, f32), dev(tv, f32), None, dev(ode, f32), dev(nn_r, f32), H, L, want_tape=True)
    ref_r = O.solve(x0, t, meal, tv, None, ode, nn_r, H, L, rtol=1e-10, atol=1e-12, dtype=np.float64, want_tape=True)
    assert int(s_r.status.max()) == 0 and rel(s_r.y.cpu().numpy(), ref_r.y) < 1e-4
    _, rnn_r, _ = O.solve_bwd(ref_r, c)
    _, gnn_r, _ = hode.solve_bwd(s_r, dev(c, f32))
    assert relnorm(gnn_r.cpu().numpy(), rnn_r) < 1e-4
    # two parameter sets x 16 patients, batched time grid
    nn2 = np.concatenate([nn, 0.5 * nn])
    tb = np.tile(t, (32, 1)) * (1 + 0.01 * np.arange(32)[:, None])
    dt = torch.float64
    s2 = hode.solve_fwd(dev(x0, dt), dev(tb, dt), dev(meal, dt), dev(tv, dt), None, dev(np.concatenate([ode, ode]), dt), dev(nn2, dt), H, L,
                        n_sets=2, rtol=1e-8, atol=1e-10, want_tape=True, max_steps=1500)
    gx2, gnn2, _ = hode.solve_bwd(s2, dev(c, dt))
    P = nn.size
    for k, (p_k, sl) in enumerate(((nn, slice(0, 16)), (0.5 * nn, slice(16, 32)))):
        r_k = O.solve(x0[sl], tb[sl], meal[sl], tv[sl], None, ode, p_k, H, L, rtol=1e-8, atol=1e-10, dtype=np.float64, want_tape=True,
                      max_steps=1500)
        assert rel(s2.y[sl].cpu().numpy(), r_k.y) < 2e-7
        rx_k, rnn_k, _ = O.solve_bwd(r_k, c[sl])
        assert relnorm(gnn2[k * P:(k + 1) * P].cpu().numpy(), rnn_k) < 5e-5 and relnorm(gx2[sl].cpu().numpy(), rx_k) < 5e-5
    # step budget exhausted -> status 1, zero rows, finite gradients of what was written
    s3 = hode.solve_fwd(dev(x0[:3], dt), dev(t, dt), dev(meal[:3], dt), dev(tv[:3], dt), None, dev(ode, dt), dev(nn, dt), H, L,
                        rtol=1e-8, atol=1e-10, want_tape=True, max_steps=20)
    r3 = O.solve(x0[:3], t, meal[:3], tv[:3], None, ode, nn, H, L, rtol=1e-8, atol=1e-10, dtype=np.float64, want_tape=True, max_steps=20)
    assert (s3.status.cpu().numpy() == 1).all() and rel(s3.y.cpu().numpy(), r3.y) < 2e-7
    g3x, g3n, _ = hode.solve_bwd(s3, dev(c[:3], dt))
    r3x, r3n, _ = O.solve_bwd(r3, c[:3])
    assert relnorm(g3x.cpu().numpy(), r3x) < 5e-5 and relnorm(g3n.cpu().numpy(), r3n) < 5e-5


@pytest.mark.parametrize("H,L", [(128, 5), (96, 3), (100, 3), (64, 5), (40, 3), (7, 3), (100, 1)])
def test_team_kernels_of_the_forward_agree_with_each_other_and_the_oracle(hode, golden_dir, g0, H, L):
    """The forward solve of a generic network picks one of three team kernels by batch size (hode_generic.hip launch_fwd_generic_m):
    eight waves with register-resident rows (small batches), eight or four waves streaming the rows from L2 (larger ones; 16-byte
    loads when the width is a multiple of the chunk and the parameter set is 16-byte aligned, element loads otherwise).  All of them
    evaluate the same sums in the same order, so a trajectory must come out BITWISE the same whichever kernel integrates it -- widths
    that fill the chunks (128, 64), leave whole chunks empty (96, 40) or cut a chunk (100, 7); three parameter sets in one launch (the
    second set's parameters start 8 bytes off a 16-byte boundary: the element-load path) -- and match the fp64 oracle."""
    f32 = torch.float32
    g = np.load(os.path.join(golden_dir, "g4_t61_pulses.npz"))
    sel = np.arange(0, 61, 5)
    t = g["t"][sel].astype(np.float64)
    nn, ode = net(H, L, seed=3), g0["ode"].astype(np.float64)
    B = 600                                                   # > 512: four waves, streamed
    rng = np.random.default_rng(H)
    x0 = g["x0"][rng.integers(0, 8, B)] * (1 + 0.02 * rng.standard_normal((B, 6)))
    meal, tv = g["meal"][rng.integers(0, 8, B)][:, sel], np.zeros((B, sel.size))
    big = hode.solve_fwd(dev(x0, f32), dev(t, f32), dev(meal, f32), dev(tv, f32), None, dev(ode, f32), dev(nn, f32), H, L, want_tape=True)
    assert int(big.status.max()) == 0
    for n_small in (5, 300):                                  # resident rows (L - 1 <= 4) / eight waves streamed or resident
        sub = slice(17, 17 + n_small)
        small = hode.solve_fwd(dev(x0[sub], f32), dev(t, f32), dev(meal[sub], f32), dev(tv[sub], f32), None, dev(ode, f32), dev(nn, f32), H, L)
        assert torch.equal(small.y, big.y[sub]) and torch.equal(small.nsteps, big.nsteps[sub]), (H, L, n_small)
    ref = O.solve(x0[:6], t, meal[:6], tv[:6], None, ode, nn, H, L, rtol=1e-10, atol=1e-12, dtype=np.float64)
    assert rel(big.y[:6].cpu().numpy(), ref.y) < 1e-3        # north_star's fp32 bar (gain-0.5 layers: activations of O(10), 2e-4 measured)
    # three parameter sets x 200 patients in one launch against the sets on their own (and the stage tape they leave: the adjoint)
    nn3 = np.concatenate([nn, 0.7 * nn, 1.2 * nn])
    ode3 = np.concatenate([ode, ode, ode])
    s3 = hode.solve_fwd(dev(x0, f32), dev(t, f32), dev(meal, f32), dev(tv, f32), None, dev(ode3, f32), dev(nn3, f32), H, L, n_sets=3, want_tape=True)
    c = rng.standard_normal((B, sel.size, 6)) / B
    _, gnn3, _ = hode.solve_bwd(s3, dev(c, f32))
    P = nn.size
    for k, f in enumerate((1.0, 0.7, 1.2)):
        sl = slice(200 * k, 200 * (k + 1))
        sk = hode.solve_fwd(dev(x0[sl], f32), dev(t, f32), dev(meal[sl], f32), dev(tv[sl], f32), None, dev(ode, f32), dev(f * nn, f32), H, L, want_tape=True)
        assert torch.equal(sk.y, s3.y[sl]), (H, L, k)
        _, gk, _ = hode.solve_bwd(sk, dev(c[sl], f32))
        assert relnorm(gnn3[k * P:(k + 1) * P].cpu().numpy(), gk.cpu().numpy()) < 2e-5


def test_class_surface_with_the_ablation_network():
    """HybridODENN(nn_hidden=128, nn_layers=5) -- what train_hybrid.py builds from configs/ablation_no_physics.yaml -- runs
    forward / loss / backward / an Adam step on the HIP path, and its ode_residual equals the torch modules."""
    import models
    torch.manual_seed(0)
    m = models.HybridODENN(nn_hidden=128, nn_layers=5, device="cuda")
    with torch.no_grad():
        m.nn_residual.network[-1].weight.normal_(0, 0.01)
    assert sum(p.numel() for p in m.nn_residual.parameters()) == 68102
    B, T = 32, 61
    g = torch.Generator().manual_seed(1)
    x0 = (torch.tensor([5.0, 60.0, 80.0, 10.0, 0.0, 1.0]) * (1 + 0.05 * torch.randn(B, 6, generator=g))).cuda()
    t = torch.linspace(0, 5, T).cuda()
    meal = torch.zeros(B, T)
    meal[:, 6] = meal[:, 30] = 1.0
    u = {"meal": meal.cuda(), "tVNS": torch.zeros(B, T).cuda()}
    with torch.no_grad():
        y = m.forward(x0, t, u)
        f = m.ode_residual(t[3], x0, {"meal": u["meal"][:, 3], "tVNS": u["tVNS"][:, 3]})
        eager = m.ode_core(t[3], x0, {"meal": u["meal"][:, 3]}) + m.nn_residual(t[3], x0, x0[:, 3], u["tVNS"][:, 3])
    assert y.shape == (B, T, 6) and m.solve_failures() == 0
    assert rel(f.cpu().numpy(), eager.double().cpu().numpy()) < 5e-6
    batch = {"initial_state": x0, "observations": y + 0.05 * torch.randn(y.shape, device="cuda"), "time_points": t, "external_inputs": u}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    before = [p.detach().clone() for p in m.nn_residual.parameters()]
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = m.loss(batch, lambda1=0.0, lambda2=0.1, use_physics_loss=False)      # the ablation's settings
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0)
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.nn_residual.parameters())
    assert all(float((a - b).abs().max()) > 0 for a, b in zip(before, m.nn_residual.parameters()))
    # the physics-loss route (K1 + K5 under autograd) works for this shape too
    opt.zero_grad()
    m.loss(batch, lambda1=1.0, lambda2=0.1).backward()
    assert float(m.nn_residual.network[0].weight.grad.abs().max()) > 0


# ------------------------------------------------------------------------------------------------ activations other than ReLU
ACTS = [("tanh", 1), ("elu", 2), ("leaky_relu", 3)]


@pytest.mark.parametrize("name,code", ACTS)
def test_activations_rhs_solve_adjoint_vs_reference_and_oracle(hode, golden_dir, name, code):
    """NNResidual's other activations (reference models/nn_residual.py:50-56): K1 / K5 against values captured from the reference
    itself (G-act: forward fp32 / fp64, autograd VJP), K2 against its converged trajectories, K4 against the oracle's adjoint
    (which those captures pin, tests/test_oracle_golden.py).  Any shape with a non-ReLU code takes the generic kernels."""
    g = np.load(os.path.join(golden_dir, f"g_act_{name}.npz"))
    H, L = 16, hode.capi.layers(3, code)
    assert hode.capi.layers(3, code) == 3 + 256 * code and hode.n_params(16, L) == g["nn_flat"].size
    for dt, tol in ((torch.float64, 1e-11), (torch.float32, 2e-5)):
        out = hode.rhs_fwd(dev(g["x"], dt), dev(g["t"], dt), dev(g["meal"], dt), dev(g["tvns"], dt), None, dev(g["ode"], dt), dev(g["nn_flat"], dt), H, L)
        assert rel(out.cpu().numpy(), g["rhs_f64"]) < tol, (name, dt)
    d64 = torch.float64
    gx, _, gnn, _ = hode.rhs_bwd(dev(g["x"], d64), dev(g["t"], d64), dev(g["meal"], d64), dev(g["tvns"], d64), None, dev(g["ode"], d64),
                                 dev(g["nn_flat"], d64), H, L, dev(g["vjp_w"], d64), want_gnn=True)
    assert rel(gx.cpu().numpy(), g["vjp_gx_f64"]) < 1e-9 and relnorm(gnn.cpu().numpy(), g["vjp_gnn_f64"]) < 1e-10
    # trajectories: fp64 at tight tolerances and fp32 at the defaults vs the reference's rk45 at 1e-10 (its RHS is fp32: 2e-5)
    x0, t, meal, tv = g["traj_x0"], g["traj_t"], g["traj_meal"], g["traj_tvns"]
    s64 = hode.solve_fwd(dev(x0, d64), dev(t, d64), dev(meal, d64), dev(tv, d64), None, dev(g["ode"], d64), dev(g["nn_flat"], d64), H, L,
                         rtol=1e-10, atol=1e-12, want_tape=True)
    assert int(s64.status.max()) == 0 and rel(s64.y.cpu().numpy(), g["traj_y_rk45_tight"].astype(np.float64)) < 2e-5
    d32 = torch.float32
    s32 = hode.solve_fwd(dev(x0, d32), dev(t, d32), dev(meal, d32), dev(tv, d32), None, dev(g["ode"], d32), dev(g["nn_flat"], d32), H, L, want_tape=True)
    assert int(s32.status.max()) == 0 and rel(s32.y.cpu().numpy(), g["traj_y_rk45_tight"].astype(np.float64)) < 1e-4
    # adjoint vs oracle (same activation code convention: oracle.layers)
    ref = O.solve(x0, t, meal, tv, None, g["ode"], g["nn_flat"], H, O.layers(3, code), rtol=1e-10, atol=1e-12, dtype=np.float64, want_tape=True)
    c = np.random.default_rng(5).standard_normal(ref.y.shape)
    rx, rnn, rode = O.solve_bwd(ref, c)
    gx0, gnn, gode = hode.solve_bwd(s64, dev(c, d64), want_gode=True)
    assert relnorm(gx0.cpu().numpy(), rx) < 1e-7 and relnorm(gnn.cpu().numpy(), rnn) < 1e-7 and relnorm(gode.cpu().numpy(), rode) < 1e-6
    gx0, gnn, _ = hode.solve_bwd(s32, dev(c, d32))
    assert relnorm(gx0.cpu().numpy(), rx) < 1e-4 and relnorm(gnn.cpu().numpy(), rnn) < 1e-4
    # a code the library does not know is refused, not guessed
    with pytest.raises(hode.HodeError, match="EUNSUPPORTED"):
        hode.rhs_fwd(dev(g["x"], d32), dev(g["t"], d32), None, None, None, dev(g["ode"], d32), dev(g["nn_flat"], d32), H, hode.capi.layers(3, 4))


def test_model_with_a_tanh_residual_trains_through_the_class_surface(golden_dir):
    """`model.nn_residual = NNResidual(activation='tanh')` -- the only way the reference reaches a non-ReLU network -- runs forward,
    loss and backward on the device (generic kernels) and agrees with the torch module it replaces on ode_residual."""
    import models as M
    g = np.load(os.path.join(golden_dir, "g_act_tanh.npz"))
    m = M.HybridODENN(nn_hidden=16, nn_layers=3, device="cuda")
    m.nn_residual = M.NNResidual(9, 16, 6, 3, activation="tanh").cuda()
    flat, off = torch.tensor(g["nn_flat"]), 0
    with torch.no_grad():
        for p in m.nn_residual.parameters():
            p.copy_(flat[off:off + p.numel()].reshape(p.shape))
            off += p.numel()
    assert m.nn_residual.hip_supported() and m.nn_residual.hip_layers == 3 + 256
    x, t = torch.tensor(g["x"]).cuda(), torch.tensor(g["t"]).cuda()
    f = m.ode_residual(t, x, {"meal": torch.tensor(g["meal"]).cuda(), "tVNS": torch.tensor(g["tvns"]).cuda()})
    assert rel(f.detach().cpu().numpy(), g["rhs_f64"]) < 2e-5
    x0, tt = torch.tensor(g["traj_x0"]).cuda(), torch.tensor(g["traj_t"]).cuda()
    ext = {"meal": torch.tensor(g["traj_meal"]).cuda(), "tVNS": torch.tensor(g["traj_tvns"]).cuda()}
    with torch.no_grad():
        y = m.forward(x0, tt, ext)
    assert rel(y.cpu().numpy(), g["traj_y_rk45_tight"].astype(np.float64)) < 1e-4
    batch = {"initial_state": x0, "observations": y + 0.05 * torch.randn_like(y), "time_points": tt, "external_inputs": ext}
    loss = m.loss(batch, lambda1=0.5, lambda2=0.1)
    loss.backward()
    gr = torch.cat([p.grad.reshape(-1) for p in m.nn_residual.parameters()])
    assert torch.isfinite(loss) and torch.isfinite(gr).all() and float(gr.abs().max()) > 0
    with pytest.raises(NotImplementedError):
        bad = M.HybridODENN(nn_hidden=16, nn_layers=3, device="cuda")
        bad.nn_residual = M.NNResidual(9, 16, 6, 3, dropout=0.1).cuda()
        bad.forward(x0, tt, ext)


@pytest.mark.parametrize("H,L,B,n_sets", [(128, 5, 96, 1), (64, 5, 40, 1), (100, 2, 24, 3), (128, 5, 2100, 1), (64, 5, 640, 2), (96, 3, 1030, 1)])
def test_generic_adjoint_is_bit_reproducible_and_matches_the_oracle(hode, g0, H, L, B, n_sets):
    """Round 4 (VERDICT r3 missing 3): networks beyond 64 x 4 -- configs/ablation_no_physics.yaml trains 128 x 5 -- leave their
    parameter gradients as one row per workgroup + a fixed-order reduction, like the tuned path: the same bits run to run
    (the reference's CPU training is deterministic; the atomics of rounds 2-3 were not), with ODE-constant gradients, several
    parameter sets, and more trajectories than gradient rows (2 100 > 1 024: workgroups loop).  Values: fp32 vs the fp64 oracle."""
    import bench
    T = 31
    x0, t, meal, tv = (v.numpy().astype(np.float64) for v in bench.synth_cohort(B, 21))
    t, meal, tv = t[:T], meal[:, :T], tv[:, :T]
    per = B // n_sets
    nn = np.concatenate([net(H, L, seed=s) for s in range(n_sets)])
    ode = np.tile(g0["ode"].astype(np.float64), n_sets)
    f = lambda a: dev(a, torch.float32)      # noqa: E731
    sol = hode.solve_fwd(f(x0), f(t), f(meal), f(tv), None, f(ode), f(nn), H, L, n_sets=n_sets, want_tape=True)
    assert int(sol.status.max()) == 0
    gy = torch.randn(B, T, 6, device="cuda", generator=torch.Generator("cuda").manual_seed(2)) / (B * T)
    runs = [hode.solve_bwd(sol, gy, want_gode=True) for _ in range(3)]
    for r in runs[1:]:
        assert torch.equal(r[0], runs[0][0]) and torch.equal(r[1], runs[0][1]) and torch.equal(r[2], runs[0][2])
    gx0, gnn, gode = (v.cpu().numpy() for v in runs[0])
    P = O.n_params(H, L)
    nb = min(per, 6)                          # oracle on the first trajectories of every set: linearity gives their share
    for s in range(n_sets):
        sl = slice(s * per, s * per + nb)
        ref = O.solve(x0[sl], t, meal[sl], tv[sl], None, ode[17 * s:17 * s + 17], nn[P * s:P * s + P], H, L, rtol=1e-10, atol=1e-12,
                      dtype=np.float64, want_tape=True)
        c = np.zeros((nb, T, 6))
        c[:] = gy[sl].cpu().numpy()
        rx, rn, ro = O.solve_bwd(ref, c)
        assert relnorm(gx0[sl], rx) < 1e-4
        # the same cotangent restricted to these trajectories through the kernel: its gnn is what the oracle computed
        g2 = torch.zeros_like(gy)
        g2[sl] = gy[sl]
        k = hode.solve_bwd(sol, g2, want_gode=True)
        # (fp32 steps at 1e-6 through gain-0.5 layers against fp64 at 1e-10: 1.2e-3 measured for 5 x 128, the figure
        #  test_reference_ablation_shape_fp32_forward_adjoint_and_sets documents; a realistically initialised network meets 1e-4 there)
        assert relnorm(k[1].view(n_sets, P)[s].cpu().numpy(), rn) < 5e-3
        assert np.max(np.abs(k[2].view(n_sets, 17)[s].cpu().numpy() - ro)) < 5e-3 * np.max(np.abs(ro))
        others = [q for q in range(n_sets) if q != s]
        assert all(float(k[1].view(n_sets, P)[q].abs().max()) == 0.0 for q in others)


@pytest.mark.parametrize("H,L,B", [(128, 3, 1030), (40, 4, 2050), (96, 3, 402), (64, 5, 700)])      # teams of 8, 8, 2 and 4 trajectories
def test_teams_that_walk_several_tapes_equal_the_one_trajectory_teams(hode, g0, H, L, B):
    """Above 512 trajectories a team of the generic adjoint walks 4 or 8 tapes at once (solve_bwd_generic_multi_kernel): tapes of
    DIFFERENT lengths (tight tolerances: the step count follows each patient's meals; a few trajectories starved of steps), a batch
    that does not fill its last team.  Against the same trajectories through the one-trajectory teams, 400 at a time."""
    import bench
    T = 25
    x0, t, meal, tv = bench.synth_cohort(B, 33)
    x0 = (x0 * (0.6 + 0.8 * torch.rand(B, 6, generator=torch.Generator().manual_seed(1)))).cuda()
    t, meal, tv = t[:T].cuda() * 3.0, meal[:, :T].contiguous().cuda(), tv[:, :T].contiguous().cuda()
    nn, ode = torch.as_tensor(net(H, L, seed=5), dtype=torch.float32).cuda(), torch.as_tensor(g0["ode"], dtype=torch.float32).cuda()
    probe = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L, rtol=1e-7, atol=1e-9, max_steps=200)
    budget = int(np.percentile(probe.nsteps.cpu().numpy(), 80))          # a fifth of the trajectories runs out of steps: status 1, short tape
    sol = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L, rtol=1e-7, atol=1e-9, want_tape=True, max_steps=budget)
    ns = sol.nsteps.cpu().numpy()
    assert len(set(ns.tolist())) > 3 and int((sol.status != 0).sum()) > 0 and int((sol.status == 0).sum()) > B // 2, (set(ns.tolist()), sol.status.unique())
    gy = torch.randn(B, T, 6, device="cuda", generator=torch.Generator("cuda").manual_seed(4)) / (B * T)
    gx0, gnn, gode = hode.solve_bwd(sol, gy, want_gode=True)
    rx, rn, ro = [], 0.0, 0.0
    for lo in range(0, B, 250):                       # (<= 256: the one-trajectory teams)
        sl = slice(lo, min(lo + 250, B))
        part = hode.solve_fwd(x0[sl].contiguous(), t, meal[sl].contiguous(), tv[sl].contiguous(), None, ode, nn, H, L, rtol=1e-7, atol=1e-9,
                              want_tape=True, max_steps=budget)
        assert torch.equal(part.y, sol.y[sl]) and torch.equal(part.nsteps, sol.nsteps[sl])
        px, pn, po = hode.solve_bwd(part, gy[sl].contiguous(), want_gode=True)
        rx.append(px); rn = rn + pn.double(); ro = ro + po.double()
    rx = torch.cat(rx)
    assert torch.isfinite(gnn).all() and relnorm(gx0.cpu().numpy(), rx.cpu().numpy()) < 1e-6
    assert relnorm(gnn.cpu().numpy(), rn.cpu().numpy()) < 2e-5 and relnorm(gode.cpu().numpy(), ro.cpu().numpy()) < 2e-5
