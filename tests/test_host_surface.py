"""CPU tests of the host-side mirror (class surface, state_dict layout, torch evaluations of the
two RHS modules, VI bookkeeping), of the C-ABI library's exported symbols and of the
"fail loudly, never fall back" rule.  No GPU needed.

The autograd tests restate reference tests/test_ode_jacobians.py:58-206 and
tests/test_gradient_correctness.py:18-62 (same states, shapes, seeds and assertions).
"""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import hode
from models import HybridODENN, NNResidual, ODECore, VariationalParameters, bayes_loss, compute_posterior_predictive  # noqa: F401
from models.ode_core import ODE_PARAM_NAMES

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hybrid-ode-for-glp-1-and-glucose_amd")


def rel(a, b, floor=1e-3):
    return float(np.max(np.abs(np.asarray(a, np.float64) - b) / (np.abs(b) + floor)))


# ------------------------------------------------------------------ C ABI
def test_library_exports_every_declared_symbol():
    """Every function include/hode.h declares is exported by libhode.so (no compute calls here)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "hode.h")).read()
    declared = sorted(set(re.findall(r"\b(hode_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(hode.capi.SYMBOLS)
    lib = ctypes.CDLL(hode.lib_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hode.version().startswith("hode ")
    assert lib.hode_nn_param_count(64, 4) == 13510 and lib.hode_nn_param_count(32, 2) == 1574
    # entries + interval indices + the stage tape (h_1..h_4 + state per stage) + the adjoint's gradient rows (one per workgroup,
    # at most 1 024, never more than trajectories; row = P + 17 reals padded to a multiple of 64)
    row = (13510 + 17 + 63) // 64 * 64
    assert hode.load().hode_tape_bytes(4096, 300, 4, 4) == 4096 * 300 * 36 + 4096 * 300 * 6 * (4 * 64 + 8) * 4 + 1024 * row * 4
    assert hode.load().hode_tape_bytes(4096, 300, 8, 4) == 4096 * 300 * 68 + 4096 * 300 * 6 * (4 * 64 + 8) * 8 + 1024 * row * 8
    assert hode.load().hode_tape_bytes(64, 300, 4, 4) == 64 * 300 * 36 + 64 * 300 * 6 * (4 * 64 + 8) * 4 + 64 * row * 4
    # generic path: fp32 leaves its gradients as rows too since round 4 (deterministic reduction); fp64 keeps atomics: no rows
    row_g = (lib.hode_nn_param_count(128, 5) + 17 + 63) // 64 * 64
    assert hode.load().hode_tape_bytes_hl(64, 300, 4, 128, 5) == 64 * 300 * 36 + 64 * 300 * 6 * (2 * 5 * 64 + 8) * 4 + 64 * row_g * 4
    assert hode.load().hode_tape_bytes_hl(64, 300, 8, 128, 5) == 64 * 300 * 68 + 64 * 300 * 6 * (2 * 5 * 64 + 8) * 8


def test_argument_validation_without_gpu():
    """Bad arguments are rejected on the host before any launch (return codes of include/hode.h)."""
    lib = hode.load()
    z = ctypes.c_void_p(0)
    assert lib.hode_solve_fwd_f32(z, 4, 10, z, z, 0, z, 0, z, 0, z, 0, z, z, 1, 64, 4, 0, ctypes.c_double(1e-6),
                                  ctypes.c_double(1e-8), 100, z, z, z, z, z) == -1          # null pointers
    one = ctypes.c_void_p(16)
    assert lib.hode_solve_fwd_f32(z, 4, 10, one, one, 0, z, 0, z, 0, z, 0, one, one, 1, 129, 4, 0,
                                  ctypes.c_double(1e-6), ctypes.c_double(1e-8), 100, one, one, z, z, z) == -2  # H > 128
    assert lib.hode_solve_fwd_f32(z, 4, 10, one, one, 0, z, 0, z, 0, z, 0, one, one, 1, 64, 9, 0,
                                  ctypes.c_double(1e-6), ctypes.c_double(1e-8), 100, one, one, z, z, z) == -2  # L > 8
    assert lib.hode_solve_fwd_f32(z, 4, 10, one, one, 0, z, 0, z, 0, z, 0, one, one, 3, 64, 4, 0,
                                  ctypes.c_double(1e-6), ctypes.c_double(1e-8), 100, one, one, z, z, z) == -1  # B % n_sets
    assert lib.hode_solve_fwd_f32(z, 4, 10, one, one, 0, z, 2, z, 0, z, 0, one, one, 1, 64, 4, 0,
                                  ctypes.c_double(1e-6), ctypes.c_double(1e-8), 100, one, one, z, z, z) == -1  # meal mode w/o ptr


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_hot_path_fails_loudly_without_gpu():
    """No CPU fallback: solve / ode_residual / loss raise when there is no HIP device."""
    m = HybridODENN(nn_hidden=32, nn_layers=2, device="cpu")
    with pytest.raises(hode.HodeError):
        m.forward(torch.zeros(2, 6), torch.linspace(0, 1, 5))
    with pytest.raises(hode.HodeError):
        m.ode_residual(torch.zeros(2), torch.zeros(2, 6))
    with pytest.raises(hode.HodeError):
        hode.solve_fwd(torch.zeros(2, 6), torch.linspace(0, 1, 5), None, None, None, torch.zeros(17),
                       torch.zeros(13510), 64, 4)


# ------------------------------------------------------------------ class surface
def test_state_dict_layout_matches_reference(golden_dir):
    """G0: keys, order, shapes and parameter count of the reference's state_dict."""
    ref = json.load(open(os.path.join(golden_dir, "g0_state_dict_keys.json")))
    m = HybridODENN(device="cpu")
    sd = m.state_dict()
    assert list(sd.keys()) == ref["keys"]
    assert [list(v.shape) for v in sd.values()] == ref["shapes"]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params"] == 13510
    assert all(v.dtype == torch.float32 for v in sd.values())
    # reference checkpoints load (tests/test_training.py:297-341 round trip)
    w = np.load(os.path.join(golden_dir, "g0_weights_h64_l4.npz"))
    m.load_state_dict({k.replace("__", "."): torch.tensor(w[k]) for k in w.files if "__" in k})
    assert np.array_equal(m.nn_residual.flat_parameters().detach().numpy(), w["nn_flat"])
    assert np.array_equal(m.ode_core.param_vector().numpy(), w["ode"])


def test_constructor_surface():
    m = HybridODENN(nn_hidden=32, nn_layers=2, use_variational=False, device="cpu")   # str device, as the reference tests pass
    assert m.device == torch.device("cpu") and m.n_states == 6 and m.variational_params is None
    assert m.state_names == ["Glucose", "Insulin", "Glucagon", "GLP1", "GE", "FFA"]
    assert isinstance(m.nn_residual.network, torch.nn.Sequential)
    assert m.nn_residual.network[-1].weight.shape == (6, 32)
    assert float(m.nn_residual.network[-1].weight.detach().abs().sum()) == 0.0      # zero-initialised output layer
    assert (m.nn_residual.input_dim, m.nn_residual.hidden_dim, m.nn_residual.output_dim, m.nn_residual.n_layers) == (9, 32, 6, 2)
    with pytest.raises(ValueError):
        m.get_variational_params()
    with pytest.raises(ValueError):
        m.sample_posterior()
    m.register_parameter("_dummy_param", torch.nn.Parameter(torch.zeros(1)))        # train_hybrid.py:435
    core = ODECore({"a_GI": 0.02})
    assert abs(float(core.a_GI) - 0.02) < 1e-9 and float(core.K_m) == 7.0
    assert [n for n, _ in core.named_buffers()] == list(ODE_PARAM_NAMES)
    core.K_m = torch.tensor(3.0)                                                      # setattr-able (plots/plot_all.py:179-181)
    assert float(core.param_vector()[9]) == 3.0
    assert torch.equal(core.get_steady_state(), torch.tensor([5.0, 60.0, 80.0, 0.0, 0.0, 1.0]))
    assert set(core.check_mass_balance(torch.ones(6), torch.zeros(6))) == {"non_negative", "glucose_range", "insulin_range"}


# ------------------------------------------------------------------ torch evaluations vs goldens
def _load_g0(golden_dir):
    w = np.load(os.path.join(golden_dir, "g0_weights_h64_l4.npz"))
    m = HybridODENN(device="cpu")
    m.load_state_dict({k.replace("__", "."): torch.tensor(w[k]) for k in w.files if "__" in k})
    return m


@pytest.mark.parametrize("tag", ["nogd", "gd", "none"])
def test_ode_core_forward_vs_reference(golden_dir, tag):
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    m = _load_g0(golden_dir)
    ext = None if tag == "none" else {"meal": torch.tensor(r["meal"]), "tVNS": torch.tensor(r["tvns"])}
    if tag == "gd":
        ext["GD"] = torch.tensor(r["gd"])
    with torch.no_grad():
        out = m.ode_core(torch.tensor(r["t"]), torch.tensor(r["x"]), ext).numpy()
    assert rel(out, r[f"ode_f32_{tag}"]) < 2e-6
    md = ODECore().double()
    extd = None if ext is None else {k: v.double() for k, v in ext.items()}
    with torch.no_grad():
        outd = md(torch.tensor(r["t"]).double(), torch.tensor(r["x"]).double(), extd).numpy()
    assert rel(outd, r[f"ode_f64_{tag}"]) < 1e-12


def test_nn_residual_forward_vs_reference(golden_dir):
    r = np.load(os.path.join(golden_dir, "g123_rhs.npz"))
    m = _load_g0(golden_dir)
    x = torch.tensor(r["x"])
    with torch.no_grad():
        out = m.nn_residual(torch.tensor(r["t"]), x, x[:, 3], torch.tensor(r["tvns"])).numpy()
        one = m.nn_residual(torch.tensor(r["t"][0]), x[0], x[0, 3], torch.tensor(r["tvns"][0])).numpy()
    assert np.max(np.abs(out - r["nn_f32"])) < 1e-6
    assert one.shape == (6,) and np.max(np.abs(one - r["nn_f32"][0])) < 1e-6
    assert abs(float(m.nn_residual.regularization_loss(l2_weight=0.1)) - 0.1 * sum(
        float(p.pow(2).sum()) for n, p in m.nn_residual.named_parameters() if n.endswith("weight"))) < 1e-6


# ---- restated reference autograd tests (CPU: these two modules are plain torch) -----------------
def _fd_jacobian(func, x, eps=1e-6):
    x = x.clone().detach().unsqueeze(0)
    f0 = func(x).detach()
    jac = torch.zeros(1, f0.shape[-1], x.shape[-1])
    for i in range(x.shape[-1]):
        xp, xm = x.clone(), x.clone()
        xp[:, i] += eps
        xm[:, i] -= eps
        jac[:, :, i] = (func(xp).detach() - func(xm).detach()) / (2 * eps)
    return jac.squeeze(0)


def test_ode_jacobian_basic():
    torch.manual_seed(0)
    np.random.seed(0)
    core = ODECore()
    state = torch.tensor([5.0, 100.0, 50.0, 20.0, 0.0, 1.0], requires_grad=True)
    t = torch.tensor(0.0)
    ext = {"meal": torch.tensor(0.0), "tVNS": torch.tensor(0.0)}
    f = lambda s: core(t, s, ext)  # noqa: E731
    jac_auto = torch.autograd.functional.jacobian(f, state)
    jac_fd = _fd_jacobian(f, state)
    assert torch.allclose(jac_auto, jac_fd, rtol=0.1, atol=0.1)
    # and against the analytic Jacobian the adjoint kernel uses (exact check, fp64)
    cd = ODECore().double()
    xd = state.detach().double()
    J = torch.autograd.functional.jacobian(lambda s: cd(t.double(), s, None), xd)
    G, I, Glu, GLP1, _, F = xd.tolist()
    assert abs(J[1, 0] - (1 + float(cd.rho) * GLP1) * float(cd.a_GI)) < 1e-12 and abs(J[3, 0] - 9 * 7 / (7 + G) ** 2) < 1e-9
    assert abs(J[2, 3] + float(cd.E_max) * 50 / (50 + GLP1) ** 2 * (Glu - 80)) < 1e-9 and J[4].abs().sum() == 0


def test_ode_jacobian_batch():
    torch.manual_seed(0)
    core = ODECore()
    states = torch.randn(4, 6) * 0.1 + torch.tensor([[5.0, 100.0, 50.0, 20.0, 0.0, 1.0]])
    states.requires_grad = True
    out = core(torch.zeros(4), states, {"meal": torch.zeros(4), "tVNS": torch.zeros(4)})
    assert out.shape == (4, 6)
    out.sum().backward()
    assert states.grad is not None and not torch.any(torch.isnan(states.grad))


def test_ode_jacobian_with_inputs():
    core = ODECore()
    state = torch.tensor([8.0, 150.0, 40.0, 30.0, 0.5, 1.2], requires_grad=True)
    t = torch.tensor(1.0)
    a = core(t, state, {"meal": torch.tensor(10.0), "tVNS": torch.tensor(0.0)})
    b = core(t, state, {"meal": torch.tensor(0.0), "tVNS": torch.tensor(1.0)})
    assert not torch.allclose(a, b)


def test_ode_stability():
    core = ODECore()
    for s in ([20.0, 500.0, 200.0, 100.0, 2.0, 5.0], [2.0, 10.0, 10.0, 5.0, 0.0, 0.1], [5.0, 100.0, 50.0, 20.0, 0.0, 1.0]):
        state = torch.tensor(s, requires_grad=True)
        out = core(torch.tensor(0.0), state, {"meal": torch.tensor(0.0), "tVNS": torch.tensor(0.0)})
        assert torch.isfinite(out).all()
        out.sum().backward()
        assert torch.isfinite(state.grad).all()


def test_nn_residual_gradients():
    torch.manual_seed(0)
    np.random.seed(0)
    net = NNResidual(hidden_dim=32, n_layers=2)
    with torch.no_grad():
        for p in net.parameters():
            p.data.normal_(0, 0.01)
    t, state, glp1, tvns = torch.rand(4), torch.randn(4, 6), torch.rand(4) * 50, torch.rand(4)
    for v in (t, state, glp1, tvns):
        v.requires_grad = True
    out = net(t, state, glp1, tvns)
    assert out.shape == (4, 6)
    out.sum().backward()
    for v in (state, glp1, tvns):
        assert v.grad is not None and not torch.any(torch.isnan(v.grad)) and torch.any(v.grad != 0)
    assert t.grad is not None and not torch.any(torch.isnan(t.grad))


# ------------------------------------------------------------------ VI bookkeeping vs goldens (G6)
def test_variational_parameters_vs_reference(golden_dir):
    names = json.load(open(os.path.join(golden_dir, "g6_vi_names.json")))
    g = np.load(os.path.join(golden_dir, "g6_vi.npz"))
    torch.manual_seed(0)
    m = HybridODENN(nn_hidden=16, nn_layers=2, use_variational=True, device="cpu")
    vp = m.variational_params
    assert list(vp.param_shapes.keys()) == names["names"]
    mu, ls = m.get_variational_params()
    assert mu.numel() == names["latent_dims"] == 542
    assert abs(float(vp.kl_divergence()) - float(g["kl_init"])) < 1e-3 * abs(float(g["kl_init"]))
    with torch.no_grad():
        for n in names["names"]:
            vp.means[n].copy_(torch.tensor(g["mean__" + n]))
            vp.log_stds[n].copy_(torch.tensor(g["logstd__" + n]))
    assert abs(float(vp.kl_divergence()) - float(g["kl_perturbed"])) < 1e-5 * abs(float(g["kl_perturbed"]))
    mu, ls = vp.get_flattened_params()
    assert np.array_equal(mu.detach().numpy(), g["mu_flat"]) and np.array_equal(ls.detach().numpy(), g["log_sigma_flat"])
    torch.manual_seed(2)
    s = m.sample_posterior(1)[0]
    for n in names["names"]:
        assert np.allclose(s[n].detach().numpy(), g["sample__" + n], rtol=1e-6, atol=1e-7)
    assert float(vp.kl_divergence().requires_grad) == 1.0


def test_inference_package_surface_and_checkpoint_without_gpu(tmp_path):
    """`from inference import VariationalInference` resolves to the mirror (no arviz needed); constructor contract,
    history layout and checkpoint round trip are host-side and need no GPU; the sampler placeholders of the reference's
    inference/mcmc.py are out of scope: without the reference on sys.path the import names the module and says where it
    lives (an ImportError, as any missing module; never a stub)."""
    import torch
    import inference
    import models
    from inference.vi import VariationalInference
    assert inference.VariationalInference is VariationalInference
    with pytest.raises(ImportError, match=r"inference\.mcmc .*outside the accelerated path"):
        inference.run_nuts
    with pytest.raises(ImportError, match=r"inference\.mcmc"):
        from inference.mcmc import run_nuts  # noqa: F401
    with pytest.raises(AttributeError):
        inference.no_such_name
    with pytest.raises(ValueError, match="use_variational=True"):
        VariationalInference(models.HybridODENN(nn_hidden=8, nn_layers=1, device="cpu"))
    m = models.HybridODENN(nn_hidden=8, nn_layers=1, use_variational=True, device="cpu")
    vi = VariationalInference(m, learning_rate=5e-3, device=torch.device("cpu"))
    assert vi.history == {"elbo": [], "kl": [], "log_likelihood": []} and vi.learning_rate == 5e-3
    assert vi.variational_params is m.variational_params and isinstance(vi.optimizer, torch.optim.Adam)
    assert len(vi.sample_posterior(3)) == 3
    vi.history["elbo"].append(-1.5)
    p = str(tmp_path / "ck.pt")
    vi.save_checkpoint(p)
    vi2 = VariationalInference(models.HybridODENN(nn_hidden=8, nn_layers=1, use_variational=True, device="cpu"), device=torch.device("cpu"))
    vi2.load_checkpoint(p)
    assert vi2.history["elbo"] == [-1.5]
    for (k, a), (_, b) in zip(vi.variational_params.state_dict().items(), vi2.variational_params.state_dict().items()):
        assert torch.equal(a, b), k
    if not torch.cuda.is_available():          # the ELBO itself needs the device
        import hode
        batch = {"initial_state": torch.zeros(1, 6), "observations": torch.zeros(1, 3, 6), "time_points": torch.linspace(0, 1, 3)}
        with pytest.raises(hode.HodeError):
            vi.elbo(batch, n_samples=1)


def test_inference_is_a_merged_package_missing_submodules_fall_through(tmp_path):
    """The reference's callers import `inference.mcmc` next to `inference.vi` (train/train_hybrid.py:32-33).  The mirror
    ships no mcmc: with a tree that has `inference/mcmc.py` APPENDED to sys.path after the first import of the package (as the
    reference's scripts do, :28), `inference.vi` / `models.*` still resolve here and `inference.mcmc` resolves there."""
    import subprocess
    import sys
    other = tmp_path / "other_root"
    (other / "inference").mkdir(parents=True)
    (other / "inference" / "__init__.py").write_text("raise RuntimeError('the shadowed package __init__ must never run')\n")
    (other / "inference" / "mcmc.py").write_text("def run_nuts(*a, **k):\n    return 'their sampler'\n")
    (other / "inference" / "vi.py").write_text("raise RuntimeError('vi must resolve to the mirror')\n")
    (other / "models").mkdir()
    (other / "models" / "__init__.py").write_text("raise RuntimeError('models must resolve to the mirror')\n")
    code = f"""
import sys, importlib.util
import inference                                   # imported BEFORE the other root is on sys.path
sys.path.append({str(other)!r})
from models.hybrid_ode_nn import HybridODENN
from models.ode_core import ODECore
from inference.vi import VariationalInference
from inference.mcmc import run_nuts
import models, inference.vi, inference.mcmc
assert models.__file__.startswith({PKG!r}), models.__file__
assert inference.vi.__file__.startswith({PKG!r}), inference.vi.__file__
assert inference.mcmc.__file__.startswith({str(other)!r}), inference.mcmc.__file__
assert run_nuts() == 'their sampler' and inference.run_nuts is run_nuts
from inference import run_nuts as again, VariationalInference as V
assert again is run_nuts and V is VariationalInference
print('merged ok')
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=PKG), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0 and "merged ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("restore_best", [True, False])
def test_vi_train_early_stopping_and_restore_semantics(restore_best):
    """`VariationalInference.train` (reference inference/vi.py:157-260): validation ELBO per epoch, patience counter, stop after
    `early_stopping_patience` epochs without improvement, history holds the TRAIN means of the epochs run.  The reference keeps
    `best_state = state_dict()` -- live tensors -- so it ends on the last epoch's posterior (`restore_best=False` here); the
    mirror's default restores the best epoch.  The ELBO itself (GPU) is replaced by a script of values: host logic only."""
    import torch
    from inference.vi import VariationalInference
    m = HybridODENN(nn_hidden=8, nn_layers=1, use_variational=True, device="cpu")
    vi = VariationalInference(m, device=torch.device("cpu"), restore_best=restore_best)
    name = next(iter(vi.variational_params.state_dict()))
    epoch = {"n": 0}

    def fake_train_step(batch, n_samples=5):
        epoch["n"] += 1
        with torch.no_grad():
            for p in vi.variational_params.parameters():
                p.fill_(float(epoch["n"]))
        return {"loss": -float(epoch["n"]), "elbo": float(epoch["n"]), "kl": 0.5, "log_likelihood": float(epoch["n"]) + 0.5}
    val_script = [1.0, 3.0, 2.0, 2.5, 0.0, 9.0, 9.0]       # best = epoch 2; patience 3 stops after epoch 5

    def fake_elbo(batch, n_samples=5, noise_sigma=1.0):
        e = torch.tensor(val_script[epoch["n"] - 1])
        return e, {"elbo": e, "kl": torch.tensor(0.0), "log_likelihood": e}
    vi.train_step, vi.elbo = fake_train_step, fake_elbo
    loader = [{"x": torch.zeros(1)}]
    vi.train(loader, val_loader=loader, epochs=7, n_samples=2, early_stopping_patience=3, verbose=False)
    assert epoch["n"] == 5                                            # epochs 3, 4, 5 did not improve on epoch 2
    assert vi.history["elbo"] == [1.0, 2.0, 3.0, 4.0] and vi.history["kl"] == [0.5] * 4     # the stopping epoch is not logged (vi.py:243-251)
    final = float(vi.variational_params.state_dict()[name].flatten()[0])
    assert final == (2.0 if restore_best else 5.0)


def test_product_library_carries_no_experiment_kernels_and_no_env_dispatch():
    """libhode.so = production kernels only; the experiment kernels of csrc/lab/ and their HODE_FWD / HODE_BWD / HODE_BWD_WT
    switches exist in hode/lab/libhode_lab.so alone (csrc/Makefile `lab`).  Checked on the binaries: symbol tables and strings."""
    import subprocess
    from hode import _build
    assert _build.is_current()                                    # linked from exactly the sources in this tree

    def text(path):
        syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
        return syms, open(path, "rb").read()
    syms, blob = text(_build.SO)
    for marker in (b"HODE_FWD", b"HODE_BWD", b"HODE_BWD_WT"):
        assert marker not in blob, marker
    for name in ("solve_fwd_wg", "solve_fwd_quad", "solve_fwd_rows", "solve_bwd_split", "split_adjoint_enabled"):
        assert name not in syms, name
    if os.path.exists(_build.LAB_SO):
        lsyms, lblob = text(_build.LAB_SO)
        assert b"HODE_FWD" in lblob and b"HODE_BWD" in lblob
        assert "launch_solve_fwd_wg" in lsyms and "launch_solve_bwd_split" in lsyms
        for name in hode.capi.SYMBOLS:                            # the lab library is the same C ABI
            assert name in lsyms, name


def test_no_dpp_read_sits_closer_than_two_wait_states_behind_its_producer():
    """tools/dpp_hazard_check.py (DPP / lane-swap reads behind their producers; LDS results named before their wait) on the two sources whose hot loops are inline asm (cross-compiles for gfx950, no GPU needed): hipcc
    does not look inside asm statements and may re-order independent ones; a DPP / lane-swap read of a register one instruction
    after the vector instruction that wrote it returns stale lanes -- silently, and only in the instantiations where the
    scheduler happened to do it (round 4: RK4 x three layers x tape)."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "hybrid-ode-for-glp-1-and-glucose_amd", "csrc")
    # the checker itself: a kernel with one hazard of each class it knows (a DPP read right behind its producer; an LDS result named
    # before its wait) must be reported
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        bad = os.path.join(tmp, "bad.hip")
        open(bad, "w").write('''#include <hip/hip_runtime.h>
__global__ void bad_kernel(float *out, const float *in)
{
    __shared__ float sm[64];
    sm[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    float a = in[threadIdx.x + 64], r, c, q, z;
    asm volatile("v_add_f32 %0, %1, %1\\n\\tv_mov_b32_dpp %2, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "=&v"(r), "+v"(a), "=&v"(c));
    unsigned addr = threadIdx.x * 4;
    asm volatile("ds_read_b32 %0, %2\\n\\tv_add_f32 %1, %0, %0\\n\\ts_waitcnt lgkmcnt(0)" : "=&v"(q), "=&v"(z) : "v"(addr));
    out[threadIdx.x] = r + c + q + z;
}
''')
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "dpp_hazard_check.py"), bad], capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "2 hazard(s)" in r.stdout and "v_mov_b32_dpp" in r.stdout and "LDS read" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "dpp_hazard_check.py"), os.path.join(csrc, "hode_solve_fwd.hip"),
                        os.path.join(csrc, "hode_solve_bwd_ws.hip")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 hazard(s)" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
