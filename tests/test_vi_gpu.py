"""VariationalInference mirror (inference/vi.py) on the HIP path against values captured from the reference's own
VariationalInference (tests/golden/g6_vi_elbo.npz, tools/capture_golden_vi.py): ELBO value and components for S = 2,
B = 2, T = 10; posterior predictive; one train_step.  The model lives on 'cpu' (variational parameters and torch's RNG
stream as in the reference), the solves run on the GPU."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def setup(golden_dir):
    import models
    from inference import VariationalInference
    g = np.load(os.path.join(golden_dir, "g6_vi_elbo.npz"))
    ode8 = {"a_GI": 0.0104, "k_I": 0.025, "rho": 0.003, "E_max": 0.1, "EC_50": 50.0, "V_max": 9.0, "K_m": 7.0, "k_L": 0.02}
    prior = {f"ode_{n}": {"mean": v, "std": 0.05 * v} for n, v in ode8.items()}
    torch.manual_seed(0)
    m = models.HybridODENN(nn_hidden=16, nn_layers=2, use_variational=True, prior_params=prior, device="cpu")
    vp = m.variational_params
    with torch.no_grad():
        for n in vp.param_shapes:
            vp.means[n].copy_(torch.tensor(g["mean__" + n]))
            vp.log_stds[n].copy_(torch.tensor(g["logstd__" + n]))
    batch = {"initial_state": torch.tensor(g["x0"]), "observations": torch.tensor(g["obs"]), "time_points": torch.tensor(g["t"]),
             "external_inputs": {"meal": torch.tensor(g["meal"]), "tVNS": torch.tensor(g["tvns"])}}
    return m, VariationalInference(m, learning_rate=1e-2, device=torch.device("cpu")), batch, g


def test_elbo_value_and_components_vs_reference(setup):
    m, vi, batch, g = setup
    m.adjoint = False                       # value parity: the reference's likelihood is detached
    torch.manual_seed(7)
    elbo, comp = vi.elbo(batch, n_samples=2, noise_sigma=0.5)
    assert set(comp) == {"elbo", "kl", "log_likelihood"} and elbo.requires_grad      # through the KL, as in the reference
    assert abs(float(comp["kl"]) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    assert abs(float(comp["log_likelihood"]) - float(g["log_likelihood"])) < 2e-4 * abs(float(g["log_likelihood"]))
    assert abs(float(elbo) - float(g["elbo"])) < 1e-4 * abs(float(g["elbo"]))
    # the draws behind it are the reference's draws (same RNG consumption), and so are the trajectories
    torch.manual_seed(7)
    draws = [m.variational_params.sample(1)[0] for _ in range(2)]
    for i, d in enumerate(draws):
        for n, v in d.items():
            # same RNG stream as the reference; exp(log_sigma) may differ in the last bit between host CPUs (vectorised libm)
            np.testing.assert_allclose(v.detach().numpy(), g[f"draw{i}__{n}"], rtol=1e-5, atol=1e-8, err_msg=n)
        with torch.no_grad():
            y = m.forward_with_params({k: v.detach() for k, v in d.items()}, batch["initial_state"], batch["time_points"],
                                      batch["external_inputs"])
        np.testing.assert_allclose(y.numpy(), g[f"pred{i}"], rtol=2e-5, atol=1e-5)
    # with the adjoint on, the SAME value -- plus a likelihood gradient the reference does not have
    m.adjoint = True
    torch.manual_seed(7)
    elbo2, _ = vi.elbo(batch, n_samples=2, noise_sigma=0.5)
    assert abs(float(elbo2) - float(elbo)) < 1e-6 * abs(float(elbo))
    elbo2.backward()
    g_adj = m.variational_params.means["nn_network_4_weight"].grad.clone()
    m.variational_params.zero_grad()
    m.adjoint = False
    torch.manual_seed(7)
    vi.elbo(batch, n_samples=2, noise_sigma=0.5)[0].backward()
    g_kl = m.variational_params.means["nn_network_4_weight"].grad.clone()
    assert float((g_adj - g_kl).abs().max()) > 1e-3 * float(g_kl.abs().max())


def test_posterior_predictive_vs_reference(setup):
    m, vi, batch, g = setup
    torch.manual_seed(8)
    mean, std = vi.posterior_predictive(batch["initial_state"], batch["time_points"], batch["external_inputs"], n_samples=3)
    assert mean.shape == (2, 10, 6) and std.shape == (2, 10, 6)
    np.testing.assert_allclose(mean.numpy(), g["pp_mean"], rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(std.numpy(), g["pp_std"], rtol=2e-3, atol=2e-5)
    assert len(vi.sample_posterior(4)) == 4


def test_train_step_detached_matches_reference_and_adjoint_trains_the_likelihood(setup):
    m, vi, batch, g = setup
    m.adjoint = False
    torch.manual_seed(9)
    met = vi.train_step(batch, n_samples=2)
    assert set(met) == {"loss", "elbo", "kl", "log_likelihood"}
    assert abs(met["kl"] - float(g["step_kl"])) < 1e-5 * abs(float(g["step_kl"]))
    assert abs(met["log_likelihood"] - float(g["step_ll"])) < 2e-4 * abs(float(g["step_ll"]))
    assert abs(met["loss"] - float(g["step_loss"])) < 1e-4 * abs(float(g["step_loss"]))
    for n in m.variational_params.param_shapes:                # KL gradient -> clip 5.0 -> Adam(lr 1e-2), as in the reference
        np.testing.assert_allclose(m.variational_params.means[n].detach().numpy(), g["after__mean__" + n], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(m.variational_params.log_stds[n].detach().numpy(), g["after__logstd__" + n], rtol=1e-5, atol=1e-6)
    # adjoint on: a few steps on a tight-noise likelihood raise the ELBO's likelihood term (the reference cannot: F3)
    m.adjoint = True
    with torch.no_grad():
        for n, p in m.variational_params.log_stds.items():
            p.fill_(-7.0)
    torch.manual_seed(3)
    ll0 = vi.train_step(batch, n_samples=2)["log_likelihood"]
    for _ in range(15):
        ll = vi.train_step(batch, n_samples=2)["log_likelihood"]
    assert ll > ll0


def test_train_loop_history_early_stopping_and_checkpoint(setup, tmp_path):
    m, vi, batch, g = setup
    loader = [dict(batch, external_inputs=dict(batch["external_inputs"])) for _ in range(2)]
    vi.train(loader, val_loader=loader[:1], epochs=3, n_samples=1, early_stopping_patience=1, verbose=False)
    n = len(vi.history["elbo"])
    assert 1 <= n <= 3 and len(vi.history["kl"]) == n and len(vi.history["log_likelihood"]) == n
    assert hasattr(vi, "best_state")
    path = str(tmp_path / "vi.pt")
    vi.save_checkpoint(path)
    before = {k: v.clone() for k, v in m.variational_params.state_dict().items()}
    hist = {k: list(v) for k, v in vi.history.items()}
    with torch.no_grad():
        for p in m.variational_params.parameters():
            p.add_(1.0)
    vi.history = {"elbo": [], "kl": [], "log_likelihood": []}
    vi.load_checkpoint(path)
    assert vi.history == hist
    for k, v in m.variational_params.state_dict().items():
        assert torch.equal(v, before[k])
    ck = torch.load(path, weights_only=True)
    assert set(ck) == {"variational_params", "optimizer", "history"}        # the reference's checkpoint layout (vi.py:314-327)
