#!/usr/bin/env python3
"""Capture golden vectors of the reference's VariationalInference (inference/vi.py:19-340).

Runs ONLY in the build container (needs /root/reference, read-only).  `inference/__init__.py` imports arviz (absent
here and staying absent), so vi.py is loaded directly from its file; it needs only torch, numpy and tqdm.  Writes DATA
(inputs, parameter values, expected outputs) to tests/golden/g6_vi_elbo.npz -- no source text.

  * model: HybridODENN(nn_hidden=16, nn_layers=2, use_variational=True, prior_params=<ODE constants at their defaults>)
    on the CPU, seeded; means / log-stds perturbed under a second seed (all stored);
  * batch: B = 2, T = 10, meal = tVNS = 0 (with meals the reference's default-tolerance DOP853 answer moves by ~1e-2
    with the last bit of the RHS; meal-free it is converged to ~3e-7, so the ELBO value can be pinned);
  * elbo(batch, n_samples=2, noise_sigma=0.5) under torch.manual_seed(7): value, KL, log-likelihood, and the two
    parameter draws it consumed (same seed replayed);
  * posterior_predictive(x0, t, ext, n_samples=3) under torch.manual_seed(8): mean and std;
  * train_step(batch, n_samples=2) under torch.manual_seed(9): metrics and the variational parameters afterwards
    (the reference's likelihood term carries no gradient -- SURVEY F3 -- so only the KL moves them).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)
from models.hybrid_ode_nn import HybridODENN  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_vi", os.path.join(REF, "inference", "vi.py"))
ref_vi = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref_vi)

torch.set_num_threads(1)
CPU = torch.device("cpu")
ODE8 = {"a_GI": 0.0104, "k_I": 0.025, "rho": 0.003, "E_max": 0.1, "EC_50": 50.0, "V_max": 9.0, "K_m": 7.0, "k_L": 0.02}
prior = {f"ode_{n}": {"mean": v, "std": 0.05 * v} for n, v in ODE8.items()}

torch.manual_seed(0)
model = HybridODENN(nn_hidden=16, nn_layers=2, use_variational=True, prior_params=prior, device=CPU)
vp = model.variational_params
names = list(vp.param_shapes.keys())
torch.manual_seed(1)
with torch.no_grad():
    for n in names:
        if n.startswith("nn_"):
            vp.means[n].add_(0.05 * torch.randn_like(vp.means[n]))
        vp.log_stds[n].add_(0.1 * torch.randn_like(vp.log_stds[n]))
arrs = {}
for n in names:
    arrs["mean__" + n] = vp.means[n].detach().numpy().copy()
    arrs["logstd__" + n] = vp.log_stds[n].detach().numpy().copy()

g = torch.Generator().manual_seed(3)
B, T = 2, 10
x0 = torch.tensor([5.0, 60.0, 80.0, 10.0, 0.0, 1.0]) * (1 + 0.05 * torch.randn(B, 6, generator=g))
t = torch.linspace(0, 1.5, T)
ext = {"meal": torch.zeros(B, T), "tVNS": torch.zeros(B, T)}
obs = x0[:, None, :].expand(B, T, 6) * (1 + 0.02 * torch.randn(B, T, 6, generator=g))
batch = {"initial_state": x0, "observations": obs, "time_points": t, "external_inputs": ext}
arrs.update(x0=x0.numpy(), t=t.numpy(), obs=obs.numpy(), meal=ext["meal"].numpy(), tvns=ext["tVNS"].numpy())

vi = ref_vi.VariationalInference(model, learning_rate=1e-2, device=CPU)
torch.manual_seed(7)
elbo, comp = vi.elbo(batch, n_samples=2, noise_sigma=0.5)
torch.manual_seed(7)
draws = [vp.sample(1)[0] for _ in range(2)]
arrs.update(elbo=np.float64(elbo.item()), kl=np.float64(comp["kl"].item()), log_likelihood=np.float64(comp["log_likelihood"].item()),
            elbo_requires_grad=np.bool_(elbo.requires_grad), n_samples=2, noise_sigma=0.5)
for i, d in enumerate(draws):
    for n in names:
        arrs[f"draw{i}__{n}"] = d[n].detach().numpy()
with torch.no_grad():
    for i, d in enumerate(draws):
        arrs[f"pred{i}"] = model.forward_with_params({k: v.detach() for k, v in d.items()}, x0, t, ext).numpy()

torch.manual_seed(8)
mean, std = vi.posterior_predictive(x0, t, ext, n_samples=3)
arrs.update(pp_mean=mean.numpy(), pp_std=std.numpy())

torch.manual_seed(9)
metrics = vi.train_step(batch, n_samples=2)
arrs.update(step_loss=np.float64(metrics["loss"]), step_elbo=np.float64(metrics["elbo"]), step_kl=np.float64(metrics["kl"]),
            step_ll=np.float64(metrics["log_likelihood"]))
for n in names:
    arrs["after__mean__" + n] = vp.means[n].detach().numpy().copy()
    arrs["after__logstd__" + n] = vp.log_stds[n].detach().numpy().copy()
np.savez_compressed(os.path.join(OUT, "g6_vi_elbo.npz"), **arrs)
print("elbo", elbo.item(), "kl", comp["kl"].item(), "ll", comp["log_likelihood"].item(), "requires_grad", elbo.requires_grad)
print("pp std max", float(std.max()), "step", metrics)
print("wrote g6_vi_elbo.npz", os.path.getsize(os.path.join(OUT, "g6_vi_elbo.npz")) // 1024, "KiB")
