"""The Sobol consumer (reference plots/plot_all.py:139-196): parameter sets of the 7 mechanistic constants x ONE
patient per set, n_sets == B.  Checked against the ORACLE (fp64, converged tolerances), not against another run of the HIP path:
the forward through the class surface (HybridODENN.forward_ode_sets, network shared by all sets), the study's three outputs
(:191-193), and the adjoint of a launch with as many parameter sets as trajectories -- more sets than gradient rows for the
tuned kernels' workgroups would be a problem only above 1 024 sets; below, every set is a workgroup of its own.  `-m gpu`."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import bench  # noqa: E402  (the benchmark's Saltelli design, inputs and weights: the leg `sobol` of bench.py runs the same workload)
from oracle import oracle as O  # noqa: E402  (checker only)

H, L = 64, 4
NCPU = max(1, min(os.cpu_count() or 1, 16))


def _ode_rows(sets):
    """[S, 17] fp64: the model's constants with the seven of the study replaced -- rounded to fp32 first, as the reference's
    `setattr(model.ode_core, name, torch.tensor(value))` (plot_all.py:179-181) and the class surface here do."""
    ode = np.tile(bench.ODE_DEFAULT.numpy().astype(np.float64), (len(sets), 1))
    for i, name in enumerate(bench.SOBOL_NAMES):
        ode[:, bench.ODE_NAMES.index(name)] = sets[:, i].astype(np.float32).astype(np.float64)
    return ode


def _oracle_sets(ode_rows, nn, x0, t, meal, tvns, cot=None):
    def work(s):
        sol = O.solve(x0[None], t, meal, tvns, None, ode_rows[s], nn, H, L, rtol=1e-10, atol=1e-12, dtype=np.float64,
                      want_tape=cot is not None)
        if cot is None:
            return sol.y[0], None
        return sol.y[0], O.solve_bwd(sol, cot[s:s + 1], want_gode=True)
    with ThreadPoolExecutor(NCPU) as ex:
        return list(ex.map(work, range(len(ode_rows))))


def test_sobol_64_sets_through_the_class_surface_vs_oracle():
    sets = bench.saltelli_sets(1024)[:64]            # the first four base samples of the design: blocks A, AB_i, BA_i, B
    m = bench.class_model(torch.device("cuda"))
    x0, t, meal, tvns = bench.sobol_inputs()
    ode_sets = {k: torch.as_tensor(sets[:, i], dtype=torch.float32) for i, k in enumerate(bench.SOBOL_NAMES)}
    y = m.forward_ode_sets(ode_sets, x0.cuda(), t.cuda(), {"meal": meal.cuda(), "tVNS": tvns.cuda()})
    assert tuple(y.shape) == (64, 61, 6) and m.solve_failures() == 0
    nn = bench.synth_weights(0).numpy().astype(np.float64)
    ref = _oracle_sets(_ode_rows(sets), nn, x0.numpy().astype(np.float64), t.numpy().astype(np.float64),
                       meal.numpy().astype(np.float64), tvns.numpy().astype(np.float64))
    yr = np.stack([r[0] for r in ref])
    yk = y.cpu().numpy().astype(np.float64)
    err = np.max(np.abs(yk - yr) / (np.abs(yr) + 1e-3))
    assert err < 1e-4, err                                                    # north_star: 1e-3 for fp32
    # the sets really differ (the kernel read each trajectory's own constants)
    assert np.max(np.abs(yk[0] - yk[1])) > 1e-3
    # the three outputs of the study
    ok, orf = bench.sobol_outputs(y).cpu().numpy().astype(np.float64), bench.sobol_outputs(yr)
    ref_np = np.stack([np.trapz(yr[:, :, 0], dx=5 / 60, axis=1), yr[:, :, 1].max(1), yr[:, 6:, 3].mean(1)], 1)     # plot_all.py:191-193 verbatim forms
    assert np.allclose(orf, ref_np, rtol=1e-12)
    assert np.max(np.abs(ok - orf) / np.abs(orf)) < 1e-4


def test_shared_network_launch_equals_one_network_copy_per_set():
    """HODE_LAYERS_NN_SHARED reads one network for all sets; forward_param_sets (list of dicts) ships a copy per set:
    the same trajectories bit for bit, and equal to model.forward after setattr -- what the reference's loop does."""
    sets = bench.saltelli_sets(64)[5:22]
    m = bench.class_model(torch.device("cuda"))
    x0, t, meal, tvns = (v.cuda() for v in bench.sobol_inputs())
    ext = {"meal": meal, "tVNS": tvns}
    ode_sets = {k: torch.as_tensor(sets[:, i], dtype=torch.float32) for i, k in enumerate(bench.SOBOL_NAMES)}
    y = m.forward_ode_sets(ode_sets, x0, t, ext)
    dicts = [{f"ode_{k}": torch.tensor(float(np.float32(sets[s, i]))) for i, k in enumerate(bench.SOBOL_NAMES)} for s in range(len(sets))]
    with torch.no_grad():
        y2 = m.forward_param_sets(dicts, x0, t, ext)
        for s in (0, 7):
            for i, k in enumerate(bench.SOBOL_NAMES):
                setattr(m.ode_core, k, torch.tensor(float(np.float32(sets[s, i])), device="cuda"))
            assert torch.equal(m(x0.unsqueeze(0), t, ext)[0], y[s])
    assert torch.equal(y, y2)
    with pytest.raises(ValueError):
        m.forward_ode_sets({"not_a_constant": torch.ones(3)}, x0, t, ext)


def test_adjoint_with_one_patient_per_parameter_set_vs_oracle():
    """32 sets x 1 patient, n_sets == B: per-set network and ODE-constant gradients, per-trajectory gx0, against the oracle's
    adjoint of every set on its own (fp64, 1e-10 / 1e-12).  north_star: adjoint gradients to 1e-4."""
    import hode
    S = 32
    sets = bench.saltelli_sets(1024)[100:100 + S]
    ode_rows = _ode_rows(sets)
    x0, t, meal, tvns = bench.sobol_inputs()
    nn = bench.synth_weights(0)
    dev = torch.device("cuda")
    sol = hode.solve_fwd(x0.repeat(S, 1).to(dev), t.to(dev), meal.repeat(S, 1).to(dev), tvns.repeat(S, 1).to(dev), None,
                         torch.as_tensor(ode_rows, dtype=torch.float32, device=dev).reshape(-1), nn.repeat(S).to(dev), H, L,
                         n_sets=S, want_tape=True)
    assert int(sol.status.max()) == 0
    cot = np.random.default_rng(5).standard_normal((S, 61, 6))
    gx0, gnn, gode = hode.solve_bwd(sol, torch.as_tensor(cot, dtype=torch.float32, device=dev), want_gode=True)
    P = hode.n_params(H, L)
    gnn, gode, gx0 = gnn.view(S, P).cpu().numpy(), gode.view(S, 17).cpu().numpy(), gx0.cpu().numpy()
    ref = _oracle_sets(ode_rows, nn.numpy().astype(np.float64), x0.numpy().astype(np.float64), t.numpy().astype(np.float64),
                       meal.numpy().astype(np.float64), tvns.numpy().astype(np.float64), cot=cot)
    worst = {"gx0": 0.0, "gnn": 0.0, "gode": 0.0}
    for s, (_, (rx, rn, ro)) in enumerate(ref):
        worst["gx0"] = max(worst["gx0"], np.linalg.norm(gx0[s] - rx[0]) / np.linalg.norm(rx[0]))
        worst["gnn"] = max(worst["gnn"], np.linalg.norm(gnn[s] - rn) / np.linalg.norm(rn))
        # the constants' gradients span ten orders of magnitude (d/d rho against d/d IGD_50): each against the set's largest
        worst["gode"] = max(worst["gode"], np.max(np.abs(gode[s] - ro)) / np.max(np.abs(ro)))
    assert worst["gx0"] < 1e-4 and worst["gnn"] < 1e-4 and worst["gode"] < 1e-4, worst
    # sets are independent: set 3 on its own gives the same gradient bits
    one = hode.solve_fwd(x0[None].to(dev), t.to(dev), meal.to(dev), tvns.to(dev), None,
                         torch.as_tensor(ode_rows[3], dtype=torch.float32, device=dev), nn.to(dev), H, L, want_tape=True)
    g1 = hode.solve_bwd(one, torch.as_tensor(cot[3:4], dtype=torch.float32, device=dev), want_gode=True)
    assert torch.equal(g1[0][0].cpu(), torch.as_tensor(gx0[3]))
    assert np.allclose(g1[1].cpu().numpy(), gnn[3], rtol=2e-5, atol=1e-9 * np.abs(gnn[3]).max())
