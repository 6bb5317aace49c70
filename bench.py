#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): patient-trajectories/s of the
batched 6-state DP5(4) solve (ODE + 4x64 MLP residual, fp32, T = 241 grid points = 240 intervals).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path over one synthetic cohort shard
(BASELINE config[1]: 4 096 patients per GPU, forward solve).  Inputs are resident in HBM before
the timed region.  Weak scaling: every rank integrates its own 4 096-patient shard, no data-path
collective in the forward path.  After the headline region the same shard is pushed through the
TRAINING step of config[2]/[3] (forward with tape -> fused MSE -> adjoint -> one RCCL all-reduce
of the 54 KB gradient buffer -> fused clip+Adam) and reported under "train_step".

At N = 1 two more legs follow: "vi_step" (BASELINE config 5 at its per-GPU size: 8 192 patients x 16 VI draws, ELBO +
reparameterised gradient + Adam) and "data_side" (65 536-subject 4GI cohort generation + dataset windows).

Rank 0 prints ONE JSON line.  `roofline` prices the solve kernel against the fp32 compute peak
(the binding roof: ~5 000 flop/byte) and also carries the HBM reading;
`cpu_baseline` times the oracle (C port of the same algorithm) on the host cores, N = 1 only.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

# dmabuf IPC is the only mode the host driver supports (RCCL / device-tensor sharing across processes): the default has to be
# in the environment before anything initialises the HIP runtime, i.e. before torch is even imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

H, L, T = 64, 4, 241
FLOP_PER_RHS = 2 * (9 * 64 + 3 * 64 * 64 + 6 * 64) + 80      # 26 576: MLP MACs x2 + mechanistic terms
FLOP_PER_STEP_ALGEBRA = 400                                   # stage sums, error norm, controller
BYTES_FWD_PER_TRAJ = 24 + 964 + 964 + 241 * 6 * 4            # x0 + meal row + tVNS row + y  = 7 736 B
BYTES_BWD_PER_TRAJ = 2 * 241 * 6 * 4 + 2 * 964 + 24          # y + dLoss/dy + meal/tVNS rows + gx0 = 13 520 B
BYTES_TRAIN_PER_TRAJ = BYTES_FWD_PER_TRAJ + BYTES_BWD_PER_TRAJ  # 21 256 B
PEAK_FP32_TFLOPS = 157.3                                      # MI355X_MICROARCH.md: fp32 vector == f32-MFMA dense peak
PEAK_FP64_TFLOPS = 78.6                                       # fp64 vector = half the fp32 vector rate (MI355X data sheet; the guide lists fp32 only)
PEAK_HBM_GBS = 8000.0
# adjoint, per stage of an accepted step: outer products dW += delta (x) h and the W^T delta products = 2x the MACs of the
# forward RHS (nothing is recomputed: activations come from the stage tape), plus the mechanistic J^T and edge layers
FLOP_PER_ADJ_STAGE = 2 * FLOP_PER_RHS
# stage tape record = h_1..h_L (L rows x 64 lanes) + the stage state (8) reals per stage; tape entry 32 B + 4 B interval index per step
STAGE_REC_BYTES = (L * 64 + 8) * 4
# 4GI generator (K7): per RHS ~60 add/mul/div + 3 pow (exp(p log x), ~40 flop each); DP5(4) stage algebra of 8 states
FLOP_PER_4GI_RHS = 60 + 3 * 40
FLOP_PER_4GI_STEP = 6 * FLOP_PER_4GI_RHS + 8 * 2 * (21 + 7) + 40


def _strip_c_comments(text):
    """C / C++ source without comments and with runs of white space collapsed (string literals are kept as they are)."""
    import re
    pat = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)
    text = pat.sub(lambda m: m.group(0) if m.group(0)[0] in "\"'" else " ", text)
    return re.sub(r"\s+", " ", text)


def kernel_source_sha(root=None):
    """sha256 over the CODE of the kernel sources (comments and white space do not count): profiles/pmc_traffic.json carries
    the value it was measured at, so a PMC reading taken on other kernels is never reported as this run's traffic."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root or ROOT, "hybrid-ode-for-glp-1-and-glucose_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(_strip_c_comments(open(f, "r", errors="replace").read()).encode())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """HBM bytes per launch from the committed PMC passes (tools/profile_gpu.sh -> tools/summarize_profile.py), or None when
    the kernels have changed since (or the file is missing)."""
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(pmc))
    except Exception:
        return None, "no PMC file"
    if d.get("kernel_source_sha") != kernel_source_sha():
        return None, f"stale: PMC passes were taken at kernel sources {d.get('kernel_source_sha')}, this run is {kernel_source_sha()}"
    return d.get(key), d.get("source")


def ensure_built(may_build=True):
    """Build BEFORE any GPU / collective call when libhode.so is missing OR was linked from other sources than the ones in this
    tree (csrc/Makefile stamps the binary with a hash of its sources; hode/_build.py compares) -- a stale binary is never
    benchmarked.  Every rank takes the same path: an exclusive file lock, then `make` (the Makefile links to a temporary name
    and renames it, so no rank can dlopen a half-written library).  may_build=False (--no-build / HODE_NO_BUILD=1; every
    invocation under rocprofv3, where the profiler's tool library has initialised the GPU before Python starts and a child
    process must not be spawned): a missing or stale artefact is an error instead."""
    import subprocess
    from hode import _build
    _build.ensure(may_build=may_build)
    osoname = os.path.join(ROOT, "oracle", "_build", "libhode_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("hode_oracle.c", "hode_oracle_impl.h", "fourgi_oracle.c")]
    if not os.path.exists(osoname) or any(os.path.getmtime(f) > os.path.getmtime(osoname) for f in srcs if os.path.exists(f)):
        if not may_build:
            if not os.path.exists(osoname):
                raise SystemExit(f"{osoname} is missing and --no-build was given: run `make -C oracle` first")
            return
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True, stdout=subprocess.DEVNULL)


def synth_weights(seed=0):
    """G0-style weights: hidden layers xavier_normal(gain 0.1), zero biases, NON-zero output layer."""
    g = torch.Generator().manual_seed(seed)
    parts = []
    dims = [(64, 9)] + [(64, 64)] * 3 + [(6, 64)]
    for i, (o, n) in enumerate(dims):
        std = 0.01 if i == len(dims) - 1 else 0.1 * (2.0 / (o + n)) ** 0.5
        parts.append(torch.randn(o, n, generator=g) * std)
        parts.append(torch.randn(o, generator=g) * 0.01 if i == len(dims) - 1 else torch.zeros(o))
    return torch.cat([p.reshape(-1) for p in parts])


ODE_DEFAULT = torch.tensor([0.0104, 0.025, 0.003, 5.0, 60.0, 0.1, 50.0, 80.0, 9.0, 7.0, 0.02, 0.01, 1000.0, 2.0,
                            0.05, 0.001, 0.01])
ODE_NAMES = ["a_GI", "k_I", "rho", "G_b", "I_b", "E_max", "EC_50", "Glu_b", "V_max", "K_m", "k_L", "k_GE0", "IGD_50", "g", "p_7", "p_8", "p_9"]


def synth_cohort(B, seed):
    """4GI-style synthetic cohort (SURVEY 8d 'physio' regime): x0 = basal*(1+5% noise), 5-min grid over 20 h,
    4 unit meal pulses per patient at random grid indices in 6..234, tVNS = 0."""
    g = torch.Generator().manual_seed(seed)
    base = torch.tensor([5.0, 60.0, 80.0, 10.0, 0.0, 1.0])
    x0 = base * (1 + 0.05 * torch.randn(B, 6, generator=g))
    t = (torch.arange(T, dtype=torch.float64) * (5.0 / 60.0)).float()
    idx = torch.rand(B, 229, generator=g).argsort(dim=1)[:, :4] + 6
    meal = torch.zeros(B, T).scatter_(1, idx, 1.0)
    tvns = torch.zeros(B, T)
    return x0, t, meal, tvns


# ---- the Sobol study of the reference's plots/plot_all.py:139-196, its only workload with a published time
SOBOL_NAMES = ["a_GI", "k_I", "rho", "E_max", "V_max", "K_m", "k_L"]                                        # plot_all.py:139
SOBOL_BOUNDS = [[0.008, 0.012], [0.02, 0.03], [0.002, 0.004], [0.08, 0.12], [7.0, 11.0], [5.5, 8.5], [0.015, 0.025]]   # :140-148


def saltelli_sets(n=1024, seed=0):
    """Saltelli's design for first / total / second-order indices: n (2 D + 2) rows = 16 384 for n = 1 024, D = 7 -- the row
    count and block layout (A, AB_1..AB_D, BA_1..BA_D, B per base sample) of SALib's `saltelli.sample(problem, 1024)`
    (plot_all.py:158).  SALib is not installed here: the base sample comes from SciPy's scrambled Sobol sequence, so the
    values differ from SALib's, the workload (16 384 single-patient solves over the same box) does not."""
    from scipy.stats import qmc
    D = len(SOBOL_NAMES)
    base = qmc.Sobol(d=2 * D, scramble=True, seed=seed).random(n)
    lo, hi = np.array(SOBOL_BOUNDS).T
    A, Bm = lo + base[:, :D] * (hi - lo), lo + base[:, D:] * (hi - lo)
    rows = np.empty((n, 2 * D + 2, D))
    rows[:, 0] = A
    rows[:, -1] = Bm
    for i in range(D):
        rows[:, 1 + i] = A
        rows[:, 1 + i, i] = Bm[:, i]
        rows[:, 1 + D + i] = Bm
        rows[:, 1 + D + i, i] = A[:, i]
    return rows.reshape(-1, D)


def sobol_inputs():
    """plot_all.py:164-165,184-187: x0 = [5, 60, 80, 0, 0, 1], 61 points over 5 h, 75 mmol of glucose at 30 min, no tVNS."""
    x0 = torch.tensor([5.0, 60.0, 80.0, 0.0, 0.0, 1.0])
    t = torch.linspace(0, 5, 61)
    meal = torch.zeros(61)
    meal[6] = 75.0
    return x0, t, meal.unsqueeze(0), torch.zeros(1, 61)


def sobol_outputs(y):
    """plot_all.py:191-193 on [S, 61, 6] trajectories: glucose AUC (trapezoid, dx = 5 / 60), insulin peak, mean GLP-1 from the meal on."""
    xp = torch if torch.is_tensor(y) else np
    auc = ((y[:, 1:, 0] + y[:, :-1, 0]) * 0.5).sum(1) * (5.0 / 60.0)
    peak = y[:, :, 1].max(1)
    peak = peak.values if torch.is_tensor(y) else peak
    return xp.stack([auc, peak, y[:, 6:, 3].mean(1)], 1)


def class_model(dev, seed=0):
    """HybridODENN (the drop-in class) carrying the benchmark's G0-style weights."""
    from models import HybridODENN
    torch.manual_seed(0)
    m = HybridODENN(device=dev)
    with torch.no_grad():
        off, w = 0, synth_weights(seed)
        for p in m.nn_residual.parameters():
            p.copy_(w[off:off + p.numel()].reshape(p.shape))
            off += p.numel()
    return m


def sobol_leg(dev, n=1024, host=None):
    """All 16 384 parameter sets x 1 patient of the Sobol study through HybridODENN.forward_ode_sets: ONE launch, the network
    shared (HODE_LAYERS_NN_SHARED).  The reference runs them one model.forward at a time: "~5-10 minutes" (README.md:248)."""
    sets = saltelli_sets(n)
    m = class_model(dev)
    x0, t, meal, tvns = (v.to(dev) for v in sobol_inputs())
    ode_sets = {k: torch.as_tensor(sets[:, i], dtype=torch.float32, device=dev) for i, k in enumerate(SOBOL_NAMES)}

    def run():
        return m.forward_ode_sets(ode_sets, x0, t, {"meal": meal, "tVNS": tvns})
    y = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    y = run()
    e1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    out3 = sobol_outputs(y)
    torch.cuda.synchronize()
    wall_all = time.perf_counter() - t0
    st = m.last_solve_info
    res = {"workload": f"plots/plot_all.py:139-196: {sets.shape[0]} Saltelli parameter sets (N = {n}, 7 mechanistic constants, second-order design) "
                       "x 1 patient, x0 = [5, 60, 80, 0, 0, 1], 61 points over 5 h, 75 mmol meal pulse at 30 min; DP5(4) 1e-6 / 1e-8, fp32, "
                       "4x64 network shared by all sets",
           "call": "HybridODENN.forward_ode_sets (class surface) -> one hode_solve_fwd_f32 launch, n_sets = B = 16 384",
           "sets": int(sets.shape[0]), "seconds": wall, "seconds_with_outputs": wall_all, "kernel_ms": e0.elapsed_time(e1),
           "value": sets.shape[0] / wall, "unit": "patient-trajectories/s", "trajectories_ok": int((st["status"] == 0).sum()),
           "mean_steps": float(st["nsteps"].float().mean()),
           "outputs_mean": dict(zip(["glucose_auc", "insulin_peak", "glp1_response"], out3.double().mean(0).tolist())),
           "sampling": "SciPy scrambled Sobol base sample in Saltelli's block layout (SALib absent: other values, same workload)",
           "vs_reference_published": {"published": "~5-10 minutes (README.md:248, the reference's only timing that matches its code)",
                                      "speedup_vs_300_s": 300.0 / wall, "speedup_vs_600_s": 600.0 / wall}}
    if host is not None and host.get("sobol_reference_style"):
        r = host["sobol_reference_style"]
        res["reference_style_this_box"] = dict(r, speedup=r["seconds_for_16384_one_thread"] / wall)
    return res


def class_path(dev, x0, t, meal, tvns, headline_ms):
    """What a user of the drop-in CLASSES gets (SURVEY 8d defines the metric on HybridODENN.forward): the forward at the
    benchmark size, and one optimisation step of reference train/train_hybrid.py:237-261 (zero_grad -> loss -> backward ->
    clip_grad_norm_ -> Adam.step) at the reference's own batch (32 windows x 61 points, configs/default.yaml:19) and at 4 096 x 241."""
    m = class_model(dev)
    B = x0.shape[0]
    ext = {"meal": meal, "tVNS": tvns}
    with torch.no_grad():
        for _ in range(2):
            y = m(x0, t, ext)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(5):
            y = m(x0, t, ext)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5
    fwd = {"call": "HybridODENN.forward under torch.no_grad()", "patients": B, "grid_points": int(t.shape[-1]), "ms_wall": wall * 1e3,
           "ms_events": e0.elapsed_time(e1) / 5, "value": B / wall, "unit": "patient-trajectories/s",
           "over_headline_kernel_time": wall * 1e3 / headline_ms}
    steps = []
    for Bs, Ts, n in ((32, 61, 30), (B, int(t.shape[-1]), 4)):
        xs, ts_, ms_, vs_ = (v.to(dev) for v in synth_cohort(Bs, 1000))
        ts_, ms_, vs_ = ts_[:Ts].contiguous(), ms_[:, :Ts].contiguous(), vs_[:, :Ts].contiguous()
        mm = class_model(dev)
        with torch.no_grad():
            obs = mm(xs, ts_, {"meal": ms_, "tVNS": vs_}) + 0.1 * torch.randn(Bs, Ts, 6, device=dev, generator=torch.Generator(dev).manual_seed(5))
        batch = {"initial_state": xs, "observations": obs, "time_points": ts_, "external_inputs": {"meal": ms_, "tVNS": vs_}}
        opt = torch.optim.Adam(mm.parameters(), lr=1e-3)

        def step():
            opt.zero_grad()
            loss = mm.loss(batch, 1.0, 0.01)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(mm.parameters(), 5.0)
            opt.step()
            return loss
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            loss = step()
        e1.record()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        steps.append({"windows": Bs, "grid_points": Ts, "ms_wall": wall / n * 1e3, "ms_host_issue": host / n * 1e3,
                      "ms_events": e0.elapsed_time(e1) / n, "value": Bs * n / wall, "unit": "windows/s", "loss": float(loss)})
    return {"forward": fwd, "train_step": {"call": "zero_grad -> HybridODENN.loss(batch, 1.0, 0.01) (data + physics + L2) -> backward -> "
                                                   "clip_grad_norm_(5.0) -> torch.optim.Adam.step (train/train_hybrid.py:237-261)", "cases": steps}}


def train_problem(dev, x0, t, meal, tvns, ode, nn_teacher, rank):
    """Observations of this rank's shard (teacher trajectories + N(0, 0.1^2) noise, seeded per rank) and the common student
    initialisation of the training leg -- also used by tests/test_bench_contract_gpu.py to rebuild the 2-rank step in one process."""
    import hode
    with torch.no_grad():
        y = hode.solve_fwd(x0, t, meal, tvns, None, ode, nn_teacher, H, L, rtol=1e-6, atol=1e-8).y
        obs = y + 0.1 * torch.randn(x0.shape[0], T, 6, device=dev, generator=torch.Generator(dev).manual_seed(7 + rank))
    student = (nn_teacher * (1 + 0.05 * torch.randn(nn_teacher.shape, generator=torch.Generator().manual_seed(99)).to(dev))).contiguous()
    return obs, student


def cpu_baseline_host(c_sample, ref_sample=1024, ref_one_thread=128):
    """The reference's CPU path beside the GPU number, on this box's host cores:
      * reference-style (the headline of this object): per-patient scipy.integrate.solve_ivp(method='RK45', 1e-6 / 1e-8)
        loop with an fp32 torch-CPU RHS -- what reference models/hybrid_ode_nn.py:184-256 does, restated over this repo's own
        ODECore / NNResidual modules (oracle/reference_style.py; the reference's Python cannot travel to the GPU box).
        One thread, and one worker process per core;
      * c_port: the C oracle (same grid-broken DP5(4) as the kernel, oracle/hode_oracle.c), one thread per core."""
    from oracle import oracle as O
    from oracle import reference_style as RS
    O.lib()
    ncores = max(1, min(os.cpu_count() or 1, 16))     # the GPU box gives one GPU a 16-core share
    x0, t, meal, tvns = (v.numpy() for v in synth_cohort(max(c_sample, ref_sample), 12345))
    nn, ode = synth_weights().numpy(), ODE_DEFAULT.numpy()

    # reference-style, one thread
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    n1 = min(ref_one_thread, ref_sample)
    RS.solve(x0[:2], t, {"meal": meal[:2], "tVNS": tvns[:2]}, nn, ode, H, L, solver="rk45")          # warm-up
    cnt = {}
    t0 = time.perf_counter()
    y_ref = RS.solve(x0[:n1], t, {"meal": meal[:n1], "tVNS": tvns[:n1]}, nn, ode, H, L, solver="rk45", count=cnt)
    dt1 = time.perf_counter() - t0
    torch.set_num_threads(torch_threads)
    # reference-style, all cores (worker start-up and imports are outside the timed region)
    _, nfev_all, procs, dt_all = RS.solve_all_cores(x0[:ref_sample], t, {"meal": meal[:ref_sample], "tVNS": tvns[:ref_sample]}, nn, ode,
                                                    H, L, solver="rk45", procs=ncores, timed=True)

    def work(sl):
        s = O.solve(x0[sl], t, meal[sl], tvns[sl], None, ode, nn, H, L, rtol=1e-6, atol=1e-8, dtype=np.float32)
        return int(s.nsteps.sum())

    chunks = [slice(i, min(i + 16, c_sample)) for i in range(0, c_sample, 16)]
    work(slice(0, 4))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncores) as ex:        # ctypes releases the GIL: real threads
        list(ex.map(work, chunks))
    dtc = time.perf_counter() - t0
    out = {"value": ref_sample / dt_all, "unit": "patient-trajectories/s", "cores": procs, "kind": "port",
           "path": "reference-style: per-patient scipy solve_ivp(RK45, rtol 1e-6, atol 1e-8) loop, fp32 torch-CPU RHS "
                   "(reference models/hybrid_ode_nn.py:184-256 restated in oracle/reference_style.py)",
           "sample": f"{ref_sample} trajectories of the same synthetic cohort (T=241) over {procs} worker processes, {dt_all:.1f} s wall",
           "per_core": ref_sample / dt_all / procs,
           "one_thread": {"value": n1 / dt1, "unit": "patient-trajectories/s", "cores": 1,
                          "sample": f"{n1} trajectories, {dt1:.1f} s", "rhs_calls_per_trajectory": cnt["nfev"] / n1},
           "c_port": {"value": c_sample / dtc, "unit": "patient-trajectories/s", "cores": ncores, "kind": "port",
                      "per_core": c_sample / dtc / ncores,
                      "sample": f"{c_sample} trajectories, C oracle (the kernel's own grid-broken DP5(4), fp32), {ncores} threads, {dtc:.1f} s wall"}}
    # the Sobol study's per-solve cost in the reference's style on this box: model.forward(solver='dopri5') = SciPy DOP853 at
    # 1e-6 / 1e-8 (models/hybrid_ode_nn.py:174-181), one patient per call, 24 of the 16 384 sets on one thread
    sx0, st_, smeal, stv = (v.numpy() for v in sobol_inputs())
    ssets = saltelli_sets(1024)[:24]
    torch.set_num_threads(1)
    t0 = time.perf_counter()
    for row in ssets:
        o2 = ode.copy()
        for name, val in zip(SOBOL_NAMES, row):
            o2[ODE_NAMES.index(name)] = np.float32(val)
        RS.solve(sx0[None], st_, {"meal": smeal, "tVNS": stv}, nn, o2, H, L, solver="dopri5")
    dts = time.perf_counter() - t0
    torch.set_num_threads(torch_threads)
    out["sobol_reference_style"] = {"seconds_per_set_one_thread": dts / len(ssets), "seconds_for_16384_one_thread": dts / len(ssets) * 16384,
                                    "sample": f"{len(ssets)} of the 16 384 sets, solve_ivp(DOP853, 1e-6 / 1e-8) per set, fp32 torch-CPU RHS, one thread"}
    out["_host"] = (x0, t, meal, tvns, nn, ode, n1, y_ref)
    return out


def cpu_baseline_gpu(out):
    """The half of cpu_baseline that needs the GPU: parity of the HIP path with the checkers on the benchmark cohort."""
    from oracle import oracle as O
    from oracle import reference_style as RS
    x0, t, meal, tvns, nn, ode, n1, y_ref = out.pop("_host")
    out["parity_check"] = parity_check(O, x0[:16], t, meal[:16], tvns[:16], nn, ode)
    # what the reference itself would have returned for the first trajectories (default tolerances: ~1e-2 with meals)
    import hode
    dev = torch.device("cuda", torch.cuda.current_device())
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)          # noqa: E731
    yk = hode.solve_fwd(f32(x0[:n1]), f32(t), f32(meal[:n1]), f32(tvns[:n1]), None, f32(ode), f32(nn), H, L).y.cpu().numpy()
    # the SciPy-driven path at CONVERGED tolerances (the parity target) on 4 trajectories of this cohort
    y_tight = RS.solve(x0[:4], t, {"meal": meal[:4], "tVNS": tvns[:4]}, nn, ode, H, L, solver="rk45", rtol=1e-10, atol=1e-12)
    out["parity_check"]["vs_reference_style_rk45_tight"] = {
        "forward_rel_err": float(np.max(np.abs(yk[:4] - y_tight) / (np.abs(y_tight) + 1e-3))), "bar": 1e-3,
        "sample": "4 trajectories; HIP fp32 at the benchmark tolerances vs solve_ivp(RK45, rtol 1e-10, atol 1e-12) with the torch-CPU RHS"}
    out["parity_check"]["reference_own_error_at_its_default_tolerances"] = {
        "value": float(np.max(np.abs(yk - y_ref) / (np.abs(y_ref) + 1e-3))),
        "note": "solve_ivp(RK45, 1e-6 / 1e-8) vs the converged solution on this cohort: SciPy's steps straddle the kinks of the "
                "piecewise-linear meal forcing (SURVEY F6); it is the reference's discretisation error, not a parity gap"}
    return out


def parity_check(O, x0, t, meal, tvns, nn, ode):
    """The metric's second half ("adjoint grad rel-err"): 16 trajectories of the benchmark cohort, HIP fp32 at the benchmark
    tolerances against the oracle in fp64 at 1e-10 / 1e-12 (checker only)."""
    import hode
    dev = torch.device("cuda", torch.cuda.current_device())
    f32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)          # noqa: E731
    sol = hode.solve_fwd(f32(x0), f32(t), f32(meal), f32(tvns), None, f32(ode), f32(nn), H, L, want_tape=True)
    ref = O.solve(x0, t, meal, tvns, None, ode, nn, H, L, rtol=1e-10, atol=1e-12, dtype=np.float64, want_tape=True)
    c = np.random.default_rng(0).standard_normal(ref.y.shape)
    _, gnn, _ = hode.solve_bwd(sol, f32(c))
    _, rnn, _ = O.solve_bwd(ref, c)
    return {"forward_rel_err": float(np.max(np.abs(sol.y.cpu().numpy() - ref.y) / (np.abs(ref.y) + 1e-3))),
            "adjoint_grad_rel_err": float(np.linalg.norm(gnn.cpu().numpy() - rnn) / np.linalg.norm(rnn)),
            "bars": {"forward": 1e-3, "adjoint": 1e-4}, "sample": "16 trajectories of the benchmark cohort, T=241"}


def vi_step(dev, patients, samples=16):
    """BASELINE config 5 at its per-GPU size: one ELBO evaluation (S Monte-Carlo parameter draws x patients, KL and
    likelihood in fp64) + reparameterised gradient through the adjoint + Adam on the variational parameters."""
    from models import HybridODENN
    x0, t, meal, tvns = (v.to(dev) for v in synth_cohort(patients, 777))
    prior = {f"ode_{n}": {"mean": v, "std": 0.02 * v} for n, v in
             [("a_GI", 0.0104), ("k_I", 0.025), ("rho", 0.003), ("E_max", 0.1), ("EC_50", 50.0), ("V_max", 9.0), ("K_m", 7.0), ("k_L", 0.02)]}
    torch.manual_seed(0)
    m = HybridODENN(use_variational=True, prior_params=prior, device=dev)
    teacher = synth_weights(0)
    with torch.no_grad():
        off = 0
        for name, p in m.nn_residual.named_parameters():
            m.variational_params.means["nn_" + name.replace(".", "_")].copy_(teacher[off:off + p.numel()].reshape(p.shape))
            off += p.numel()
        for n, p in m.variational_params.log_stds.items():
            p.fill_(-6.0 if n.startswith("nn_") else float(np.log(0.02 * prior[n]["mean"])))
        obs = m.forward_with_params({k: v.detach() for k, v in m.variational_params.means.items()}, x0, t, {"meal": meal, "tVNS": tvns})
    batch = {"initial_state": x0, "observations": obs + 0.1 * torch.randn_like(obs), "time_points": t,
             "external_inputs": {"meal": meal, "tVNS": tvns}}
    opt = torch.optim.Adam(m.variational_params.parameters(), lr=1e-3)
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        e = m.elbo(batch, n_samples=samples, noise_sigma=0.1)
        (-e).backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    n = patients * samples
    return {"workload": f"{patients} patients x {samples} VI draws (BASELINE config 5 per GPU), T={T}, fp32 solve, fp64 KL / likelihood",
            "value": n / min(times[1:]), "unit": "patient-trajectories/s", "s_per_step": min(times[1:]), "trajectories": n,
            "trajectories_ok": int((m.last_solve_info["status"] == 0).sum()), "elbo": float(e),
            "peak_mem_gib": torch.cuda.max_memory_allocated(dev) / 2 ** 30}


def generic_path(dev):
    """The reference's largest configuration, nn_hidden 128 / nn_layers 5 (configs/ablation_no_physics.yaml:11-12), through the generic
    kernels (csrc/hode_generic.hip): forward with tape + adjoint at the reference's batch (32 windows x 61
    points) and at 1 024 x 61.  HIP events on the launch stream; secondary numbers, never the headline."""
    import hode
    Hg, Lg, Tg = 128, 5, 61
    g = torch.Generator().manual_seed(Hg + Lg)
    P = hode.n_params(Hg, Lg)
    nn = torch.randn(P, generator=g) * (0.5 * (2.0 / (2 * Hg)) ** 0.5)
    nn[-(6 * Hg + 6):] *= 0.1
    nn, ode = nn.to(dev), ODE_DEFAULT.to(dev)
    rows = []
    for Bg in (32, 1024):
        x0, t, meal, tv = (v.to(dev) for v in synth_cohort(Bg, 5))
        t, meal, tv = t[:Tg].contiguous(), meal[:, :Tg].contiguous(), tv[:, :Tg].contiguous()
        st = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, Hg, Lg, want_tape=True)
        gy = torch.randn(st.y.shape, device=dev, generator=torch.Generator(dev).manual_seed(1)) / st.y.numel()

        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        ms_f = timed(lambda: hode.solve_fwd(x0, t, meal, tv, None, ode, nn, Hg, Lg, want_tape=True, tape=st.tape))
        ms_b = timed(lambda: hode.solve_bwd(st, gy))
        flop_rhs = 2 * (9 * Hg + (Lg - 1) * Hg * Hg + 6 * Hg) + 80          # 134 992 for 128 x 5
        f_fwd = float(st.nfev.double().sum()) * flop_rhs
        f_adj = float(st.nsteps.double().sum()) * 6 * 2 * flop_rhs
        rows.append({"patients": Bg, "grid_points": Tg, "forward_with_tape_ms": ms_f, "adjoint_ms": ms_b,
                     "trajectories_per_s_train": Bg / (ms_f + ms_b) * 1e3, "trajectories_ok": int((st.status == 0).sum()),
                     "roofline": {"bound": "valu_fp32", "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "flop_per_rhs": flop_rhs,
                                  "forward_with_tape": {"achieved": f_fwd / (ms_f * 1e-3) / 1e12, "frac": f_fwd / (ms_f * 1e-3) / 1e12 / PEAK_FP32_TFLOPS},
                                  "adjoint": {"achieved": f_adj / (ms_b * 1e-3) / 1e12, "frac": f_adj / (ms_b * 1e-3) / 1e12 / PEAK_FP32_TFLOPS}}})
    return {"workload": "MLP 9 -> 128 x 5 -> 6 (67 k parameters), DP5(4), fp32, forward with tape + adjoint",
            "kernels": "solve_fwd_generic_kernel / solve_bwd_generic_kernel: teams of 4-8 waves per trajectory", "cases": rows}


def data_side(dev, B, cpu=True):
    """SURVEY 8f-3 leg: generate a B-subject 4GI cohort (5 h at 5 min, 2 meals, 10 % noise: the reference's
    data/generate4GI.py __main__ configuration) on the device, then cut and z-score the windows (31 / 15)."""
    import hode
    from hode.datagen import grid_points
    T = grid_points(5, 5)
    g = torch.Generator(device=dev).manual_seed(0)
    base = torch.tensor([7.0, 50.0, 10.0, 25.0, 20.0], dtype=torch.float64, device=dev)
    cv = torch.tensor([0.1, 0.15, 0.15, 0.15, 0.15], dtype=torch.float64, device=dev)
    bsl = base * (1 + cv * torch.randn(B, 5, dtype=torch.float64, device=dev, generator=g))
    z = torch.randn(T, 5, B, dtype=torch.float64, device=dev, generator=g)      # the kernel's layout

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps, r
    ms_gen, (table, status) = timed(lambda: hode.capi.fourgi_generate(bsl, T, 5.0, [0.5, 2.5], [75.0, 50.0], z_tcb=z, noise_cv=0.1))
    S, stride = 31, 15
    n_win = (T - S) // stride + 1
    row0 = (torch.arange(B, device=dev)[:, None] * T + torch.arange(n_win, device=dev)[None, :] * stride).reshape(-1)
    cols = dict(time=2, glucose=3, insulin=4, glucagon=6, glp1=5, meal=8)
    ms_win, _ = timed(lambda: hode.capi.fourgi_windows(table, cols, 60.0, row0, S, True, check_bounds=False))
    alg = table.numel() * 8 + row0.numel() * S * 9 * 4          # table read once + fp32 batches written once
    out = {"workload": f"{B} subjects x {T} grid points (5 h at 5 min), T2DM, 2 meals, fp64 DP5(4) rtol 1e-10; windows {S}/{stride}",
           "generate": {"ms": ms_gen, "subjects_per_s": B / ms_gen * 1e3, "failed": int((status != 0).sum()),
                        "kernel": "fourgi_generate_kernel (one subject per lane)", "bound": "fp64 VALU latency (3 pow per RHS), 1 wave/SIMD",
                        "algorithmic_bytes": B * T * (72 + 40)},
           "windows": {"ms": ms_win, "windows": int(row0.numel()), "algorithmic_bytes": alg,
                       "roofline": {"bound": "hbm", "achieved": alg / ms_win / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": alg / ms_win / 1e6 / PEAK_HBM_GBS}}}
    gen = out["generate"]
    traffic, src = measured_traffic("fourgi_generate_hbm_bytes_per_launch")
    gen["roofline_hbm"] = {"bound": "hbm", "achieved": gen["algorithmic_bytes"] / ms_gen / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": gen["algorithmic_bytes"] / ms_gen / 1e6 / PEAK_HBM_GBS, "traffic": traffic,
                           "traffic_source": None if traffic is None else f"{src}; this leg's workload: {B} subjects x {T} grid points, fp64"}
    if cpu:
        from oracle import fourgi
        n = 512
        t0 = time.perf_counter()
        _, _, steps = fourgi.simulate(bsl[:n].cpu().numpy(), T, 5.0, [0.5, 2.5], [75.0, 50.0])
        dt = time.perf_counter() - t0
        gen["cpu_baseline"] = {"value": n / dt, "unit": "subjects/s", "cores": 1, "kind": "port",
                               "sample": f"{n} subjects of the same cohort, C oracle (DP5(4) at the same tolerances)"}
        # fp64 vector roofline of the generator: accepted steps per subject from the oracle (same algorithm) on that sample
        flops = B * (steps / n) * FLOP_PER_4GI_STEP
        gen["roofline"] = {"bound": "valu_fp64", "achieved": flops / (ms_gen * 1e-3) / 1e12, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
                           "frac": flops / (ms_gen * 1e-3) / 1e12 / PEAK_FP64_TFLOPS, "accepted_steps_per_subject": steps / n,
                           "algorithmic_flops_per_step": FLOP_PER_4GI_STEP,
                           "note": "one subject per lane, 1 wave/SIMD (56 live fp64 stage derivatives): latency-bound, far from either roof"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patients-per-gpu", type=int, default=0, help="default: 4 096 (BASELINE config[1] per GPU); with --gpus 8 "
                    "8 192 = BASELINE config[3], the 65 536-patient cohort over the node")
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=32768, help="trajectories of the C-oracle leg of cpu_baseline")
    ap.add_argument("--ref-sample", type=int, default=1024, help="trajectories of the reference-style (solve_ivp loop) leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-data-side", action="store_true")
    ap.add_argument("--no-vi", action="store_true")
    ap.add_argument("--no-generic", action="store_true", help="skip the generic-path (128 x 5 network) block")
    ap.add_argument("--no-sobol", action="store_true", help="skip the Sobol-study leg (16 384 parameter sets, plots/plot_all.py:139-196)")
    ap.add_argument("--no-class-path", action="store_true", help="skip the class-surface leg (HybridODENN.forward / loss / optimiser step)")
    ap.add_argument("--vi-patients", type=int, default=8192, help="patients of the VI leg (BASELINE config 5: 8 192 per GPU x 16 draws)")
    ap.add_argument("--cohort", type=int, default=65536, help="subjects of the data-side leg (4GI generator + windows)")
    ap.add_argument("--no-zscore", action="store_true", help="skip the z-scored-regime leg (profiling: the headline kernel's "
                    "launches are then all the benchmark workload, so rocprofv3's average IS ms_per_step)")
    ap.add_argument("--no-build", action="store_true", help="never spawn make (profiler runs); missing / stale artefacts are errors")
    ap.add_argument("--total-patients", type=int, default=0, help="N > 1: shard this many patients over the ranks with "
                    "hode.train.shard_bounds (uneven shards allowed) instead of --patients-per-gpu each")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal: ranks may share one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    ensure_built(may_build=not (args.no_build or os.environ.get("HODE_NO_BUILD")))   # before any GPU / collective call
    if args.patients_per_gpu <= 0:
        args.patients_per_gpu = 8192 if args.gpus == 8 else 4096
    # the host-only half of cpu_baseline (reference-style solve_ivp loops: one thread, then one spawned worker process per core;
    # the C oracle on threads) runs BEFORE the first call that initialises the GPU: no process is started from a GPU-initialised one
    host = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        host = cpu_baseline_host(args.cpu_sample, args.ref_sample)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")                     # rehearsal only (collectives staged through the host)

    import hode
    hode.load()
    if args.total_patients > 0:
        # one cohort sharded contiguously over the ranks (hode.train.shard_bounds: the first N % world ranks hold one more)
        B_total = args.total_patients
        lo, hi = hode.train.shard_bounds(B_total, rank, world)
        B = hi - lo
        x0, t, meal, tvns = (v.to(dev) for v in synth_cohort(B_total, 1000))
        x0, meal, tvns = x0[lo:hi].contiguous(), meal[lo:hi].contiguous(), tvns[lo:hi].contiguous()
    else:
        B = args.patients_per_gpu
        B_total = world * B
        x0, t, meal, tvns = (v.to(dev) for v in synth_cohort(B, 1000 + rank))
    nn_teacher = synth_weights(0).to(dev)
    ode = ODE_DEFAULT.to(dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def allreduce_max(tensor):
        if world > 1:
            if args.backend == "nccl":
                dist.all_reduce(tensor, op=dist.ReduceOp.MAX)
            else:
                c = tensor.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX)
                tensor.copy_(c)

    # ------------------------------------------------------------------ headline: forward solve
    def fwd_step():
        return hode.solve_fwd(x0, t, meal, tvns, None, ode, nn_teacher, H, L, rtol=1e-6, atol=1e-8)

    for _ in range(args.warmup):
        sol = fwd_step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()                      # HIP events on torch's current stream == the stream the kernel runs on
    for _ in range(args.steps):
        sol = fwd_step()
    e1.record()
    barrier()
    wall = time.perf_counter() - t0
    kern_ms = e0.elapsed_time(e1) / args.steps
    tm = torch.tensor([wall], dtype=torch.float64, device=dev)
    allreduce_max(tm)
    wall = float(tm)
    nfev = float(sol.nfev.double().sum())
    nsteps = float(sol.nsteps.double().sum())
    ok = int((sol.status == 0).sum())
    value = B_total * args.steps / wall

    # ------------------------------------------------------------------ secondary: z-scored regime
    # what GlucoseDataset(normalize=True) actually feeds (train/train_hybrid.py:139): x0 ~ N(0,1)^6
    zs = None
    if rank == 0 and not args.no_zscore:
        xz = torch.randn(B, 6, generator=torch.Generator().manual_seed(4242)).to(dev)
        zsol = hode.solve_fwd(xz, t, meal, tvns, None, ode, nn_teacher, H, L, rtol=1e-6, atol=1e-8)
        torch.cuda.synchronize()
        z0, z1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        z0.record()
        for _ in range(5):
            zsol = hode.solve_fwd(xz, t, meal, tvns, None, ode, nn_teacher, H, L, rtol=1e-6, atol=1e-8)
        z1.record()
        torch.cuda.synchronize()
        zms = z0.elapsed_time(z1) / 5
        zs = {"value": B / zms * 1e3, "unit": "patient-trajectories/s (this rank)", "ms_per_step": zms,
              "mean_steps": float(zsol.nsteps.float().mean()), "mean_nfev": float(zsol.nfev.float().mean()),
              "trajectories_ok": int((zsol.status == 0).sum()), "x0": "N(0,1)^6 (z-scored states)",
              "note": "status-2 trajectories run into the pole of G/(K_m+G) in finite time (step underflow); the oracle "
                      "and SciPy stop on the same trajectories -- reported per trajectory, never raised"}
    barrier()

    # ------------------------------------------------------------------ secondary: training step
    train = None
    if not args.no_train:
        obs, student = train_problem(dev, x0, t, meal, tvns, ode, nn_teacher, rank)
        state = hode.train.TrainState(student.clone())
        n_glob = B_total * T * 6

        def compute(p):
            ls, gnn, gode, _ = hode.train.hip_loss_and_grads(p, ode, x0, t, meal, tvns, obs, H, L, n_glob, state=state)
            return ls, gnn, gode, B * T * 6

        ts_kw = {} if args.backend == "nccl" else {"host_staged": True}
        losses = [float(hode.train.train_step(state, compute, **ts_kw)) for _ in range(2)]       # warm-up
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            loss = hode.train.train_step(state, compute, **ts_kw)
        barrier()
        tw = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        allreduce_max(tw)
        losses.append(float(loss))
        # the two kernels of the step, timed one by one with HIP events on the launch stream (rocprofv3 agrees: profiles/)
        def kernel_ms(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            for _ in range(reps):
                r = fn()
            a1.record()
            torch.cuda.synchronize()
            return a0.elapsed_time(a1) / reps, r
        ms_ft, solt = kernel_ms(lambda: hode.solve_fwd(x0, t, meal, tvns, None, ode, state.p, H, L, want_tape=True, tape=state.tape))
        gy_probe = torch.randn(B, T, 6, device=dev, generator=torch.Generator(dev).manual_seed(3)) / (B * T * 6)
        ms_adj, _ = kernel_ms(lambda: hode.solve_bwd(solt, gy_probe))
        st_steps = float(solt.nsteps.double().sum())
        st_nfev = float(solt.nfev.double().sum())
        adj_flops = st_steps * 6 * FLOP_PER_ADJ_STAGE
        tape_bytes = st_steps * (6 * STAGE_REC_BYTES + 36) + B * (T * 6 * 4 + 2 * T * 4)      # stage tape + entries + dLoss/dy + inputs
        fwt_flops = st_nfev * FLOP_PER_RHS + st_steps * FLOP_PER_STEP_ALGEBRA
        fwt_bytes = st_steps * (6 * STAGE_REC_BYTES + 36) + B * BYTES_FWD_PER_TRAJ
        tr_adj, src_adj = measured_traffic("solve_bwd_hbm_bytes_per_launch")
        tr_fwt, _ = measured_traffic("solve_fwd_tape_hbm_bytes_per_launch")
        roof = {"adjoint": {"kernel": "solve_bwd_ws_kernel<4,2> (wave-specialised: 8 propagation + 8 accumulation waves per CU) + adj_reduce_kernel", "kernel_ms": ms_adj, "bound": "valu_fp32",
                            "achieved": adj_flops / (ms_adj * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                            "frac": adj_flops / (ms_adj * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, "algorithmic_flops_per_launch": adj_flops,
                            "flops_model": "2 x forward RHS flops per stage (dW outer products + W^T delta products; no recompute)",
                            "traffic": tr_adj, "traffic_source": src_adj,
                            "hbm": {"achieved": tape_bytes / (ms_adj * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": tape_bytes / (ms_adj * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": tape_bytes,
                                    "note": "bytes the adjoint must stream back: stage tape (activations + stage states) + step entries + dLoss/dy"}},
                "forward_with_tape": {"kernel": "solve_fwd_kernel<float,4,DP54,TAPE>", "kernel_ms": ms_ft, "bound": "valu_fp32",
                                      "achieved": fwt_flops / (ms_ft * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                      "frac": fwt_flops / (ms_ft * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, "traffic": tr_fwt,
                                      "hbm": {"achieved": fwt_bytes / (ms_ft * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                              "frac": fwt_bytes / (ms_ft * 1e-3) / 1e9 / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": fwt_bytes}}}
        # SURVEY 8(d) counts only what a training step HAS to move per trajectory (inputs, y, dLoss/dy, gx0: 7 736 + 13 520 B);
        # the stage tape is this design's memory-for-compute trade and is reported as such, not hidden in the bytes
        survey_bytes = B * (BYTES_FWD_PER_TRAJ + BYTES_BWD_PER_TRAJ)
        step_traffic = (tr_adj or 0) + (tr_fwt or 0)
        roof["step_hbm_vs_survey_8d"] = {
            "survey_algorithmic_bytes_per_step": survey_bytes, "tape_inclusive_algorithmic_bytes_per_step": tape_bytes + fwt_bytes,
            "measured_traffic_per_step": step_traffic or None,
            "measured_over_survey": (step_traffic / survey_bytes) if step_traffic else None,
            "tape_inclusive_over_survey": (tape_bytes + fwt_bytes) / survey_bytes,
            "achieved_on_survey_bytes": {"value": survey_bytes / ((ms_ft + ms_adj) * 1e-3) / 1e9, "unit": "GB/s",
                                         "frac": survey_bytes / ((ms_ft + ms_adj) * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "note": "the ratio is the price of taping the stage activations instead of recomputing them (recompute measured 27.5 ms "
                    "against 7-8 ms); the step stays VALU-bound, the tape streams at < 2 TB/s"}
        train = {"roofline": roof,
                 "metric": "patient-trajectories/s (fwd + adjoint + all-reduce + fused Adam)",
                 "value": B_total * args.train_steps / float(tw), "ms_per_step": float(tw) / args.train_steps * 1e3,
                 "steps": args.train_steps, "loss_first_last": [losses[0], losses[-1]],
                 "collective": "1 x all_reduce(sum) of 13 529 fp64 (108 KB: gradients | loss sum | element count) per step" if world > 1 else "none (1 rank)"}

    if rank == 0:
        flops = nfev * FLOP_PER_RHS + nsteps * FLOP_PER_STEP_ALGEBRA
        tflops = flops / (kern_ms * 1e-3) / 1e12
        gbs = B * BYTES_FWD_PER_TRAJ / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic("solve_fwd_hbm_bytes_per_launch")
        out = {
            "metric": "patient-trajectories/s (6-state, 240-step dopri5)", "value": value,
            "unit": "patient-trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE config[3]: {B_total}-patient 4GI-style synthetic cohort sharded over {world} GPUs ({B_total // world} per GPU)"
                                    if world > 1 and B_total // world == 8192 else
                                    f"BASELINE config[1]: {B_total // world}-patient 4GI-style synthetic cohort per GPU") +
                                   ", DP5(4) adaptive (rtol 1e-6, atol 1e-8), ODE + 4x64 MLP residual, fp32, forward solve, T=241"
                                   + ("; train_step = config[3]'s step: adjoint + ONE RCCL all-reduce of the gradients + fused Adam" if world > 1 else ""),
                       "patients_per_gpu": B_total / world, "patients_total": B_total, "grid_points": T, "parallelism": f"patients sharded x{world}, no data-path collective",
                       "trajectories_ok": ok, "mean_steps": nsteps / B, "mean_nfev": nfev / B},
            "roofline": {"bound": "valu_fp32", "bound_detail": "fp32 vector FMA issue (MFMA deliberately unused, north_star); peak = "
                         "fp32 vector peak 157.3 TFLOP/s (= the f32 MFMA dense peak)",
                         "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_TFLOPS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "solve_fwd_kernel<float,4,DP54>", "kernel_ms": kern_ms,
                         "algorithmic_flops_per_launch": flops,
                         "hbm": {"achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                 "algorithmic_bytes_per_launch": B * BYTES_FWD_PER_TRAJ}},
        }
        if zs is not None:
            out["zscore_regime"] = zs
        if train is not None:
            out["train_step"] = train
        if world == 1 and not args.no_vi:
            out["vi_step"] = vi_step(dev, args.vi_patients)
        if world == 1 and not args.no_generic:
            out["generic_path"] = generic_path(dev)
        if world == 1 and not args.no_data_side:
            out["data_side"] = data_side(dev, args.cohort, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_sobol:
            out["sobol"] = sobol_leg(dev, host=host)
        if world == 1 and not args.no_class_path:
            out["class_path"] = class_path(dev, x0, t, meal, tvns, kern_ms)
        if host is not None:
            host.pop("sobol_reference_style", None)
            out["cpu_baseline"] = cpu_baseline_gpu(host)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
