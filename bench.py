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
BYTES_TRAIN_PER_TRAJ = 21_300
PEAK_FP32_TFLOPS = 157.3                                      # MI355X_MICROARCH.md: fp32 vector == f32-MFMA dense peak
PEAK_HBM_GBS = 8000.0


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


def cpu_baseline(sample, seconds_budget=20.0):
    """Oracle (C port of the same grid-broken DP5(4), oracle/hode_oracle.c) on the host cores."""
    from oracle import oracle as O
    O.lib()
    ncores = max(1, min(os.cpu_count() or 1, 16))     # the GPU box gives one GPU a 16-core share
    x0, t, meal, tvns = (v.numpy() for v in synth_cohort(sample, 12345))
    nn, ode = synth_weights().numpy(), ODE_DEFAULT.numpy()

    def work(sl):
        s = O.solve(x0[sl], t, meal[sl], tvns[sl], None, ode, nn, H, L, rtol=1e-6, atol=1e-8, dtype=np.float32)
        return int(s.nsteps.sum())

    chunks = [slice(i, min(i + 16, sample)) for i in range(0, sample, 16)]
    work(slice(0, 4))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncores) as ex:        # ctypes releases the GIL: real threads
        list(ex.map(work, chunks))
    dt = time.perf_counter() - t0
    out = {"value": sample / dt, "unit": "patient-trajectories/s", "cores": ncores, "kind": "port",
           "sample": f"{sample} trajectories of the same synthetic cohort (T=241, fp32, rtol 1e-6/atol 1e-8), "
                     f"C oracle, {ncores} threads, {dt:.1f} s wall",
           "per_core": sample / dt / ncores}
    out["parity_check"] = parity_check(O, x0[:16], t, meal[:16], tvns[:16], nn, ode)
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
                        "kernel": "fourgi_generate_kernel (one subject per lane)", "bound": "fp64 VALU latency (3 pow per RHS), 1 wave/SIMD"},
           "windows": {"ms": ms_win, "windows": int(row0.numel()), "algorithmic_bytes": alg,
                       "roofline": {"bound": "hbm", "achieved": alg / ms_win / 1e6, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                    "frac": alg / ms_win / 1e6 / PEAK_HBM_GBS}}}
    if cpu:
        from oracle import fourgi
        n = 512
        t0 = time.perf_counter()
        fourgi.simulate(bsl[:n].cpu().numpy(), T, 5.0, [0.5, 2.5], [75.0, 50.0])
        dt = time.perf_counter() - t0
        out["generate"]["cpu_baseline"] = {"value": n / dt, "unit": "subjects/s", "cores": 1, "kind": "port",
                                           "sample": f"{n} subjects of the same cohort, C oracle (DP5(4) at the same tolerances)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--patients-per-gpu", type=int, default=4096)
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true")
    ap.add_argument("--no-data-side", action="store_true")
    ap.add_argument("--no-vi", action="store_true")
    ap.add_argument("--vi-patients", type=int, default=8192, help="patients of the VI leg (BASELINE config 5: 8 192 per GPU x 16 draws)")
    ap.add_argument("--cohort", type=int, default=65536, help="subjects of the data-side leg (4GI generator + windows)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal: ranks may share one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if args.backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")                     # rehearsal only (collectives staged through the host)

    if not os.path.exists(os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd", "hode", "libhode.so")):
        # fresh checkout (built artefacts are git-ignored): rank 0 of the node builds, the others wait
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        if world > 1:
            dist.barrier()
    import hode
    hode.load()
    B = args.patients_per_gpu
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
    value = world * B * args.steps / wall

    # ------------------------------------------------------------------ secondary: z-scored regime
    # what GlucoseDataset(normalize=True) actually feeds (train/train_hybrid.py:139): x0 ~ N(0,1)^6
    zs = None
    if rank == 0:
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
        with torch.no_grad():
            obs = fwd_step().y + 0.1 * torch.randn(B, T, 6, device=dev, generator=torch.Generator(dev).manual_seed(7 + rank))
        student = (nn_teacher * (1 + 0.05 * torch.randn(nn_teacher.shape, generator=torch.Generator().manual_seed(99)).to(dev))).contiguous()
        state = hode.train.TrainState(student.clone())
        n_glob = world * B * T * 6

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
        train = {"metric": "patient-trajectories/s (fwd + adjoint + all-reduce + fused Adam)",
                 "value": world * B * args.train_steps / float(tw), "ms_per_step": float(tw) / args.train_steps * 1e3,
                 "steps": args.train_steps, "loss_first_last": [losses[0], losses[-1]],
                 "collective": "1 x all_reduce(sum) of 13 529 fp32 (54 KB) per step" if world > 1 else "none (1 rank)"}

    if rank == 0:
        flops = nfev * FLOP_PER_RHS + nsteps * FLOP_PER_STEP_ALGEBRA
        tflops = flops / (kern_ms * 1e-3) / 1e12
        gbs = B * BYTES_FWD_PER_TRAJ / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("solve_fwd_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "patient-trajectories/s (6-state, 240-step dopri5)", "value": value,
            "unit": "patient-trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE config[1]: 4096-patient 4GI-style synthetic cohort per GPU, DP5(4) adaptive "
                                   "(rtol 1e-6, atol 1e-8), ODE + 4x64 MLP residual, fp32, forward solve, T=241",
                       "patients_per_gpu": B, "grid_points": T, "parallelism": f"patients sharded x{world}, no data-path collective",
                       "trajectories_ok": ok, "mean_steps": nsteps / B, "mean_nfev": nfev / B},
            "roofline": {"bound": "mfma", "bound_detail": "fp32 vector FMA (MFMA deliberately unused, north_star); the f32 "
                         "MFMA dense peak equals the fp32 vector peak, 157.3 TFLOP/s",
                         "achieved": tflops, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tflops / PEAK_FP32_TFLOPS,
                         "traffic": traffic, "kernel": "solve_fwd_kernel<float,4,DP54>", "kernel_ms": kern_ms,
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
        if world == 1 and not args.no_data_side:
            out["data_side"] = data_side(dev, args.cohort, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
