"""Quick forward-throughput probe (development aid; the driver contract lives in bench.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd"))
import numpy as np, torch
import hode

def cohort(B, T=241, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    base = torch.tensor([5., 60., 80., 10., 0., 1.])
    x0 = base * (1 + 0.05 * torch.randn(B, 6, generator=g))
    t = torch.arange(T, dtype=torch.float64) * (5.0 / 60.0)
    meal = torch.zeros(B, T)
    idx = torch.stack([torch.randperm(229, generator=g)[:4] + 6 for _ in range(B)])
    meal.scatter_(1, idx, 1.0)
    tv = torch.zeros(B, T)
    return [v.to("cuda", dtype) for v in (x0, t, meal, tv)]

w = np.load(os.path.join(ROOT, "tests/golden/g0_weights_h64_l4.npz"))
for B in [int(a) for a in sys.argv[1:]] or [4096]:
    x0, t, meal, tv = cohort(B)
    nn = torch.tensor(w["nn_flat"], device="cuda"); ode = torch.tensor(w["ode"], device="cuda")
    for it in range(3):
        s = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for it in range(n):
        s = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"B={B}: {ms:.3f} ms/solve  {B/ms*1e3:,.0f} traj/s  steps mean {s.nsteps.float().mean():.1f} max {int(s.nsteps.max())} nfev mean {s.nfev.float().mean():.0f} status max {int(s.status.max())}")
