#!/usr/bin/env python3
"""Does the ORDER of the trajectories in a launch matter?  4 096 trajectories on 2 048 wave slots: the right-hand-side evaluations
per trajectory (nfev) of the benchmark cohort, the makespan of list scheduling in index order / longest first against the mean
load, and the measured time of the forward with the cohort permuted longest first (same work, same results per trajectory)."""
import os, sys, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, torch, hode, numpy as np  # noqa: E402

B = 4096
dev = torch.device("cuda")
x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
nn, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
sol = hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4)
nfev = sol.nfev.cpu().numpy().astype(np.float64)
print(f"nfev: mean {nfev.mean():.0f}  min {nfev.min():.0f}  max {nfev.max():.0f}  std {nfev.std():.0f}")


def makespan(costs, slots=2048):
    h = [0.0] * slots
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


ideal = nfev.sum() / 2048
print(f"list scheduling on 2048 slots: index order {makespan(nfev) / ideal:.3f} x mean load, longest first "
      f"{makespan(np.sort(nfev)[::-1]) / ideal:.3f} x")


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


perm = torch.as_tensor(np.argsort(-nfev, kind="stable").copy(), device=dev)
xs, ms, vs = x0[perm].contiguous(), meal[perm].contiguous(), tvns[perm].contiguous()
for _ in range(3):
    a = timed(lambda: hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4))
    b = timed(lambda: hode.solve_fwd(xs, t, ms, vs, None, ode, nn, 64, 4))
    print(f"forward: index order {a:.3f} ms   longest first {b:.3f} ms")
s2 = hode.solve_fwd(xs, t, ms, vs, None, ode, nn, 64, 4)
print("same trajectories bit for bit:", bool(torch.equal(s2.y, sol.y[perm])))
