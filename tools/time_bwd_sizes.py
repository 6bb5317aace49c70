#!/usr/bin/env python3
"""Adjoint time against batch size (launch geometry check)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd"))
import bench  # noqa: E402
import hode  # noqa: E402

dev = torch.device("cuda")
nn, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
for B in (256, 384, 512, 512, 640, 768, 1024, 1536, 4096):
    x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
    gy = torch.randn(B, bench.T, 6, device=dev)
    sol = hode.solve_fwd(x0, t, meal, tvns, None, ode, nn, 64, 4, want_tape=True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hode.solve_bwd(sol, gy)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"B={B}: bwd {1e3 * (t1 - t0):.2f} ms  ({B / (t1 - t0) / 1e3:.0f} k traj/s)")
