#!/usr/bin/env python3
"""The z-scored regime of bench.py (x0 ~ N(0,1)^6, what GlucoseDataset(normalize=True) feeds) as its own program, so that a
kernel trace holds ONLY these launches of solve_fwd_kernel (tools/profile_gpu.sh -> profiles/<tag>_kernel_stats_zscore.csv).
No build, no child processes: safe under rocprofv3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
import hode  # noqa: E402

dev = torch.device("cuda")
B = 4096
_, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
nn, ode = bench.synth_weights(0).to(dev), bench.ODE_DEFAULT.to(dev)
xz = torch.randn(B, 6, generator=torch.Generator().manual_seed(4242)).to(dev)
for _ in range(7):
    sol = hode.solve_fwd(xz, t, meal, tvns, None, ode, nn, bench.H, bench.L, rtol=1e-6, atol=1e-8)
torch.cuda.synchronize()
print("z-scored launches done; ok", int((sol.status == 0).sum()), "of", B)
