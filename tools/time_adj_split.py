"""Per-kernel time of the split adjoint (development aid): run under rocprofv3 --kernel-trace --stats, or read the event times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")); sys.path.insert(0, ROOT)
import torch, hode, bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
x0, t, meal, tv = (v.to(dev) for v in bench.synth_cohort(B, 1000))
nn = bench.synth_weights(0).to(dev); ode = bench.ODE_DEFAULT.to(dev)
sol = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4, want_tape=True)
gy = torch.randn_like(sol.y) / sol.y.numel()
for _ in range(3): hode.solve_bwd(sol, gy)
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record(); hode.solve_bwd(sol, gy, want_gnn=False); e1.record(); hode.solve_bwd(sol, gy); e2.record(); torch.cuda.synchronize()
print(f"B={B}: propagation only {e0.elapsed_time(e1):.2f} ms | propagation + accumulation {e1.elapsed_time(e2):.2f} ms")
