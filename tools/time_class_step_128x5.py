"""The reference's largest configuration (nn_hidden 128, nn_layers 5: configs/ablation_no_physics.yaml) at its own batch, 32 windows x 61
points, through the class surface: zero_grad -> loss -> backward -> clip -> Adam (the generic kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")); sys.path.insert(0, ROOT)
import torch, bench
from models import HybridODENN
dev = torch.device("cuda")
B, T = 32, 61
x0, t, meal, tvns = (v.to(dev) for v in bench.synth_cohort(B, 1000))
t, meal, tvns = t[:T].contiguous(), meal[:, :T].contiguous(), tvns[:, :T].contiguous()
for lam1 in (0.0, 1.0):                       # the ablation config trains without the physics term
    torch.manual_seed(0)
    m = HybridODENN(nn_hidden=128, nn_layers=5, device=dev)
    with torch.no_grad():
        m.nn_residual.network[-1].weight.normal_(0, 0.01)
        obs = m(x0, t, {"meal": meal, "tVNS": tvns}) + 0.1 * torch.randn(B, T, 6, device=dev)
    batch = {"initial_state": x0, "observations": obs, "time_points": t, "external_inputs": {"meal": meal, "tVNS": tvns}}
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    def step():
        opt.zero_grad(); loss = m.loss(batch, lam1, 0.01, use_physics_loss=lam1 > 0); loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 5.0); opt.step(); return loss
    for _ in range(3): step()
    torch.cuda.synchronize(); n = 30; t0 = time.perf_counter()
    for _ in range(n): loss = step()
    torch.cuda.synchronize()
    print(f"128 x 5, B={B} T={T}, lambda1={lam1}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per step, loss {float(loss.detach()):.4f}", flush=True)
