"""Generic-path train timings of one network shape over batch sizes:  python tools/time_generic_b.py H L T B [B ...]  (development aid;
HODE_LIB=<variant .so> times an experiment build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")); sys.path.insert(0, ROOT)
import torch, hode, bench
dev = torch.device("cuda")
H, L, T = (int(v) for v in sys.argv[1:4])
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r = f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, r
g = torch.Generator().manual_seed(H + L)
P = hode.n_params(H, L)
nn = (torch.randn(P, generator=g) * (0.5 * (2.0 / (2 * H)) ** 0.5)).to(dev)
nn[-(6 * H + 6):] *= 0.1
ode = bench.ODE_DEFAULT.to(dev)
for B in (int(v) for v in sys.argv[4:]):
    x0, t, meal, tv = (v.to(dev) for v in bench.synth_cohort(B, 5))
    t, meal, tv = t[:T].contiguous(), meal[:, :T].contiguous(), tv[:, :T].contiguous()
    ms_f, s = timeit(lambda: hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L))
    st = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L, want_tape=True)
    ms_ft, st = timeit(lambda: hode.solve_fwd(x0, t, meal, tv, None, ode, nn, H, L, want_tape=True, tape=st.tape))
    gy = torch.randn_like(st.y) / st.y.numel()
    ms_b, _ = timeit(lambda: hode.solve_bwd(st, gy))
    print(f"{os.environ.get('HODE_LIB', 'product')[-24:]:>24s} H={H} L={L} B={B:5d} T={T}: fwd {ms_f:8.2f} ms | fwd+tape {ms_ft:8.2f} | adjoint {ms_b:8.2f} | "
          f"{B / (ms_ft + ms_b) * 1e3:9.0f} traj/s train", flush=True)
