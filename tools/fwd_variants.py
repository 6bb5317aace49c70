"""Forward-kernel tuning points (development aid).  One process per variant (the environment switches are read once):
    (default)                          register-resident kernel (hode_solve_fwd.hip)
    HODE_FWD=wg HODE_FWD_CFG=<NREG><WPB/4>   workgroup kernel (csrc/lab/hode_solve_fwd_wg.hip): 4, 22, default (14)
    HODE_FWD=quad                      four trajectories per four waves, column-split weights (csrc/lab/hode_solve_fwd_quad.hip)
    HODE_FWD=rows                      four trajectories per four waves, weights split by output rows / input blocks (csrc/lab/hode_solve_fwd_rows.hip)
Usage: python tools/fwd_variants.py [B ...]   -> one line per batch size; the first run writes /tmp/fwd_ref_<B>.pt, later runs compare bitwise."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd")); sys.path.insert(0, ROOT)
import torch, hode, bench
_m = os.environ.get("HODE_FWD", "")
tag = "wg cfg " + os.environ.get("HODE_FWD_CFG", "default") if _m.startswith("w") else ("quad" if _m.startswith("q") else ("rows" if _m.startswith("ro") else "regs"))
dev = torch.device("cuda")
for B in [int(v) for v in sys.argv[1:]] or [4096]:
    x0, t, meal, tv = (v.to(dev) for v in bench.synth_cohort(B, 1000))
    ode, nn = bench.ODE_DEFAULT.to(dev), bench.synth_weights(0).to(dev)
    for _ in range(3): s = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): s = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    st = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4, want_tape=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5): st = hode.solve_fwd(x0, t, meal, tv, None, ode, nn, 64, 4, want_tape=True, tape=st.tape)
    e1.record(); torch.cuda.synchronize()
    mst = e0.elapsed_time(e1) / 5
    ref = f"/tmp/fwd_ref_{B}.pt"
    same = "ref written"
    if os.path.exists(ref):
        r = torch.load(ref)
        same = f"bitwise y {bool(torch.equal(r['y'], s.y.cpu()))} nfev {bool(torch.equal(r['nfev'], s.nfev.cpu()))} tape-y {bool(torch.equal(r['y'], st.y.cpu()))}"
    else:
        torch.save({"y": s.y.cpu(), "nfev": s.nfev.cpu()}, ref)
    print(f"{tag:16s} B={B:6d}: fwd {ms:7.3f} ms ({B/ms*1e3/1e6:.3f} M traj/s)  fwd+tape {mst:7.3f} ms  ok {int((s.status==0).sum())}  {same}", flush=True)
