#!/usr/bin/env python3
"""End-to-end training on the device, in the shape of the reference's train/train_hybrid.py (train_epoch :225-275, validate
:278-302, Adam + clip_grad_norm_(5.0) + CosineAnnealingLR :438-446) -- without the reference's CLI, tensorboard or pandas:

    cohort      hode.datagen.FourGIModel.generate_cohort   (replaces data/generate4GI.py; 8-state 4GI ODE, fp64, one subject per lane)
    dataset     hode.datagen.GlucoseDataset                (replaces the class at train/train_hybrid.py:43-155; windows + z-scores)
    model       models.HybridODENN                         (the reference's class surface; solve, adjoint, loss on the GPU)

    python examples/train_4gi.py [--subjects 512] [--epochs 5] [--batch 256]

Cohort, windows, parameters and optimiser state stay in HBM; per step the host ships the <= 20 physics-term indices loss() draws from
torch's CPU generator (as the reference does) and reads nothing back -- the running loss is accumulated on the device and read
once per epoch.  Needs an MI355X (no CPU fallback)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd"))

import torch  # noqa: E402
from hode.datagen import FourGIModel, GlucoseDataset  # noqa: E402
from models import HybridODENN  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subjects", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--lr", type=float, default=1e-3)
    args = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(0)

    # 20 h at 5-min sampling, four meals per subject: the reference's 4GI protocol (data/generate4GI.py:243-271), any cohort size
    g = torch.Generator(device=dev).manual_seed(0)
    table, status = FourGIModel("T2DM", device=dev).generate_cohort(args.subjects, duration_hours=20, meal_times=[1.0, 6.0, 11.0, 16.0],
                                                                    meal_sizes=[75.0, 60.0, 75.0, 40.0], generator=g)
    assert int(status.max()) == 0
    ds = GlucoseDataset(table, sequence_length=61, stride=30, normalize=True)          # train_hybrid.py's defaults
    n = len(ds)
    perm = torch.randperm(n, device=dev)
    n_val = max(1, n // 10)
    val_idx, train_idx = perm[:n_val], perm[n_val:]
    print(f"{args.subjects} subjects -> {n} windows of 61 points ({n - n_val} train / {n_val} validation), on {torch.cuda.get_device_name(0)}")

    model = HybridODENN(device=dev)                                                    # 4 x 64 residual network
    opt = torch.optim.Adam(model.parameters(), lr=args.lr)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.epochs)
    for epoch in range(args.epochs):
        model.train()
        t0, tot, nb = time.perf_counter(), torch.zeros((), device=dev), 0
        order = train_idx[torch.randperm(train_idx.numel(), device=dev)]
        for lo in range(0, order.numel(), args.batch):
            batch = ds.batch(order[lo:lo + args.batch])
            opt.zero_grad()
            loss = model.loss(batch, lambda1=1.0, lambda2=0.01)                        # configs/4gi_baseline.yaml:19-20
            loss.backward()                                                            # data term through the adjoint kernel
            torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
            opt.step()
            tot, nb = tot + loss.detach(), nb + 1                                      # (no host synchronisation per step)
        sched.step()
        model.eval()
        with torch.no_grad():
            val = float(model.loss(ds.batch(val_idx), lambda1=1.0, lambda2=0.01))
        torch.cuda.synchronize()
        print(f"epoch {epoch + 1}: train loss {float(tot) / nb:.5f}  val loss {val:.5f}  failed trajectories {model.solve_failures()}  "
              f"{(n - n_val) / (time.perf_counter() - t0):.0f} windows/s")


if __name__ == "__main__":
    main()
