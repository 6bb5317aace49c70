#!/usr/bin/env python3
"""Capture golden vectors for the data side from the reference.  Build container only.

  G8  FourGIModel.simulate / model_equations / generate_dataset   data/generate4GI.py (imported as is)
  G9  GlucoseDataset                                              train/train_hybrid.py:43-155

train/train_hybrid.py itself is not importable here (tensorboard / arviz are not installed and stay absent), so
G9 executes the reference's GlucoseDataset class definition alone: the ClassDef node is taken from the parsed
module and run in a namespace holding numpy / pandas / torch.  Nothing is stubbed.  Only data (inputs and the
reference's outputs) is written to tests/golden/.
"""
import ast
import os
import sys

import numpy as np
import pandas as pd
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, os.path.join(REF, "data"))
import matplotlib  # noqa: E402
matplotlib.use("Agg")
from generate4GI import FourGIModel  # noqa: E402


def save(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name), **arrs)
    print("wrote", name, {k: np.asarray(v).shape for k, v in arrs.items()})


def set_bsl(m, b):
    m.BSLglc, m.BSLins, m.BSLglp, m.BSLglg, m.BSLgip = [float(v) for v in b]


def g8():
    rng = np.random.default_rng(8)
    cases = []
    # (patient_type, duration_h, interval_min, meal_times, meal_sizes)
    cfgs = [("T2DM", 5, 5, [0.5, 2.5], [75, 50]), ("HV", 5, 5, [1, 3], [75, 50]), ("T2DM", 20, 5, [1.0, 6.0, 11.5, 17.25],
            [75, 50, 60, 40]), ("HV", 3, 10, [], []), ("T2DM", 2, 15, [0.3], [100])]
    for ci, (ptype, dur, dt, mt, ms) in enumerate(cfgs):
        bsl = np.array([7.0, 50.0, 10.0, 25.0, 20.0]) * np.concatenate(
            [np.ones((1, 5)), rng.normal(1, [0.1, 0.15, 0.15, 0.15, 0.15], size=(3, 5))])
        outs = []
        for b in bsl:
            m = FourGIModel(ptype)
            set_bsl(m, b)
            t, *c = m.simulate(dur, dt, mt, ms)
            outs.append(np.stack(c, 1))
        cases.append(dict(ptype=ptype, dur=dur, dt=dt, mt=mt, ms=ms))
        save(f"g8_sim_{ci}.npz", patient_type=np.array(ptype), duration_h=dur, interval_min=dt, meal_time=np.array(mt, float),
             meal_size=np.array(ms, float), bsl=bsl, t_hours=t, conc=np.stack(outs))
    # RHS
    for ptype in ("T2DM", "HV"):
        m = FourGIModel(ptype)
        bsl = np.array([7.0, 50.0, 10.0, 25.0, 20.0]) * rng.normal(1, 0.1, size=(16, 5))
        y0 = np.array([7 * 9.33, 50 * 6.09, 160.0, 25 * 64.6, 20 * 9.21, 7 * 8.56, 50.0, 20 * 22.8])
        y = y0 * rng.uniform(0.5, 1.8, size=(16, 8))
        meal = np.where(rng.uniform(size=16) < 0.5, 0.0, rng.uniform(100, 1000, size=16))
        d = []
        for i in range(16):
            set_bsl(m, bsl[i])
            d.append(m.model_equations(list(y[i]), 0.0, float(meal[i])))
        save(f"g8_rhs_{ptype}.npz", bsl=bsl, y=y, meal=meal, d=np.array(d))
    # the generator end to end, seeded: same global-RNG stream as the reference
    for seed, n_sub, kw in [(0, 3, dict(duration_hours=5, sampling_interval_min=5, meal_times=[0.5, 2.5], meal_sizes=[75, 50],
                                         noise_cv=0.1)),
                            (1, 2, dict(duration_hours=4, sampling_interval_min=10, meal_times=[1, 3], meal_sizes=[75, 50],
                                         noise_cv=0.05))]:
        np.random.seed(seed)
        df = FourGIModel("T2DM").generate_dataset(n_subjects=n_sub, **kw)
        save(f"g8_dataset_seed{seed}.npz", table=df.values.astype(np.float64), columns=np.array(list(df.columns)), seed=seed,
             n_subjects=n_sub, **{k: np.asarray(v, float) for k, v in kw.items()})
    return


def reference_dataset_class():
    tree = ast.parse(open(os.path.join(REF, "train", "train_hybrid.py")).read())
    node = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GlucoseDataset"][0]
    ns = dict(torch=torch, np=np, pd=pd)
    exec(compile(ast.Module([node], []), "train_hybrid.py:GlucoseDataset", "exec"), ns)
    return ns["GlucoseDataset"]


def g9():
    DS = reference_dataset_class()

    def dump(name, path, **kw):
        ds = DS(path, **kw)
        items = [ds[i] for i in range(len(ds))]
        save(name, seq_len=kw["sequence_length"], stride=kw["stride"], normalize=int(kw.get("normalize", True)),
             observations=np.stack([it["observations"].numpy() for it in items]),
             initial_state=np.stack([it["initial_state"].numpy() for it in items]),
             time_points=np.stack([it["time_points"].numpy() for it in items]),
             meal=np.stack([it["external_inputs"]["meal"].numpy() for it in items]),
             tvns=np.stack([it["external_inputs"]["tVNS"].numpy() for it in items]),
             state_mean=ds.state_mean, state_std=ds.state_std, subject_of_window=np.array([s["subject_id"] for s in ds.sequences]))

    csv = os.path.join(REF, "data", "4gi_dataset.csv")
    ref_df = pd.read_csv(csv)                       # the reference's committed data file, as a numeric table
    save("g9_4gi_dataset_table.npz", table=ref_df.values.astype(np.float64), columns=np.array(list(ref_df.columns)))
    dump("g9_windows_4gi_61_30.npz", csv, sequence_length=61, stride=30)
    dump("g9_windows_4gi_20_10.npz", csv, sequence_length=20, stride=10)
    dump("g9_windows_4gi_20_7_raw.npz", csv, sequence_length=20, stride=7, normalize=False)
    # the frame of the reference's own tests/test_training.py:22-60 (no gip column, other column order, ragged subjects here)
    np.random.seed(0)
    rows = []
    for sid, n in [(0, 100), (1, 57), (2, 19), (3, 100)]:
        th = np.linspace(0, 5, n)
        g = 5.0 + 2.0 * np.sin(th) + 0.5 * np.random.randn(n)
        ins = 100.0 + 50.0 * np.sin(th + 0.5) + 10.0 * np.random.randn(n)
        glg = 50.0 + 10.0 * np.sin(th + 1.0) + 5.0 * np.random.randn(n)
        glp = 20.0 + 10.0 * np.sin(th + 1.5) + 2.0 * np.random.randn(n)
        mi = np.zeros(n)
        for mtm in (30, 90, 150):
            k = int(mtm / 300 * n)
            if k < n:
                mi[k] = 1.0
        for i in range(n):
            rows.append(dict(subject_id=sid, time_hours=th[i], time_minutes=th[i] * 60, glucose_mmol_L=g[i], insulin_pmol_L=ins[i],
                             glucagon_pmol_L=glg[i], glp1_pmol_L=glp[i], meal_indicator=mi[i]))
    df = pd.DataFrame(rows)
    path = "/tmp/_g9_frame.csv"
    df.to_csv(path, index=False)
    save("g9_frame_ragged.npz", table=pd.read_csv(path).values.astype(np.float64), columns=np.array(list(df.columns)))
    dump("g9_windows_ragged_20_10.npz", path, sequence_length=20, stride=10)
    os.unlink(path)


if __name__ == "__main__":
    g8()
    g9()
