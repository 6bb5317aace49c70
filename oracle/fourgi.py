"""ORACLE for the data side -- test infrastructure only, see oracle/oracle.py.

  simulate / rhs      ctypes over oracle/fourgi_oracle.c   (data/generate4GI.py:73-212)
  dataset_table       numpy restatement of FourGIModel.generate_dataset's table assembly
                      (data/generate4GI.py:214-271) for given standard-normal draws
  windows             numpy restatement of GlucoseDataset (train/train_hybrid.py:43-155)
"""
import ctypes as C

import numpy as np

from . import oracle as _o

COLUMNS = ["subject_id", "time_hours", "time_minutes", "glucose_mmol_L", "insulin_pmol_L", "glp1_pmol_L",
           "glucagon_pmol_L", "gip_pmol_L", "meal_indicator"]
BASELINE = np.array([7.0, 50.0, 10.0, 25.0, 20.0])        # generate4GI.py:65-71 (glc, ins, glp, glg, gip)
BASELINE_CV = np.array([0.1, 0.15, 0.15, 0.15, 0.15])     # :227-231
NOISE_CV_SCALE = np.array([1.0, 1.5, 1.5, 1.2, 1.3])      # :239-243 (glucose, insulin, glp1, glucagon, gip)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def simulate(bsl, T, interval_min, meal_time, meal_size, patient_type="T2DM", rtol=1e-10, atol=1e-12,
             max_steps=100000):
    """-> conc[B,T,5] (glucose, insulin, glp1, glucagon, gip), status[B], total accepted steps."""
    bsl = _f64(bsl).reshape(-1, 5)
    B = bsl.shape[0]
    mt, ms = _f64(meal_time), _f64(meal_size)
    per = int(mt.ndim == 2)
    n_meals = mt.shape[-1] if mt.size else 0
    conc = np.empty((B, T, 5))
    status = np.empty(B, np.int32)
    total = C.c_int64(0)
    _o.lib().fourgi_oracle_simulate(B, T, C.c_double(interval_min), int(patient_type == "HV"), _o._p(bsl), n_meals,
                                    _o._p(mt), _o._p(ms), per, C.c_double(rtol), C.c_double(atol), max_steps,
                                    _o._p(conc), _o._p(status), C.byref(total))
    return conc, status, total.value


def rhs(bsl, y, meal, patient_type="T2DM"):
    bsl, y, meal = _f64(bsl).reshape(-1, 5), _f64(y).reshape(-1, 8), _f64(meal).reshape(-1)
    d = np.empty_like(y)
    _o.lib().fourgi_oracle_rhs(y.shape[0], int(patient_type == "HV"), _o._p(bsl), _o._p(y), _o._p(meal), _o._p(d))
    return d


def grid_hours(T, interval_min):
    """generate4GI.py:168-170: t_minutes = arange(0, ...) ; t_hours = t_minutes / 60."""
    return (np.arange(T) * float(interval_min)) / 60.0


def dataset_table(conc, z, interval_min, meal_time, noise_cv=0.1, subject0=0):
    """conc[B,T,5] clean, z[B,5,T] standard normals (order glucose, insulin, glp1, glucagon, gip) -> table[B*T,9].

    generate4GI.py:214-219  noisy = data + z * (cv * |data|)  (np.random.normal(0, cv*|data|) = scale * z)
    generate4GI.py:246-257  columns, meal_indicator = any(|t - mt| < 0.01)."""
    conc, z = _f64(conc), _f64(z)
    B, T, _ = conc.shape
    t = grid_hours(T, interval_min)
    tab = np.empty((B, T, 9))
    tab[:, :, 0] = np.arange(subject0, subject0 + B)[:, None]
    tab[:, :, 1] = t
    tab[:, :, 2] = t * 60
    for c in range(5):
        d = conc[:, :, c]
        tab[:, :, 3 + c] = d + (noise_cv * NOISE_CV_SCALE[c] * np.abs(d)) * z[:, c, :] if noise_cv else d
    mt = _f64(meal_time).reshape(-1)
    ind = np.zeros(T)
    for m in mt:
        ind[np.abs(t - m) < 0.01] = 1.0
    tab[:, :, 8] = ind
    return tab.reshape(B * T, 9)


def windows(frame, sub_off, seq_len, stride, normalize=True):
    """GlucoseDataset restated on a column frame.

    frame: dict of equal-length float64 columns: time (already in hours, train_hybrid.py:92-98), glucose, insulin,
    glucagon, glp1, and optionally ge (default 0.0), ffa (default 1.0), meal (default: absent -> zeros), tvns
    (default 0.0);  sub_off[n_sub+1]: row range of each subject in groupby order (:101-103).
    -> dict(states[N,S,6] f32 normalised, meal[N,S] f32, tvns[N,S] f32, time[N,S] f32, mean[6], std[6], row0[N])."""
    n = len(frame["time"])
    cols = [frame["glucose"], frame["insulin"], frame["glucagon"], frame["glp1"],
            frame.get("ge", np.zeros(n)), frame.get("ffa", np.ones(n))]             # :72-82
    st = np.stack([_f64(c) for c in cols], 1)
    meal = _f64(frame["meal"]) if "meal" in frame else None
    tvns = _f64(frame.get("tvns", np.zeros(n)))
    row0 = []
    for s in range(len(sub_off) - 1):                                               # :105-121
        lo, hi = int(sub_off[s]), int(sub_off[s + 1])
        for a in range(0, hi - lo - seq_len + 1, stride):
            row0.append(lo + a)
    row0 = np.asarray(row0, np.int64)
    idx = row0[:, None] + np.arange(seq_len)[None, :]
    W = st[idx]                                                                     # [N,S,6]
    if normalize and len(row0):                                                     # :124-130
        allst = W.reshape(-1, 6)
        mean, std = allst.mean(0), allst.std(0) + 1e-6
    else:
        mean, std = np.zeros(6), np.ones(6)
    out = dict(states=((W - mean) / std).astype(np.float32),                        # :137-142
               meal=(meal[idx] if meal is not None else np.zeros(idx.shape)).astype(np.float32),
               tvns=tvns[idx].astype(np.float32), time=_f64(frame["time"])[idx].astype(np.float32),
               mean=mean, std=std, row0=row0)
    return out
