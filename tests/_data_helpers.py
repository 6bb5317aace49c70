"""Shared helpers of the data-side tests: fixtures -> the column frame the oracle's `windows` takes."""
import numpy as np


def frame_from_table(table, columns):
    """The column selection of GlucoseDataset.__init__ (train_hybrid.py:72-98) for a numeric table in groupby order."""
    c = {n: table[:, i] for i, n in enumerate(columns)}
    f = dict(glucose=c["glucose_mmol_L"], insulin=c["insulin_pmol_L"], glucagon=c["glucagon_pmol_L"], glp1=c["glp1_pmol_L"])
    f["time"] = c["time_minutes"] / 60.0 if "time_minutes" in c else c["time_hours"]
    for k, n in (("meal", "meal_indicator"), ("tvns", "tvns"), ("ge", "ge"), ("ffa", "ffa")):
        if n in c:
            f[k] = c[n]
    sid = c["subject_id"]
    assert (np.diff(sid) >= 0).all()
    _, counts = np.unique(sid, return_counts=True)
    return f, np.concatenate([[0], np.cumsum(counts)])


def reference_stream(seed, n_subjects, T):
    """numpy's global stream in the order FourGIModel.generate_dataset consumes it (generate4GI.py:225-243)."""
    from oracle import fourgi
    np.random.seed(seed)
    draws = np.random.normal(size=(n_subjects, 5 + 5 * T))
    bsl = fourgi.BASELINE * (1.0 + fourgi.BASELINE_CV * draws[:, :5])
    return bsl, draws[:, 5:].reshape(n_subjects, 5, T)
