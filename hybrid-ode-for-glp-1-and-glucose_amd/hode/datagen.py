"""Data side of the hot path on the device: host mirror of the reference's two data classes.

  FourGIModel      data/generate4GI.py:6-271      same constructor, attributes and methods; `simulate` /
                                                  `generate_dataset` run kernel K7 (one subject per lane) instead of one
                                                  scipy.odeint call per subject per interval.  `generate_dataset` consumes
                                                  numpy's global stream exactly as the reference does, so a seeded call
                                                  reproduces the reference's table to solver tolerance.
                                                  New: `generate_cohort` keeps everything on the device (torch RNG).
  GlucoseDataset   train/train_hybrid.py:43-155   same constructor / attributes / item layout; windows and z-scoring run
                                                  in kernel K8.  Also accepts a device table from `generate_cohort`, and
                                                  `batch(idx)` returns a device-resident batch for HybridODENN.loss.

No CPU fallback: both classes raise when libhode.so or a HIP device is missing.
"""
import numpy as np
import torch

from . import _capi as capi

_PAR_NAMES = ["CLglc", "CLglci", "Qglc", "VCglc", "VPglc", "CLins", "VCins", "Ke0ins", "VCglp", "VM_GLP", "KM_GLP", "CLglg",
              "VCglg", "CLgip", "VCgip", "Qgip", "VPgip", "GLCINS_S", "EMAX_1", "EC50_1", "HILL_1", "EMAX_4", "EC50_4",
              "FDGLP", "FDGIP", "FDGLG"]
_BSL_NAMES = ["BSLglc", "BSLins", "BSLglp", "BSLglg", "BSLgip"]
_BSL_DEFAULT = (7.0, 50.0, 10.0, 25.0, 20.0)              # generate4GI.py:65-71
_BSL_CV = (0.1, 0.15, 0.15, 0.15, 0.15)                   # generate4GI.py:227-231


def _device(device):
    dev = torch.device(device if device is not None else "cuda")
    if dev.type != "cuda" or not torch.cuda.is_available():
        raise capi.HodeError("the 4GI generator / dataset kernels need a HIP device (no CPU fallback)")
    return dev


def grid_points(duration_hours, sampling_interval_min):
    """generate4GI.py:168-170."""
    return len(np.arange(0, duration_hours * 60 + sampling_interval_min, sampling_interval_min))


class FourGIModel:
    def __init__(self, patient_type="T2DM", device=None):
        self.patient_type = patient_type
        self.device = device
        self.rtol, self.atol = 1e-10, 1e-12    # converged to ~6e-9; the reference (odeint at 1.49e-8) agrees to ~5e-7
        self._set_parameters()
        self._set_baseline_values()

    def _set_parameters(self):
        for name, v in zip(_PAR_NAMES, capi.fourgi_default_params(self.patient_type)):
            setattr(self, name, v)

    def _set_baseline_values(self):
        for name, v in zip(_BSL_NAMES, _BSL_DEFAULT):
            setattr(self, name, v)

    # attributes may be edited by the caller (as with the reference object): read them at call time
    def _par(self):
        return [float(getattr(self, n)) for n in _PAR_NAMES]

    def _bsl(self):
        return [float(getattr(self, n)) for n in _BSL_NAMES]

    def model_equations(self, y, t, meal_input=0):
        dev = _device(self.device)
        d = capi.fourgi_rhs(torch.tensor([self._bsl()], dtype=torch.float64, device=dev),
                            torch.as_tensor(np.asarray(y, dtype=np.float64), device=dev).view(1, 8),
                            torch.tensor([float(meal_input)], dtype=torch.float64, device=dev), self.patient_type, self._par())
        return d[0].tolist()

    def _tables(self, bsl, T, interval, meal_times, meal_sizes, z=None, noise_cv=0.0, subject0=0, z_tcb=None):
        return capi.fourgi_generate(bsl, T, float(interval), meal_times, meal_sizes, self.patient_type, self._par(), z=z,
                                    noise_cv=float(noise_cv), subject0=subject0, rtol=self.rtol, atol=self.atol, z_tcb=z_tcb)

    def simulate(self, duration_hours=5, sampling_interval_min=5, meal_times=[], meal_sizes=[]):
        """-> (t_hours, glucose, insulin, glp1, glucagon, gip) numpy arrays, generate4GI.py:159-212."""
        dev = _device(self.device)
        T = grid_points(duration_hours, sampling_interval_min)
        tab, _ = self._tables(torch.tensor([self._bsl()], dtype=torch.float64, device=dev), T, sampling_interval_min,
                              list(meal_times), list(meal_sizes))
        tab = tab.cpu().numpy()
        return (tab[:, 1],) + tuple(tab[:, 3 + c] for c in range(5))

    def add_measurement_noise(self, data, cv=0.1):
        noise = np.random.normal(0, cv * np.abs(data), size=np.shape(data))
        return data + noise

    def generate_dataset(self, duration_hours=5, sampling_interval_min=5, meal_times=[1, 3], meal_sizes=[75, 50], noise_cv=0.1,
                         n_subjects=10):
        """-> pandas DataFrame with the reference's 9 columns (generate4GI.py:221-271).

        numpy's global legacy stream is consumed in the reference's order -- per subject 5 baseline factors, then
        5 x T noise draws (glucose, insulin, glp1, glucagon, gip) -- so `np.random.seed(s)` gives the reference's table."""
        import pandas as pd
        dev = _device(self.device)
        T = grid_points(duration_hours, sampling_interval_min)
        draws = np.random.normal(size=(n_subjects, 5 + 5 * T))
        base = np.array(self._bsl())
        bsl = base * (1.0 + np.array(_BSL_CV) * draws[:, :5])          # BSL *= normal(1, cv)
        z = draws[:, 5:].reshape(n_subjects, 5, T)
        tab, _ = self._tables(torch.as_tensor(bsl, device=dev), T, sampling_interval_min, list(meal_times), list(meal_sizes),
                              z=torch.as_tensor(z, device=dev), noise_cv=noise_cv)
        df = pd.DataFrame(tab.cpu().numpy(), columns=capi.FOURGI_COLUMNS)
        df["subject_id"] = df["subject_id"].astype(np.int64)
        df["meal_indicator"] = df["meal_indicator"].astype(np.int64)
        return df

    def generate_cohort(self, n_subjects, duration_hours=5, sampling_interval_min=5, meal_times=(1, 3), meal_sizes=(75, 50),
                        noise_cv=0.1, generator=None, subject0=0):
        """Device-resident variant for large cohorts: baselines and noise from torch's generator ON the device, nothing
        touches the host.  meal_times / meal_sizes: [n_meals] shared or [n_subjects, n_meals] tensors.
        -> (table[n_subjects*T, 9] float64 on the device, status[n_subjects])."""
        dev = _device(self.device)
        T = grid_points(duration_hours, sampling_interval_min)
        base = torch.tensor(self._bsl(), dtype=torch.float64, device=dev)
        cv = torch.tensor(_BSL_CV, dtype=torch.float64, device=dev)
        bsl = base * (1.0 + cv * torch.randn(n_subjects, 5, dtype=torch.float64, device=dev, generator=generator))
        z = torch.randn(T, 5, n_subjects, dtype=torch.float64, device=dev, generator=generator) if noise_cv else None
        return self._tables(bsl, T, sampling_interval_min, meal_times, meal_sizes, z_tcb=z, noise_cv=noise_cv, subject0=subject0)


def merge_moments_over_group(moments, group=None):
    """All-gather one shard's {count, mean[6], M2[6]} over the process group and merge in rank order.
    -> (mean[6], std[6]) float64 CPU tensors, identical on every rank.  13 doubles per rank: the only exchange the
    sharded dataset needs (RCCL when the group's backend is "nccl"; staged through the host for gloo)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    buf = moments if dist.get_backend(group) == "nccl" else moments.cpu()
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)
    return capi.combine_moments(torch.stack([o.cpu() for o in out]))


class _Sequences:
    """Lazy stand-in for the reference's `dataset.sequences` list of dicts (train_hybrid.py:113-121): raw values."""

    def __init__(self, ds):
        self._ds = ds

    def __len__(self):
        return len(self._ds)

    def __getitem__(self, i):
        ds = self._ds
        if i < 0:
            i += len(ds)
        if not 0 <= i < len(ds):
            raise IndexError(i)
        r0 = int(ds._row0_host[i])
        rows = ds._table[r0:r0 + ds.sequence_length].cpu().numpy()
        n = rows.shape[0]

        def col(k, default):
            return rows[:, ds._cols[k]] if ds._cols.get(k, -1) >= 0 else np.full(n, default)
        states = np.stack([col("glucose", 0), col("insulin", 0), col("glucagon", 0), col("glp1", 0), col("ge", 0.0),
                           col("ffa", 1.0)], 1)
        inputs = np.stack(([col("meal", 0.0)] if "meal" in ds._cols else []) + [col("tvns", 0.0)], 1)
        return {"subject_id": ds._subject_of_window[i], "states": states, "inputs": inputs,
                "time": rows[:, ds._cols["time"]] / ds._time_div}


class GlucoseDataset(torch.utils.data.Dataset):
    def __init__(self, data_path, sequence_length=61, stride=30, normalize=True, device=None, group=None):
        """data_path: '.csv' / '.parquet' file in the reference's column format (train_hybrid.py:64-69), or a device
        table [rows, 9] from FourGIModel.generate_cohort, or a (table, column_names) pair.
        group: this process holds only a SHARD of the subjects (one process per GPU); pass the torch.distributed
        process group (True = the default group) and the z-score statistics are those of the whole dataset."""
        self.sequence_length = sequence_length
        self.stride = stride
        self.normalize = normalize
        dev = _device(device if device is not None else (data_path.device if torch.is_tensor(data_path) else None))
        self.device = dev
        if isinstance(data_path, str):
            table, names, sub_ids = self._read(data_path)
        else:
            table, names = (data_path, capi.FOURGI_COLUMNS) if torch.is_tensor(data_path) else data_path
            names = list(names)
            table = table.to(device=dev, dtype=torch.float64)
            sid = table[:, names.index("subject_id")]
            if not bool((sid[1:] >= sid[:-1]).all()):                  # groupby order (train_hybrid.py:101-103)
                table = table[torch.argsort(sid, stable=True)]
                sid = table[:, names.index("subject_id")]
            sub_ids = sid.cpu().numpy()
        self.state_cols = ["glucose_mmol_L", "insulin_pmol_L", "glucagon_pmol_L", "glp1_pmol_L", "ge", "ffa"]
        self.input_cols = (["meal_indicator"] if "meal_indicator" in names else []) + ["tvns"]
        cols = {"glucose": names.index("glucose_mmol_L"), "insulin": names.index("insulin_pmol_L"),
                "glucagon": names.index("glucagon_pmol_L"), "glp1": names.index("glp1_pmol_L")}
        for key, name in (("ge", "ge"), ("ffa", "ffa"), ("meal", "meal_indicator"), ("tvns", "tvns")):
            if name in names:
                cols[key] = names.index(name)
        if "time_minutes" in names:                                    # train_hybrid.py:92-98
            cols["time"], self._time_div = names.index("time_minutes"), 60.0
        elif "time_hours" in names:
            cols["time"], self._time_div = names.index("time_hours"), 1.0
        else:
            t = torch.arange(table.shape[0], dtype=torch.float64, device=dev) * 5 / 60.0
            table = torch.cat([table, t[:, None]], 1)
            cols["time"], self._time_div = table.shape[1] - 1, 1.0
        self._table, self._cols = table.contiguous(), cols
        # windows: subject by subject, start += stride (train_hybrid.py:105-121)
        uniq, first, counts = np.unique(sub_ids, return_index=True, return_counts=True)
        self.subjects = [u.item() if hasattr(u, "item") else u for u in uniq]
        starts, owner = [], []
        for u, lo, n in zip(self.subjects, first, counts):
            a = np.arange(0, n - sequence_length + 1, stride, dtype=np.int64)
            starts.append(lo + a)
            owner.extend([u] * len(a))
        self._row0_host = np.concatenate(starts) if starts else np.zeros(0, np.int64)
        self._subject_of_window = owner
        row0 = torch.as_tensor(self._row0_host, device=dev)
        given = None
        if normalize and group is not None:
            mom = capi.fourgi_window_moments(self._table, cols, row0, sequence_length, check_bounds=False)
            given = torch.cat(merge_moments_over_group(mom, None if group is True else group))
        self._states, self._meal, self._tvns, self._time, ms = capi.fourgi_windows(self._table, cols, self._time_div, row0,
                                                                                   sequence_length, normalize, mean_std=given,
                                                                                   check_bounds=False)
        ms = ms.cpu().numpy()
        self.state_mean, self.state_std = ms[:6].copy(), ms[6:].copy()
        self.sequences = _Sequences(self)

    @staticmethod
    def read_frame(path):
        """Host part of the loader: the file -> (fp64 table of the numeric columns, their names, subject ids), rows in
        groupby order.  `pd.read_csv` keeps its DEFAULT float parser, as the reference does (train_hybrid.py:64-67): it is
        not round-trip exact (about 8 % of the values of data/4gi_dataset.csv come back 1 ulp off), so a CSV and a
        parquet file of the same frame agree to 1e-12, not bitwise."""
        import pandas as pd
        if path.endswith(".csv"):
            df = pd.read_csv(path)
        elif path.endswith(".parquet"):
            df = pd.read_parquet(path)
        else:
            raise ValueError(f"Unsupported file format: {path}")
        df = df.iloc[np.argsort(df["subject_id"].values, kind="stable")]
        num = df.select_dtypes("number")
        return num.to_numpy(dtype=np.float64), list(num.columns), df["subject_id"].values

    def _read(self, path):
        table, names, sub_ids = self.read_frame(path)
        return torch.as_tensor(table, device=self.device), names, sub_ids

    def __len__(self):
        return self._states.shape[0]

    def batch(self, idx):
        """Device-resident batch in the layout HybridODENN.loss expects (what DataLoader collation + .to(device) of
        train_hybrid.py:238-246 produces)."""
        idx = torch.as_tensor(idx, device=self.device, dtype=torch.int64)
        obs = self._states[idx]
        return {"initial_state": obs[:, 0], "observations": obs, "time_points": self._time[idx],
                "external_inputs": {"meal": self._meal[idx], "tVNS": self._tvns[idx]}}

    def __getitem__(self, idx):
        if idx < 0:
            idx += len(self)
        if not 0 <= idx < len(self):
            raise IndexError(idx)
        obs = self._states[idx].cpu()
        return {"initial_state": obs[0], "observations": obs, "time_points": self._time[idx].cpu(),
                "external_inputs": {"meal": self._meal[idx].cpu(), "tVNS": self._tvns[idx].cpu()}}
