"""Data side, CPU part: the oracle's restatement of the reference's 4GI generator
(data/generate4GI.py) and dataset windows (train/train_hybrid.py:43-155) against vectors captured from the reference
itself (tools/capture_golden_data.py), plus the host logic of the mirror classes that needs no GPU."""
import os

import numpy as np
import pytest

from oracle import fourgi
from _data_helpers import frame_from_table, reference_stream


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize("ptype", ["T2DM", "HV"])
def test_oracle_4gi_rhs_matches_reference_bitwise(golden_dir, ptype):
    g = _g(golden_dir, f"g8_rhs_{ptype}.npz")
    d = fourgi.rhs(g["bsl"], g["y"], g["meal"], ptype)
    # same arithmetic in the same order: agreement to the last bits (libm pow may differ by an ulp)
    np.testing.assert_allclose(d, g["d"], rtol=1e-13, atol=0)


@pytest.mark.parametrize("case", range(5))
def test_oracle_4gi_simulate_matches_reference(golden_dir, case):
    g = _g(golden_dir, f"g8_sim_{case}.npz")
    T = len(g["t_hours"])
    conc, status, _ = fourgi.simulate(g["bsl"], T, float(g["interval_min"]), g["meal_time"], g["meal_size"], str(g["patient_type"]))
    assert (status == 0).all()
    assert np.array_equal(fourgi.grid_hours(T, float(g["interval_min"])), g["t_hours"])
    # the reference integrates with LSODA at rtol = atol = 1.49e-8; the restatement at 1e-10 / 1e-12
    np.testing.assert_allclose(conc, g["conc"], rtol=2e-6, atol=0)


def test_oracle_4gi_tolerance_convergence():
    a, _, na = fourgi.simulate(fourgi.BASELINE, 61, 5, [0.5, 2.5], [75, 50])            # defaults: 1e-10 / 1e-12
    b, _, nb = fourgi.simulate(fourgi.BASELINE, 61, 5, [0.5, 2.5], [75, 50], rtol=1e-13, atol=1e-15)
    assert nb > na
    np.testing.assert_allclose(a, b, rtol=2e-8)


@pytest.mark.parametrize("seed", [0, 1])
def test_oracle_generate_dataset_reproduces_the_reference_stream(golden_dir, seed):
    g = _g(golden_dir, f"g8_dataset_seed{seed}.npz")
    assert list(g["columns"]) == fourgi.COLUMNS
    n, dt = int(g["n_subjects"]), float(g["sampling_interval_min"])
    T = len(np.arange(0, float(g["duration_hours"]) * 60 + dt, dt))
    bsl, z = reference_stream(seed, n, T)
    conc, _, _ = fourgi.simulate(bsl, T, dt, g["meal_times"], g["meal_sizes"])
    tab = fourgi.dataset_table(conc, z, dt, g["meal_times"], float(g["noise_cv"]))
    ref = g["table"]
    assert tab.shape == ref.shape
    for c in (0, 1, 2, 8):                       # ids, both time columns, meal indicator: exact
        assert np.array_equal(tab[:, c], ref[:, c]), fourgi.COLUMNS[c]
    np.testing.assert_allclose(tab[:, 3:8], ref[:, 3:8], rtol=2e-6, atol=0)


@pytest.mark.parametrize("name,src", [("4gi_61_30", None), ("4gi_20_10", None), ("4gi_20_7_raw", None),
                                      ("ragged_20_10", "g9_frame_ragged.npz")])
def test_oracle_windows_match_reference_dataset(golden_dir, name, src):
    g = _g(golden_dir, f"g9_windows_{name}.npz")
    t = _g(golden_dir, src or "g9_4gi_dataset_table.npz")
    frame, off = frame_from_table(t["table"], list(t["columns"]))
    w = fourgi.windows(frame, off, int(g["seq_len"]), int(g["stride"]), bool(g["normalize"]))
    assert np.array_equal(w["states"], g["observations"])
    assert np.array_equal(w["states"][:, 0], g["initial_state"])
    assert np.array_equal(w["time"], g["time_points"])
    assert np.array_equal(w["meal"], g["meal"]) and np.array_equal(w["tvns"], g["tvns"])
    np.testing.assert_allclose(w["mean"], g["state_mean"], rtol=1e-13)
    np.testing.assert_allclose(w["std"], g["state_std"], rtol=1e-13)


def test_oracle_windows_edge_cases():
    t = np.arange(30, dtype=float)
    frame = dict(time=t / 12, glucose=t, insulin=2 * t, glucagon=3 * t, glp1=4 * t)
    # a subject shorter than one window contributes nothing; no windows at all -> identity statistics
    w = fourgi.windows(frame, [0, 10, 30], 20, 5)
    assert w["row0"].tolist() == [10]
    w0 = fourgi.windows(frame, [0, 10, 20, 30], 20, 5)
    assert w0["states"].shape == (0, 20, 6) and (w0["std"] == 1).all()
    # ge / ffa placeholders normalise to exactly 0 (mean 0 / 1, std 1e-6)
    assert (w["states"][..., 4:] == 0).all()


def test_read_frame_csv_vs_parquet(golden_dir, tmp_path):
    """Host half of GlucoseDataset's loader on both on-disk formats the reference accepts (train_hybrid.py:64-67).
    parquet round-trips the frame exactly; pandas' default CSV float parser does not (1 ulp on some values), which is
    why the GPU test compares the two datasets at 1e-12 instead of bitwise."""
    pd = pytest.importorskip("pandas")
    from hode.datagen import GlucoseDataset
    t = _g(golden_dir, "g9_4gi_dataset_table.npz")
    cols = list(t["columns"])
    df = pd.DataFrame(t["table"], columns=cols)
    csv = str(tmp_path / "frame.csv")
    df.to_csv(csv, index=False, float_format="%.17g")
    tab_c, names_c, sid_c = GlucoseDataset.read_frame(csv)
    assert names_c == cols and tab_c.dtype == np.float64 and tab_c.shape == t["table"].shape
    np.testing.assert_allclose(tab_c, t["table"], rtol=4e-16, atol=0)          # within 1-2 ulp, not necessarily equal
    assert np.array_equal(sid_c, t["table"][:, cols.index("subject_id")])
    try:
        pq = str(tmp_path / "frame.parquet")
        df.to_parquet(pq)
    except ImportError:
        pytest.skip("no parquet engine importable")
    tab_p, names_p, sid_p = GlucoseDataset.read_frame(pq)
    assert names_p == cols and np.array_equal(tab_p, t["table"]) and np.array_equal(sid_p, sid_c)
    # shuffled subjects come back in groupby order, rows of a subject in file order (stable)
    perm = np.random.default_rng(0).permutation(len(df))
    df.iloc[perm].to_parquet(pq)
    tab_s, _, sid_s = GlucoseDataset.read_frame(pq)
    assert (np.diff(sid_s) >= 0).all()
    want = t["table"][perm][np.argsort(t["table"][perm][:, cols.index("subject_id")], kind="stable")]
    assert np.array_equal(tab_s, want)
    with pytest.raises(ValueError):
        GlucoseDataset.read_frame(str(tmp_path / "frame.txt"))
    # the oracle's windows on either frame: statistics agree to 1e-12 (the bar the GPU test uses), z-scores to 1 fp32 ulp
    w = {}
    for key, tab in (("csv", tab_c), ("pq", tab_p)):
        frame, off = frame_from_table(tab, cols)
        w[key] = fourgi.windows(frame, off, 20, 10, True)
    np.testing.assert_allclose(w["csv"]["std"], w["pq"]["std"], rtol=1e-12)
    np.testing.assert_allclose(w["csv"]["mean"], w["pq"]["mean"], rtol=1e-12)
    np.testing.assert_allclose(w["csv"]["states"], w["pq"]["states"], rtol=2e-7, atol=1e-7)


# ---- host mirror, no GPU --------------------------------------------------------------------------------------
def test_mirror_parameters_match_reference_values():
    from hode.datagen import FourGIModel, grid_points
    m = FourGIModel("T2DM")
    assert (m.CLglc, m.CLglci, m.Qglc, m.VCglc, m.BSLglc, m.BSLgip) == (1.72, 0.0256, 26.5, 9.33, 7.0, 20.0)
    assert m.Ke0ins == np.exp(-0.159) and m.VM_GLP == np.exp(7.97) and m.EC50_4 == np.exp(4.59)
    h = FourGIModel("HV")
    assert (h.CLglc, h.CLglci) == (5.36, 0.072)
    assert grid_points(5, 5) == 61 and grid_points(20, 5) == 241 and grid_points(3, 10) == 19
    with pytest.raises(Exception):
        FourGIModel("T1DM")


def test_mirror_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hode import HodeError
    from hode.datagen import FourGIModel, GlucoseDataset
    with pytest.raises(HodeError):
        FourGIModel().simulate(1, 5)
    with pytest.raises(HodeError):
        FourGIModel().generate_dataset(n_subjects=1)
    with pytest.raises(HodeError):
        GlucoseDataset(torch.zeros(100, 9, dtype=torch.float64))


def test_4gi_argument_validation_without_gpu():
    import ctypes as C
    import hode
    lib = hode.load()
    par = (C.c_double * 26)()
    assert lib.hode_4gi_default_params(2, par) == -1
    assert lib.hode_4gi_default_params(1, par) == 0 and par[0] == 5.36
    one = C.c_void_p(256)           # never dereferenced: validation happens before any launch
    f = lib.hode_4gi_generate_f64
    base = [None, 4, 10, C.c_double(5.0), 0, None, one, 0, None, None, 0, None, C.c_double(0.0), C.c_int64(0), C.c_double(1e-9),
            C.c_double(1e-12), 100, one, None]

    def call(**kw):
        a = list(base)
        for k, v in kw.items():
            a[int(k[1:])] = v
        return f(*a)
    assert call(_1=-1) == -1 and call(_2=0) == -1 and call(_3=C.c_double(0.0)) == -1 and call(_4=7) == -1
    assert call(_6=None) == -1 and call(_7=2) == -1 and call(_14=C.c_double(0.0)) == -1 and call(_16=0) == -1
    assert call(_1=0) == 0          # empty cohort: nothing to do
    w = lib.hode_4gi_windows_f32
    ms = C.c_void_p(256)
    args = [None, one, 9, 2, C.c_double(60.0), 3, 4, 6, 5, -1, -1, 8, -1, one, C.c_int64(4), C.c_int64(61), 1, one, one, one, one, ms, one]
    bad = list(args); bad[3] = 9
    assert w(*bad) == -1
    bad = list(args); bad[9] = -2
    assert w(*bad) == -1
    bad = list(args); bad[15] = C.c_int64(0)
    assert w(*bad) == -1
    bad = list(args); bad[4] = C.c_double(0.0)
    assert w(*bad) == -1


# ---- sharded dataset: merging the window statistics of the shards -------------------------------------------------
def _np_moments(x):
    """{count, mean[6], M2[6]} of the rows x[n,6] -- what hode_4gi_window_moments_f64 returns for one shard."""
    if len(x) == 0:
        return np.zeros(13)
    m = x.mean(0)
    return np.concatenate([[len(x)], m, ((x - m) ** 2).sum(0)])


def test_combine_moments_equals_global_statistics():
    import hode
    rng = np.random.default_rng(0)
    x = rng.normal([5, 100, 50, 20, 0, 1], [2, 50, 10, 10, 0, 0], size=(5000, 6))
    for cuts in ([0, 5000], [0, 1, 5000], [0, 1200, 1200, 3100, 5000], [0, 0, 5000]):       # incl. empty shards
        mom = np.stack([_np_moments(x[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
        mean, std = hode.capi.combine_moments(mom)
        np.testing.assert_allclose(mean.numpy(), x.mean(0), rtol=1e-13, atol=1e-15)
        np.testing.assert_allclose(std.numpy(), x.std(0) + 1e-6, rtol=1e-12)
    mean, std = hode.capi.combine_moments(np.zeros((3, 13)))
    assert mean.tolist() == [0.0] * 6 and std.tolist() == [1.0] * 6


def _merge_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    from hode.datagen import merge_moments_over_group
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x = np.random.default_rng(1).normal(50, 20, size=(901, 6))
    lo, hi = [(0, 300), (300, 901)][rank] if world == 2 else [(0, 10), (10, 10), (10, 901)][rank]
    mean, std = merge_moments_over_group(torch.tensor(_np_moments(x[lo:hi])))
    q.put((rank, mean.numpy(), std.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_statistics_over_gloo(world):
    """One process per shard, a single all_gather of 13 doubles: every rank ends with the statistics of the whole set."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x = np.random.default_rng(1).normal(50, 20, size=(901, 6))
    for _, mean, std in res:
        assert np.array_equal(mean, res[0][1]) and np.array_equal(std, res[0][2])          # bit-identical replicas
        np.testing.assert_allclose(mean, x.mean(0), rtol=1e-13)
        np.testing.assert_allclose(std, x.std(0) + 1e-6, rtol=1e-12)
