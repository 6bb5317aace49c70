"""Host-side pieces of bench.py that need no GPU: the kernel-source hash that stamps profiles/pmc_traffic.json."""
import os
import shutil

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comment_stripper_keeps_code_and_string_literals():
    src = 'int a = 1; // trailing\n/* block\n comment */ const char *s = "// not a comment"; char c = \'"\';\n'
    out = bench._strip_c_comments(src)
    assert "trailing" not in out and "block" not in out
    assert '"// not a comment"' in out and "int a = 1;" in out and "'\"'" in out


def test_kernel_source_hash_ignores_comments_but_not_code(tmp_path):
    pkg = os.path.join("hybrid-ode-for-glp-1-and-glucose_amd", "csrc")
    dst = tmp_path / pkg
    shutil.copytree(os.path.join(ROOT, pkg), dst, ignore=shutil.ignore_patterns("_obj", "*.o", ".build.lock"))
    base = bench.kernel_source_sha(str(tmp_path))
    assert base == bench.kernel_source_sha()                      # same sources, other root
    f = dst / "hode_solve_fwd.hip"
    text = f.read_text()
    f.write_text("// a new comment line\n" + text.replace("namespace hode {", "namespace hode {   /* remark */", 1))
    assert bench.kernel_source_sha(str(tmp_path)) == base         # comments and white space do not count
    f.write_text(text.replace("dim3(64)", "dim3(64 )", 1).replace("return HODE_EUNSUPPORTED;", "return HODE_EUNSUPPORTED + 0;", 1))
    assert bench.kernel_source_sha(str(tmp_path)) != base         # code does


def test_committed_pmc_stamp_matches_the_committed_kernels():
    """profiles/pmc_traffic.json must have been taken on the kernels of this tree (bench.py drops the reading otherwise)."""
    import json

    import pytest
    stamp = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernel_source_sha"]
    if stamp != bench.kernel_source_sha():      # a reminder, not a gate: kernels may change in a container without a GPU
        pytest.skip("profiles/pmc_traffic.json is stale: re-run tools/profile_gpu.sh + tools/summarize_profile.py on the GPU box")


def test_binary_stamp_follows_the_sources(tmp_path, monkeypatch):
    """csrc/Makefile stamps libhode.so with a hash of its sources; hode/_build.py recomputes it from the tree.  The committed
    tree's binary is current; any change to a kernel source, a header, the C ABI header or the Makefile makes it stale, and a
    caller that may not build (bench.py --no-build: everything under rocprofv3) gets an error instead of a stale benchmark."""
    import pytest
    from hode import _build
    assert _build.is_current() and _build.source_stamp() == _build.binary_stamp()
    pkg = tmp_path / "pkg"
    shutil.copytree(os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd", "csrc"), pkg / "csrc",
                    ignore=shutil.ignore_patterns("_obj", "*.o", ".build.lock"))
    os.makedirs(tmp_path / "include")
    shutil.copy(os.path.join(ROOT, "include", "hode.h"), tmp_path / "include" / "hode.h")
    os.makedirs(pkg / "hode")
    shutil.copy(_build.SO, pkg / "hode" / "libhode.so")
    shutil.copy(_build.SO + ".srcsha", pkg / "hode" / "libhode.so.srcsha")
    monkeypatch.setattr(_build, "CSRC", str(pkg / "csrc"))
    monkeypatch.setattr(_build, "SO", str(pkg / "hode" / "libhode.so"))
    assert _build.is_current()                                        # same sources, other place
    f = pkg / "csrc" / "hode_optim.hip"
    f.write_text(f.read_text() + "\n// touched\n")
    assert not _build.is_current()
    with pytest.raises(RuntimeError, match="may not start a build"):
        _build.ensure(may_build=False)


def test_saltelli_design_of_the_sobol_leg():
    """bench.py's stand-in for SALib's saltelli.sample(problem, 1024) (reference plots/plot_all.py:139-158): N (2 D + 2) rows in the
    block layout A, AB_1..AB_D, BA_1..BA_D, B, inside the bounds of :140-148; the three outputs follow :191-193."""
    import numpy as np
    import bench
    s = bench.saltelli_sets(1024)
    D = len(bench.SOBOL_NAMES)
    assert s.shape == (1024 * (2 * D + 2), D) == (16384, 7)
    lo, hi = np.array(bench.SOBOL_BOUNDS).T
    assert (s >= lo).all() and (s <= hi).all()
    blk = s.reshape(1024, 2 * D + 2, D)
    A, B = blk[:, 0], blk[:, -1]
    for i in range(D):
        other = [j for j in range(D) if j != i]
        assert np.array_equal(blk[:, 1 + i][:, other], A[:, other]) and np.array_equal(blk[:, 1 + i][:, i], B[:, i])
        assert np.array_equal(blk[:, 1 + D + i][:, other], B[:, other]) and np.array_equal(blk[:, 1 + D + i][:, i], A[:, i])
    assert np.array_equal(bench.saltelli_sets(1024), s)           # seeded
    y = np.random.default_rng(0).random((5, 61, 6))
    want = np.stack([np.trapz(y[:, :, 0], dx=5 / 60, axis=1), y[:, :, 1].max(1), y[:, 6:, 3].mean(1)], 1)
    assert np.allclose(bench.sobol_outputs(y), want, rtol=1e-13)
    x0, t, meal, tvns = bench.sobol_inputs()
    assert x0.tolist() == [5.0, 60.0, 80.0, 0.0, 0.0, 1.0] and t.shape == (61,) and float(meal[0, 6]) == 75.0 and float(meal.sum()) == 75.0
