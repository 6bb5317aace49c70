"""Build bookkeeping of libhode.so: is the binary next to this file the one the sources next to it would give?

csrc/Makefile writes `libhode.so.srcsha` = sha256 over the concatenated sources (its STAMP_IN list) whenever it links the
product library.  `is_current()` recomputes that hash from the tree; `ensure(...)` runs `make` when the library is missing or
stale.  Nothing here touches the GPU; callers that must not spawn processes (anything running under rocprofv3, whose
preloaded tool has already initialised the GPU) pass `may_build=False` and get an error instead of a fork."""
import fcntl
import glob
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
SO = os.path.join(_HERE, "libhode.so")
LAB_SO = os.path.join(_HERE, "lab", "libhode_lab.so")


def _stamp_inputs():
    """csrc/Makefile STAMP_IN: sorted *.hip, sorted *.h, include/hode.h, Makefile"""
    hdr = os.path.normpath(os.path.join(CSRC, "..", "..", "include", "hode.h"))
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [hdr, os.path.join(CSRC, "Makefile")]


def source_stamp():
    """Hash of the sources in this tree, or None when they are not there (a binary-only install)."""
    files = _stamp_inputs()
    if not files or not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def lab_source_stamp():
    files = _stamp_inputs() + sorted(glob.glob(os.path.join(CSRC, "lab", "*.hip")))
    if not all(os.path.exists(f) for f in files):
        return None
    h = hashlib.sha256()
    for f in files:
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def binary_stamp(so=SO):
    try:
        return open(so + ".srcsha").read().strip()
    except OSError:
        return None


def lab_is_current():
    src = lab_source_stamp()
    return os.path.exists(LAB_SO) and (src is None or src == binary_stamp(LAB_SO))


def is_current():
    """True when libhode.so exists and was linked from exactly these sources (or no sources are present to compare with)."""
    if not os.path.exists(SO):
        return False
    src = source_stamp()
    return src is None or src == binary_stamp()


def ensure(lab=False, may_build=True, jobs=4, quiet=True):
    """Make sure the product library (and, with lab=True, hode/lab/libhode_lab.so) is built from the present sources.
    Concurrent callers (one per rank) serialise on a file lock; the Makefile links to a temporary name and renames it, so
    nobody can dlopen a half-written library."""
    need = (not is_current()) or (lab and not lab_is_current())
    if not need:
        return False
    if not may_build:
        raise RuntimeError(f"{SO} is missing or was not built from the sources in {CSRC} (binary {binary_stamp()}, sources "
                           f"{source_stamp()}) and this process may not start a build: run `make -C {CSRC}` first")
    with open(os.path.join(CSRC, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if (not is_current()) or (lab and not lab_is_current()):       # somebody else may have built meanwhile
                out = subprocess.DEVNULL if quiet else None
                subprocess.run(["make", "-C", CSRC, f"-j{jobs}", "ARCH=gfx950", "all"] + (["lab"] if lab else []), check=True, stdout=out)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return True
