#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_fwd_<mode>/ (written by tools/pmc_fwd_variant.sh on the GPU box): per-dispatch means of the SQ
counters of the forward kernel, separately for the plain solve ("fwd") and the taping one ("tape").

    python tools/pmc_fwd_variant.py regs rows
"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kind(kernel_name):
    """"tape" for the TAPE = true instantiations (demangled: '..., true, false>' / mangled: 'Lb1ELb')"""
    return "tape" if (", true, " in kernel_name or "Lb1ELb" in kernel_name) else "fwd"


for mode in sys.argv[1:] or ["regs", "rows"]:
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_fwd_{mode}", "*", "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if "solve_fwd" in row["Kernel_Name"]:
                acc[(kind(row["Kernel_Name"]), row["Counter_Name"])].append(float(row["Counter_Value"]))
    print(mode)
    for (k, name), v in sorted(acc.items()):
        print(f"   {k:5s} {name:24s} n={len(v):3d}  mean {sum(v) / len(v):16.1f}")
