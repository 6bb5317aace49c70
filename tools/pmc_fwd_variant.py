#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_fwd_<mode>/ (tools/pmc_fwd_variant.sh): per-dispatch averages of the SQ counters of the forward kernel."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in sys.argv[1:] or ["regs", "rows"]:
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_fwd_{mode}", "*", "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            if "solve_fwd" in k and "Lb1E" not in k.split("solve_fwd")[1][:40].replace("Lb0ELb0", ""):
                pass
            if "solve_fwd" in k:
                acc[(("tape" if ", true, " in k or "Lb1ELb" in k else "fwd"), row["Counter_Name"])].append(float(row["Counter_Value"]))
    print(mode)
    for (kind, name), v in sorted(acc.items()):
        print(f"   {kind:5s} {name:24s} n={len(v):3d}  mean {sum(v)/len(v):16.1f}")
