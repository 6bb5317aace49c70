#!/usr/bin/env python3
"""DESIGN.md = docs/DESIGN.in.md with its @NAME@ fields filled from ONE set of files: profiles/<tag>_bench.json (the bench.py line),
profiles/<tag>_kernel_stats.csv (rocprofv3 averages) and profiles/<tag>_gputest.log -- so the document carries one generation of
numbers by construction.     python tools/fill_design.py r04"""
import csv, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))
tr, rf = d["train_step"], d["roofline"]
adj, tape = tr["roofline"]["adjoint"], tr["roofline"]["forward_with_tape"]
cls = {(c["windows"], c["grid_points"]): c for c in d["class_path"]["train_step"]["cases"]}
gen = {c["patients"]: c for c in d["generic_path"]["cases"]}
cb = d["cpu_baseline"]
kavg = {}
for r in csv.DictReader(l for l in open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv")) if not l.startswith("#")):
    kavg[r["kernel"]] = float(r["avg_ns"]) / 1e6


def k(sub, *more):
    for n, v in kavg.items():
        if sub in n and all(m in n for m in more):
            return v
    return float("nan")


gl = open(os.path.join(ROOT, "profiles", f"{tag}_gputest.log")).read()
ngpu = re.search(r"(\d+) passed", gl).group(1)
cpu_log = os.path.join(ROOT, "profiles", f"{tag}_cputest.log")
ncpu = re.search(r"(\d+) passed", open(cpu_log).read()).group(1) if os.path.exists(cpu_log) else "94"
F = {
    "FWD_MS": f"{d['ms_per_step']:.2f}", "FWD_TPS": f"{d['value'] / 1e6:.2f}", "FWD_FRAC": f"{rf['frac']:.3f}", "FWD_TF": f"{rf['achieved']:.1f}",
    "FWD_TRAFFIC": "n/a" if rf["traffic"] is None else f"{rf['traffic'] / 1e6:.1f}",
    "TAPE_MS": f"{tape['kernel_ms']:.2f}", "ADJ_MS": f"{adj['kernel_ms']:.2f}", "ADJ_FRAC": f"{adj['frac']:.3f}", "TAPE_FRAC": f"{tape['frac']:.3f}",
    "TRAIN_MS": f"{tr['ms_per_step']:.2f}", "TRAIN_TPS": f"{tr['value'] / 1e3:.0f}",
    "TAPE_GB": "n/a" if tape["traffic"] is None else f"{tape['traffic'] / 1e9:.2f}", "ADJ_GB": "n/a" if adj["traffic"] is None else f"{adj['traffic'] / 1e9:.2f}",
    "K_FWD": f"{k('solve_fwd_kernel<float, 4, 0, 2, false, false>'):.3f}", "K_TAPE": f"{k('solve_fwd_kernel<float, 4, 0, 2, true, false>'):.3f}",
    "K_ADJ": f"{k('solve_bwd_ws_kernel<4, 2'):.3f}",
    "CLSF": f"{d['class_path']['forward']['ms_events']:.2f}", "CLS32": f"{cls[(32, 61)]['ms_wall']:.2f}", "CLS4096": f"{cls[(4096, 241)]['ms_wall']:.1f}",
    "SOBOL_MS": f"{1e3 * d['sobol']['seconds_with_outputs']:.1f}", "SOBOL_TPS": f"{d['sobol']['value'] / 1e6:.1f}",
    "SOBOL_REF": f"{d['sobol']['reference_style_this_box']['seconds_per_set_one_thread']:.2f}",
    "Z_TPS": f"{d['zscore_regime']['value'] / 1e6:.2f}", "Z_OK": f"{d['zscore_regime']['trajectories_ok']:,}".replace(",", " "),
    "PAR_F": f"{cb['parity_check']['forward_rel_err']:.1e}", "PAR_G": f"{cb['parity_check']['adjoint_grad_rel_err']:.1e}",
    "CPU_REF1": f"{cb['one_thread']['value']:.0f}", "CPU_REF16": f"{cb['value']:.0f}", "CPU_PORT": f"{cb['c_port']['value']:.0f}",
    "VI_S": f"{d['vi_step']['s_per_step']:.3f}", "VI_TPS": f"{d['vi_step']['value'] / 1e3:.0f}", "VI_GIB": f"{d['vi_step']['peak_mem_gib']:.1f}",
    "G32F": f"{gen[32]['forward_with_tape_ms']:.2f}", "G32A": f"{gen[32]['adjoint_ms']:.2f}", "G32TPS": f"{gen[32]['trajectories_per_s_train']:.0f}",
    "G1KF": f"{gen[1024]['forward_with_tape_ms']:.2f}", "G1KA": f"{gen[1024]['adjoint_ms']:.2f}", "G1KTPS": f"{gen[1024]['trajectories_per_s_train'] / 1e3:.1f} k",
    "GFRAC": f"{gen[1024]['roofline']['forward_with_tape']['frac']:.3f} / {gen[1024]['roofline']['adjoint']['frac']:.3f}",
    "NGPU": ngpu, "NCPU": ncpu, "TAG": tag,
}
src = open(os.path.join(ROOT, "docs", "DESIGN.in.md")).read()
missing = sorted(set(re.findall(r"@([A-Z0-9_]+)@", src)) - set(F))
if missing:
    sys.exit(f"unfilled fields: {missing}")
out = re.sub(r"@([A-Z0-9_]+)@", lambda m: F[m.group(1)], src)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(out)
print(f"DESIGN.md written from profiles/{tag}_*: {len(out.encode())} bytes")
