#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh run (gpurun_out/prof_<tag>/) into the tracked files under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (hode kernels + top others)
   profiles/<tag>_pmc.json           per-kernel PMC averages (each counter group collected in its own pass)
   profiles/pmc_traffic.json         HBM bytes per launch of the solve kernels, read by bench.py ("traffic")
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports half of the bytes of a streaming read, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("void ", "")
    return name[:name.index("(")] if "(" in name else name


stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if len(stats) > 1:
    sys.exit(f"{src} holds the output of {len(stats)} profiling runs (gpurun merges into gpurun_out/): delete the directory before "
             "the run, or the stale files -- the counters of different kernel versions would be averaged together")
rows = list(csv.DictReader(open(stats[-1])))
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 40 --warmup 5 --train-steps 20 --no-cpu-baseline --no-vi --no-generic --no-zscore --no-sobol --no-class-path --no-build\n")
    f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
    for r in rows[:18]:
        f.write(f"\"{short(r['Name'])[:90]}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")

zstats = sorted(glob.glob(os.path.join(src, "trace_zscore", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
if zstats:
    with open(os.path.join(dst, f"{tag}_kernel_stats_zscore.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 tools/zscore_launches.py   (7 launches, x0 ~ N(0,1)^6, 4 096 x 241)\n")
        f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
        for r in list(csv.DictReader(open(zstats[-1])))[:4]:
            f.write(f"\"{short(r['Name'])[:90]}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")

pmc = collections.defaultdict(dict)
for fcsv in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fcsv)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        if k.startswith("hode::"):
            for c, v in d.items():
                pmc[k][c] = sum(v) / len(v)
            pmc[k].setdefault("_launches", len(next(iter(d.values()))))
out = {"command": "rocprofv3 --pmc <group> -- python3 bench.py --steps 12 --warmup 3 --train-steps 12 --no-cpu-baseline ...",
       "note": "per-launch averages; one rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE in separate passes)",
       "kernels": pmc}
traffic = {}
for k, d in pmc.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["hbm_bytes_per_launch_corrected"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
        targs = [a.strip() for a in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
        tape_inst = len(targs) >= 5 and targs[4] == "true"          # solve_fwd_kernel<R, NL, METHOD, LB, TAPE, GD>
        if "generic" in k or "_wg_" in k:
            continue
        if "solve_fwd" in k and not tape_inst:          # forward-only instantiation (no tape)
            traffic["solve_fwd_hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
            traffic["solve_fwd_fetch_kib_raw"] = d["FETCH_SIZE"]
            traffic["solve_fwd_write_kib_raw"] = d["WRITE_SIZE"]
        if "solve_fwd" in k and tape_inst:              # training instantiation: also writes the stage tape
            traffic["solve_fwd_tape_hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
        if "solve_bwd" in k:
            traffic["solve_bwd_hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
        for key in ("fourgi_generate", "win_moment", "win_emit"):      # data side (bench.py "data_side" leg)
            if key in k:
                traffic[f"{key}_hbm_bytes_per_launch"] = d["hbm_bytes_per_launch_corrected"]
# derived figures for the solve kernels (what VERDICT r3 item 1 asks for): how busy the SIMDs' vector pipes are, what share of a wave's
# life is spent waiting, what a vector instruction costs.  GRBM_GUI_ACTIVE is summed over the 8 XCDs; the chip has 1 024 SIMDs;
# SQ_ACTIVE_INST_VALU counts in units of 4 cycles.
for k, d in pmc.items():
    if all(c in d for c in ("SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU")) and d["GRBM_GUI_ACTIVE"] > 0:
        cyc = d["GRBM_GUI_ACTIVE"] / 8.0
        d["derived"] = {"kernel_cycles": cyc,
                        "valu_busy": 4.0 * d["SQ_ACTIVE_INST_VALU"] / (1024.0 * cyc),
                        "wait_any_share_of_wave_cycles": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                        "pipe_cycles_per_valu_instruction": 4.0 * d["SQ_ACTIVE_INST_VALU"] / d["SQ_INSTS_VALU"],
                        "valu_instructions_per_wave": d["SQ_INSTS_VALU"] / d["SQ_WAVES"] if d.get("SQ_WAVES") else None}
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
traffic["source"] = f"profiles/{tag}_pmc.json (2*FETCH_SIZE + WRITE_SIZE) KiB per launch"
traffic["workloads"] = {"solve_*": "B=4096 T=241 fp32 (bench.py headline / train legs)",
                        "fourgi_generate / win_*": "65 536 subjects x 61 grid points fp64, 196 608 windows 31/15 (bench.py data_side leg)"}
sha_file = os.path.join(src, "kernel_source_sha.txt")
traffic["kernel_source_sha"] = open(sha_file).read().strip() if os.path.exists(sha_file) else None
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read())
print(json.dumps(traffic, indent=1))
