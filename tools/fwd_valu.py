#!/usr/bin/env python3
"""Vector instructions per DP5(4) stage of the tuned forward kernels, from the compiler's assembly (static):

    python tools/fwd_valu.py [-DHODE_FINISH_PAIRS ...]

Compiles csrc/hode_solve_fwd.hip for gfx950 with the given extra flags (device code only, to /tmp), finds the accept/reject
loop of solve_fwd_kernel<float, 4, 0, 2, TAPE, false> (the longest backward branch) and prints the vector-instruction count
of the loop body / 6 stages together with the opcode histogram of the whole kernel.  DESIGN.md section 4.3 quotes these."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd", "csrc", "hode_solve_fwd.hip")
out = os.path.join(tempfile.gettempdir(), "fwd_valu.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                "-Wno-unused-function", *sys.argv[1:], SRC, "-o", out], check=True)
txt = open(out).read()
for tape in ("0", "1"):
    name = f"_ZN4hode16solve_fwd_kernelIfLi4ELi0ELi2ELb{tape}ELb0ELb0EEEvNS_9SolveArgsIT_EEi"
    m = re.search(rf"^{re.escape(name)}:[^\n]*\n(.*?)\n\s*s_endpgm", txt, re.S | re.M)
    if not m:
        print("kernel not found:", name); continue
    lines = [l.strip() for l in m.group(1).splitlines()]
    pos, labels = [], {}
    for l in lines:
        if re.match(r"^\.LBB\d+_\d+:", l):
            labels[l.split(":")[0]] = len(pos)
        elif l and not l.startswith((";", ".")):
            pos.append(l.split(";")[0].strip())
    best = (0, 0, 0)
    for i, ins in enumerate(pos):
        b = re.match(r"s_cbranch\S*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", ins)
        if b:
            tgt = labels.get(b.group(1) or b.group(2))
            if tgt is not None and tgt < i and i - tgt > best[0]:
                best = (i - tgt, tgt, i)
    body = pos[best[1]:best[2]]
    valu = [x.split()[0] for x in body if x.startswith("v_")]
    h = collections.Counter(valu)
    print(f"tape={tape}: loop body {len(body)} instr, {len(valu)} vector = {len(valu) / 6:.1f} per stage; scratch "
          f"{sum(x.startswith('scratch_') for x in pos)}")
    print("   ", ", ".join(f"{k} {v}" for k, v in h.most_common(14)))
