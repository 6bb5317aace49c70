#!/usr/bin/env python3
"""Instruction mix of the gfx950 kernels in csrc/_obj/*.o (static counts over the whole kernel body).

    python tools/isa_mix.py solve_fwd_kernel solve_bwd_kernel        # substring match on the mangled name

Disassembles the device code object embedded in each object file with llvm-objdump and prints, per matching kernel,
the number of instructions by class: VALU (of which DPP FMAs, lane swaps, plain moves), SALU, s_nop, s_waitcnt,
LDS, VMEM, scratch.  Static counts, not executed counts -- good for "did this edit remove the s_nops", not for timing.
"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd", "csrc", "_obj")
LLVM = "/opt/rocm/lib/llvm/bin"


def device_asm(obj, tmp):
    """yield (kernel, [mnemonics]) of one host object file"""
    import shutil
    base = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, base)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", base], check=True, cwd=tmp, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    out = ""
    for co in sorted(glob.glob(base + ".*gfx950")):
        out += subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True,
                              text=True).stdout
    name, body = None, []
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name and body:
                yield name, body
            name, body = m.group(1), []
        elif name and line.startswith("\t"):
            body.append(line.strip().split("//")[0].strip())
    if name and body:
        yield name, body


def classify(ins):
    op = ins.split()[0]
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    pats = sys.argv[1:] or ["solve_fwd_kernel", "solve_bwd_kernel"]
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(OBJ, "*.o"))):
            try:
                kernels = list(device_asm(obj, tmp))
            except subprocess.CalledProcessError:
                continue
            for name, body in kernels:
                if not any(p in name for p in pats):
                    continue
                c = collections.Counter(classify(i) for i in body)
                dpp = sum(1 for i in body if "_dpp" in i.split()[0] or " row_" in i)
                swaps = sum(1 for i in body if i.startswith("v_permlane"))
                movs = sum(1 for i in body if i.startswith(("v_mov_b32", "v_accvgpr")))
                demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
                print(f"{demangled[:110]}\n    total {len(body)}  valu {c['valu']} (dpp {dpp}, swaps {swaps}, mov {movs})  salu {c['salu']}  "
                      f"s_nop {c['s_nop']}  waitcnt {c['s_waitcnt']}  lds {c['lds']}  vmem {c['vmem']}  scratch {c['scratch']}")


if __name__ == "__main__":
    main()
