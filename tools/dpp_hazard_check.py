#!/usr/bin/env python3
"""Static check of the DPP / lane-swap read hazard in the compiler's assembly of the kernels:

    python tools/dpp_hazard_check.py csrc/hode_solve_fwd.hip [csrc/... more sources] [-DFLAG ...]

gfx9 / CDNA: a VALU instruction that WRITES a VGPR must be followed by two wait states before a DPP instruction (or a
v_permlane16/32_swap) READS that VGPR through the cross-lane path.  hipcc's hazard recognizer inserts them for its own instructions
but does not look INSIDE inline asm, and independent asm statements may be re-ordered by the scheduler -- so a v_mov_b32_dpp that the
source places two instructions behind its producer can end up right behind it in one instantiation and not in the next.  This script
walks every kernel of the given sources (straight-line: a label resets the window, which can only miss, not invent, a hazard) and
reports each DPP / swap read whose source register was written by a VALU instruction fewer than two wait states earlier.

Second check (the adjoint reads LDS through inline asm, with its own counted `s_waitcnt lgkmcnt(n)`, because hipcc would otherwise wait
for every outstanding vector-memory operation in front of a read that may alias an LDS-DMA target): no instruction may name the
destination registers of an LDS read before a wait has covered it -- hipcc does not know that the asm's outputs are not there yet and is
free to copy them.  LDS operations return in order: `lgkmcnt(n)` leaves the last n of them pending."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hybrid-ode-for-glp-1-and-glucose_amd", "csrc")
srcs = [a for a in sys.argv[1:] if not a.startswith("-")] or [os.path.join(CSRC, f) for f in
        ("hode_solve_fwd.hip", "hode_solve_bwd_ws.hip", "hode_solve_bwd.hip", "hode_rhs.hip", "hode_generic.hip")]
flags = [a for a in sys.argv[1:] if a.startswith("-")]


def regs(tok):
    """registers named by one operand: v12 -> {12}; v[4:7] -> {4..7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


total = 0
for src in srcs:
    out = os.path.join(tempfile.gettempdir(), "hz_" + os.path.basename(src) + ".s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "--cuda-device-only", "-S",
                    "-Wno-unused-function", *flags, src if os.path.isabs(src) else os.path.join(ROOT, src), "-o", out], check=True,
                   stderr=subprocess.DEVNULL)
    kernel, window, nk = None, [], 0                    # window: [(written vgprs, wait states since)]
    lds_q = []                                          # LDS operations in flight, in issue order: destination registers (empty for writes)
    for raw in open(out):
        line = raw.split(";")[0].strip()
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            kernel, window, lds_q = m.group(1), [], []
            continue
        if not line or line.startswith("."):
            if re.match(r"^\.LBB", raw):
                window, lds_q = [], []
            continue
        parts = line.replace(",", " ").split()
        op, ops = parts[0], parts[1:]
        # ---- LDS results named before their wait
        if op == "s_waitcnt":
            m2 = re.search(r"lgkmcnt\((\d+)\)", line)
            if m2:
                keep = int(m2.group(1))
                lds_q = lds_q[len(lds_q) - keep:] if keep else []
            elif "vmcnt" not in line and "expcnt" not in line:          # a numeric s_waitcnt: assume it waits for everything
                lds_q = []
        elif kernel and lds_q:
            named = set().union(*(regs(o) for o in ops)) if ops else set()
            for dst in lds_q:
                if dst & named:
                    total += 1
                    print(f"{os.path.basename(src)}: {kernel}: `{line}` names v{sorted(dst & named)}, the destination of an LDS read no wait has covered yet")
                    break
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            is_read = op.startswith(("ds_read", "ds_swizzle", "ds_bpermute", "ds_permute", "ds_consume", "ds_append"))
            lds_q.append(regs(ops[0]) if (is_read and ops) else set())
        ws = 1
        if op == "s_nop":
            ws = int(ops[0]) + 1
        is_valu = op.startswith("v_")
        cross = is_valu and (("dpp" in op) or any(o.startswith(("row_", "quad_perm", "wave_")) for o in ops) or "permlane" in op)
        if cross and kernel:
            if "permlane" in op:
                src_regs = set().union(*(regs(o) for o in ops[:2]))          # a swap reads (and writes) both operands
            else:
                src_regs = regs(ops[1]) if len(ops) > 1 else set()           # DPP applies to src0
            for written, age in window:
                if age < 2 and written & src_regs:
                    total += 1
                    print(f"{os.path.basename(src)}: {kernel}: `{line}` reads v{sorted(written & src_regs)} written {age} wait state(s) earlier")
        # age the window, then record this instruction's VGPR writes
        window = [(w, a + ws) for w, a in window if a + ws < 2]
        if is_valu and ops and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            dst = regs(ops[0])
            if "permlane" in op and len(ops) > 1:
                dst |= regs(ops[1])
            if dst:
                window.append((dst, 0))
    print(f"{os.path.basename(src)}: checked")
print(f"{total} hazard(s)")
sys.exit(1 if total else 0)
