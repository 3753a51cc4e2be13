#!/usr/bin/env python3
"""Audit of the LDS reads issued from inline asm (attention_bwd.hip: tr_issue / tr_take; gemm_split.hip: lds_read_*_asm; the laboratory
attention kernels): hipcc treats an asm statement's outputs as valid at ;;#ASMEND, but a ds_read's destination registers hold garbage until
the covering `s_waitcnt lgkmcnt(N)` -- so NO instruction may read or write those registers between the read and that wait (a register-
allocator copy, a spill or a reschedule in that window would silently corrupt the operands: gfx9 has no interlock on LDS returns).

The check walks the assembly (`hipcc -S --cuda-device-only`) linearly per kernel: every `ds_read*` inside an ASM block opens a window on its
destination registers; every `s_waitcnt lgkmcnt(k)` (asm or compiler) closes the windows of all but the k youngest LDS operations (LDS
operations retire in order); any other instruction that names a register of an open window is reported.  Exit code 1 on a violation.

usage: asm_window_audit.py file.hip [extra hipcc flags ...]      (or: asm_window_audit.py --asm file.s)"""
import re
import subprocess
import sys
import tempfile
import os

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
BASE = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-S", "--cuda-device-only"]


def regs_of(token):
    """v12 -> {12}; v[12:15] -> {12..15}"""
    out = set()
    for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", token):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def audit(asm_text):
    violations, kernels, asm_reads = [], 0, 0
    in_asm = False
    kernel = None
    lds_seq = 0                     # LDS operations issued so far in this kernel
    open_windows = []               # (seq, regs, line number, text)
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        if re.match(r"^_Z\w+:", line) or re.match(r"^\w+:\s*; @", line):
            kernel, lds_seq, open_windows = line.split(":")[0], 0, []
            kernels += 1
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith((";", ".", "//")) or re.match(r"^[.\w$]+:$", line) or kernel is None:
            continue
        op = line.split()[0]
        body = line.split(";")[0]
        if op.startswith("s_waitcnt"):
            m = re.search(r"lgkmcnt\((\d+)\)", body)
            if m:
                keep = int(m.group(1))
                open_windows = [w for w in open_windows if w[0] > lds_seq - keep]
            elif "lgkmcnt" not in body and "vmcnt" not in body and "expcnt" not in body:
                open_windows = []                               # s_waitcnt 0 style
            continue
        if op.startswith("ds_"):
            lds_seq += 1
            if in_asm and op.startswith("ds_read"):
                dst = body.split(None, 1)[1].split(",")[0]
                open_windows.append((lds_seq, regs_of(dst), ln, line))
                asm_reads += 1
                # the read's own address register may not be one of ITS pending destinations either -- checked below for older windows only
                used = regs_of(body.split(",", 1)[1]) if "," in body else set()
            else:
                used = regs_of(body)
        else:
            used = regs_of(body)
        for seq, regs, wln, wtext in open_windows:
            if op.startswith("ds_") and in_asm and ln == wln:
                continue
            hit = used & regs
            if hit:
                violations.append(f"{kernel}: line {ln}: `{line}` touches v{sorted(hit)} while `{wtext}` (line {wln}) is still in flight")
    return violations, kernels, asm_reads


def main():
    args = sys.argv[1:]
    if args and args[0] == "--asm":
        text = open(args[1]).read()
    else:
        src, extra = args[0], args[1:]
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "a.s")
            here = os.path.dirname(os.path.abspath(src))
            subprocess.run([HIPCC] + BASE + extra + ["-I", here, "-o", out, src], check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
    v, k, r = audit(text)
    print(f"{k} kernels, {r} asm LDS reads, {len(v)} violation(s)")
    for x in v[:20]:
        print("  " + x)
    return 1 if v else 0


if __name__ == "__main__":
    sys.exit(main())
