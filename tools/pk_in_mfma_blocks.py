#!/usr/bin/env python3
"""Lint: basic blocks of the product kernels that hold both MFMAs and packed-f32 vector instructions (v_pk_{add,mul,fma}_f32).
On gfx950 a v_pk_*_f32 does not overlap with an MFMA in flight on the same SIMD -- it costs ~10 cycles of matrix-pipe time
(tools/probes/acc_transpose_probe.hip, profiles/r04_acc_transpose_probe.txt) -- so such blocks lose matrix throughput; hipcc forms these
instructions from two-element float vectors and, with the SLP vectoriser on, from adjacent scalar operations.
usage: python tools/pk_in_mfma_blocks.py [file.hip ...]      (default: every .hip of ucod_dpl_amd/csrc; flags taken from `make -n`)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ucod_dpl_amd", "csrc")


def flags_of(name):
    """the product compile line of build/<name>.o according to the Makefile"""
    out = subprocess.run(["make", "-n", "-B", "-C", CSRC, f"build/{name}.o"], capture_output=True, text=True).stdout
    for line in out.splitlines():
        if f"{name}.hip" in line and " -c " in line:
            toks = line.split()
            return [t for t in toks[1:] if t not in ("-c", "-fPIC") and not t.endswith(".hip") and not t.endswith(".o") and t != "-o"]
    return ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast"]


def scan(path):
    name = os.path.basename(path)[:-4]
    asm = f"/tmp/pklint_{name}.s"
    cmd = ["/opt/rocm/bin/hipcc"] + flags_of(name) + ["--cuda-device-only", "-S", os.path.abspath(path), "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC)
    if r.returncode != 0:
        print(f"{name}: compile failed: {r.stderr[-300:]}")
        return 0
    t = open(asm).read()
    hits = 0
    for m in re.finditer(r"\n(_Z\S+):.*?\n(.*?)\n\s*s_endpgm", t, re.S):
        kern, body = m.group(1), m.group(2).split("\n")
        cur, blocks = ["entry", []], []
        for line in body:
            if re.match(r"^\.LBB\d+_\d+:", line):
                blocks.append(cur)
                cur = [line.split(":")[0], []]
            else:
                cur[1].append(line)
        blocks.append(cur)
        for bn, ls in blocks:
            mf = sum("v_mfma" in x for x in ls)
            pk = sum(bool(re.search(r"v_pk_(fma|mul|add)_f32", x)) for x in ls)
            if mf >= 4 and pk:
                hits += 1
                print(f"{name}: {kern[:70]} {bn}: {mf} MFMAs, {pk} packed-f32 instructions")
    return hits


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    total = sum(scan(f) for f in files)
    print(f"{total} block(s) with MFMAs and packed-f32 instructions")
    sys.exit(1 if total else 0)
