#!/usr/bin/env python3
"""Static scan of the product kernels' vector-memory schedule: compiles each csrc/*.hip to gfx950 assembly and prints, per kernel, the order
of loads (L), stores (S) and `s_waitcnt vmcnt(n)` ([wn]).  Two patterns cost real time on gfx9 and are invisible in the source:
  * `L[w0]L[w0]...`   one load in flight at a time: hipcc sank the loads of an unrolled loop next to their uses, or a predicated load
                       (`cond ? p[i] : 0`) became a branch with its own vmcnt(0);
  * `S L.. [w0] S`    a load between two stores: vmcnt counts stores too, so the wait for the load is a wait for the store in front of it --
                       one store round trip per store instruction.
Round 3 removed both from the Gram / heads / decoder-backward kernels, the discriminator convolutions, the key-hook / patch-embedding
drains, LayerNorm and its backward.   usage: python tools/vmem_pattern_scan.py [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ucod_dpl_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only", "-S"]
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1", "-fno-slp-vectorize"], "attention_fp8.hip": ["-fno-slp-vectorize"],
         "disc.hip": ["-fno-slp-vectorize"]}


def scan(path):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA.get(os.path.basename(path), []) + [path, "-o", tmp.name], check=True,
                       stderr=subprocess.DEVNULL, cwd=CSRC)
        txt = open(tmp.name).read()
    for f in re.split(r"\n(?=_Z\w+:)", txt):
        name = f.split(":", 1)[0]
        if not name.startswith("_Z"):
            continue
        seq = []
        for line in f.split("\n"):
            if re.search(r"global_store|buffer_store", line):
                seq.append("S")
            elif re.search(r"global_load|buffer_load", line):
                seq.append("L")
            else:
                m = re.search(r"s_waitcnt vmcnt\((\d+)\)", line)
                if m:
                    seq.append(f"[w{m.group(1)}]")
        s = "".join(seq)
        single = len(re.findall(r"L\[w0\]", s))
        between = len(re.findall(r"S(?:L+)\[w[0-3]\]", s))
        flag = " <-- one load in flight" if single >= 5 else ""
        flag += " <-- loads between stores" if between >= 4 else ""
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{os.path.basename(path)}: {demangled[:90]}{flag}\n    {s[:240]}")


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    for p in files:
        scan(p if os.path.isabs(p) else os.path.join(CSRC, os.path.basename(p)))
