#!/usr/bin/env python3
"""code object -> C array header:  embed.py file.co symbol > header.h"""
import sys
data = open(sys.argv[1], "rb").read()
sym = sys.argv[2]
out = [f"// generated from {sys.argv[1].split('/')[-1]} ({len(data)} bytes) by tools/attn_asm/embed.py", f"alignas(4096) static const unsigned char {sym}[] = {{"]
for i in range(0, len(data), 24):
    out.append("  " + ",".join(str(b) for b in data[i:i + 24]) + ",")
out.append("};")
print("\n".join(out))
