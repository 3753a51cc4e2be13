#!/usr/bin/env python3
"""GPU occupancy of the timed steps from a `rocprofv3 --kernel-trace` csv of `bench.py`: the union of all kernel intervals against wall time over the last
steps (delimited by patch_im2col launches), the idle gaps longer than a threshold, and how much of the time two or more kernels overlap.
usage: timeline_gaps.py <dir with *_kernel_trace.csv> [steps=10]"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
marks = [s for s, e, n in ev if "im2col" in n]
# the pipelined pass launches one im2col per image-parallel stream and step: take every 2nd (or every) mark as a step boundary candidates; use the last 2*nsteps+2 marks
# per interval between consecutive im2col launches: duration, idle share, share with >= 2 kernels running (the pipelined passes show overlap, the serial ones none)
def stats(t0, t1):
    pts = []
    for s, e, n in ev:
        if e > t0 and s < t1:
            pts.append((max(s, t0), 1)); pts.append((min(e, t1), -1))
    pts.sort()
    depth, last, h = 0, t0, {}
    for t, d in pts:
        if t > last:
            h[min(depth, 2)] = h.get(min(depth, 2), 0) + (t - last)
        depth += d
        last = t
    return h
if "--intervals" in sys.argv:
    for a, b in zip(marks[:-1], marks[1:]):
        h = stats(a, b)
        w = b - a
        print(f"  interval {w / 1e6:7.3f} ms   idle {h.get(0, 0) / w * 100:5.1f} %   one {h.get(1, 0) / w * 100:5.1f} %   two+ {h.get(2, 0) / w * 100:5.1f} %")
    sys.exit(0)
marks = marks[-(2 * nsteps + 2):]
t0, t1 = marks[0], marks[-1]
seg = [(s, e, n) for s, e, n in ev if e > t0 and s < t1]
busy, overlap, cur_end, gaps = 0, 0, t0, []
pts = []
for s, e, n in seg:
    s, e = max(s, t0), min(e, t1)
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth, last = 0, t0
hist = {}
for t, d in pts:
    if t > last:
        hist[depth] = hist.get(depth, 0) + (t - last)
        if depth == 0 and t - last > 2000:
            gaps.append((last - t0, t - last))
    depth += d
    last = t
wall = t1 - t0
print(f"window {wall / 1e6:.3f} ms over {len(marks) - 1} im2col intervals; kernels {len(seg)}")
for k in sorted(hist):
    print(f"  {k} kernel(s) running: {hist[k] / wall * 100:5.1f} % of the window")
print(f"idle gaps > 2 us: {len(gaps)}, total {sum(g for _, g in gaps) / 1e3:.1f} us; largest: {sorted(gaps, key=lambda g: -g[1])[:5]}")
