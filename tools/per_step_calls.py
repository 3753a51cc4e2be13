#!/usr/bin/env python3
"""Launches per timed step of every kernel class, from two `rocprofv3 --kernel-trace --stats` runs of the same command that differ only in
--steps: (calls_b - calls_a) / (steps_b - steps_a).  Prints the rocclr (fill / copy) and ATen rows first, then the total.
usage: per_step_calls.py <dir a> <steps a> <dir b> <steps b>"""
import csv
import glob
import sys


def calls(d):
    out = {}
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = out.get(r["Name"], 0) + int(r["Calls"])
    return out


a, sa, b, sb = calls(sys.argv[1]), int(sys.argv[2]), calls(sys.argv[3]), int(sys.argv[4])
per = {k: (b.get(k, 0) - a.get(k, 0)) / (sb - sa) for k in set(a) | set(b)}
per = {k: v for k, v in per.items() if v}
odd = {k: v for k, v in per.items() if "rocclr" in k or "at::" in k}
print("per timed step: %d kernel classes, %.1f launches" % (len(per), sum(per.values())))
print("rocclr / ATen rows per step: %.1f" % sum(odd.values()))
for k, v in sorted(odd.items(), key=lambda kv: -kv[1]):
    print("  %6.2f  %s" % (v, k[:140]))
if "-v" in sys.argv:
    for k, v in sorted(per.items(), key=lambda kv: -kv[1]):
        print("  %6.2f  %s" % (v, k[:140]))
