#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats csv files of the two bench commands (one stream / default pipeline) -> one table of
average kernel durations.  Usage: python tools/rocprof_summary.py <1stream.csv> <pipeline.csv> > rocprofv3_summary.txt"""
import csv
import re
import sys


def load(path):
    rows = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"^void ", "", r["Name"])
            name = name.replace("(anonymous namespace)::", "")
            name = name.split("(")[0][:60]
            rows[name] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                          float(r["TotalDurationNs"]))
    return rows


one, pipe = load(sys.argv[1]), load(sys.argv[2])
print("# rocprofv3 --kernel-trace --stats, B=64 x T=500, vn checkpoint, FastNEWT, torch.rand inputs (tools/collect_profiles.sh)")
print("# average kernel duration in us: one stream (undisturbed) | default pipeline (2 audio + 2 control streams, kernels of "
      "neighbouring batches overlap)")
for name in sorted(one, key=lambda n: -one[n][4]):
    a = one[name]
    b = pipe.get(name)
    line = f"{name:60s} 1-stream: calls {a[0]:4d} avg {a[1]:8.1f} min {a[2]:8.1f} max {a[3]:8.1f}"
    if b:
        line += f" | pipeline: calls {b[0]:4d} avg {b[1]:8.1f} min {b[2]:8.1f} max {b[3]:8.1f}"
    print(line)
