#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel trace / counter collection) into a small text summary for profiles/."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main(root, out):
    lines = []
    for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(list)
        for r in csv.DictReader(open(path)):
            agg[(short(r["Kernel_Name"]), r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        lines.append(f"# kernel trace: {os.path.relpath(path, root)}  (durations in us)")
        lines.append(f"{'kernel':70s} {'grid':>14s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>9s}")
        for (k, gx, gy), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append(f"{k:70s} {gx + 'x' + gy:>14s} {len(v):6d} {sum(v) / len(v):10.2f} {min(v):10.2f} {max(v):10.2f} {sum(v) / 1e3:9.3f}")
        lines.append("")
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(list)
        for r in csv.DictReader(open(path)):
            agg[(short(r["Kernel_Name"]), r.get("Grid_Size", "?"), r["Counter_Name"])].append(float(r["Counter_Value"]))
        lines.append(f"# counters: {os.path.relpath(path, root)}  (per-dispatch averages)")
        for (k, g, c), v in sorted(agg.items()):
            lines.append(f"{k:70s} grid {g:>10s} {c:>14s} avg {sum(v) / len(v):16.1f} over {len(v)} dispatches")
        lines.append("")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
