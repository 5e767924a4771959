#!/usr/bin/env python
"""Per-kernel, per-wave digest of one rocprofv3 --pmc pass directory (counter_collection.csv)."""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name)[:28]


def main(root):
    agg = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        w = m.get("SQ_WAVES", 0) or 1
        parts = [f"{k:28s} waves {w:8.0f}"]
        for n, v in sorted(m.items()):
            if n == "SQ_WAVES":
                continue
            parts.append(f"{n.replace('SQ_', '')} {v / w:10.1f}/wave")
        print(" | ".join(parts))


if __name__ == "__main__":
    main(sys.argv[1])
