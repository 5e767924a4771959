#!/usr/bin/env python
"""Per-wave digest of a rocprofv3 --pmc pass directory, keyed by the FULL kernel name (template arguments kept): the
instruction counts of the oscillator kernel's timing ablations side by side.  CPU."""
import csv, glob, os, re, sys
from collections import defaultdict


def main(root, pat):
    agg = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            n = r["Kernel_Name"]
            if pat not in n:
                continue
            m = re.search(r"(\w+<[^>]*>)", n)
            agg[m.group(1) if m else n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(agg.items()):
        m = {n: sum(v) / len(v) for n, v in c.items()}
        w = m.get("SQ_WAVES", 0) or 1
        print(f"{k:40s} waves {w:8.0f} | " + " | ".join(f"{n.replace('SQ_', '')} {v / w:9.1f}" for n, v in sorted(m.items()) if n != "SQ_WAVES"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "exciter_newt_kernel")
