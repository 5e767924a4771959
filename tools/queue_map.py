#!/usr/bin/env python
"""Which hardware queue each kernel of a rocprofv3 --kernel-trace run went through (streams that share a queue serialise).

    python tools/queue_map.py <rocprofv3 output dir>
Prints, per Queue_Id, the kernels it carried with their counts and average durations."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"[<(].*", "", name)[:34]


def main(root):
    paths = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))
    if not paths:
        print("no kernel_trace.csv under", root)
        return
    rows = list(csv.DictReader(open(paths[0])))
    cols = rows[0].keys()
    qk = "Queue_Id" if "Queue_Id" in cols else None
    sk = "Stream_Id" if "Stream_Id" in cols else None
    print("columns:", ", ".join(cols))
    per = defaultdict(lambda: defaultdict(list))
    for r in rows:
        key = (r.get(qk, "?") if qk else "?", r.get(sk, "?") if sk else "?")
        per[key][short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for key in sorted(per):
        print(f"queue {key[0]} stream {key[1]}:")
        for k, v in sorted(per[key].items(), key=lambda kv: -sum(kv[1])):
            print(f"    {k:36s} x{len(v):5d}  avg {sum(v) / len(v):9.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
