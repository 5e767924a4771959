#!/usr/bin/env python
"""Print a compact per-stream timeline of a steady-state slice of a rocprofv3 kernel trace (who overlaps whom)."""
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"[<(].*", "", name)
    return re.sub(r"^void ", "", name)[:22]


def main(root, n_kernels=40, skip_frac=0.6):
    path = sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(path))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    i0 = int(len(rows) * skip_frac)
    sel = rows[i0:i0 + n_kernels]
    t0 = int(sel[0]["Start_Timestamp"])
    streams = sorted({r.get("Stream_Id", r.get("Queue_Id", "?")) for r in sel})
    print("stream kernel                 start_us   dur_us   end_us")
    for r in sel:
        st = r.get("Stream_Id", r.get("Queue_Id", "?"))
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"{streams.index(st):4d}   {short(r['Kernel_Name']):22s} {a:9.1f} {b - a:8.1f} {b:8.1f}")
    # union busy time over the steady-state part
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[i0:])
    busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
    for a, b in iv[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    span = iv[-1][1] - iv[0][0]
    print(f"steady-state span {span / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, sum of kernel durations "
          f"{sum(b - a for a, b in iv) / 1e6:.3f} ms, kernels {len(iv)}")


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:3]))
