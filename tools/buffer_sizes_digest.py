#!/usr/bin/env python
"""scripts/time_buffer_sizes.py summaries (FastNEWT, then exact NEWT) -> one small CSV.
Usage: python tools/buffer_sizes_digest.py fast_summary.txt exact_summary.txt > buffer_sizes.csv"""
import re
import sys

print("model,device,buffer_size,p50_s,p99_s")
for model, path in (("fastnewt", sys.argv[1]), ("newt", sys.argv[2])):
    for line in open(path):
        m = re.match(r"buffer\s+(\d+) samples.*p50\s+([\d.]+) us\s+p99\s+([\d.]+) us.*\((\w+)\)", line)
        if m:
            print(f"{model},gpu-{m.group(4).lower()},{m.group(1)},{float(m.group(2)) * 1e-6:.4e},{float(m.group(3)) * 1e-6:.4e}")
