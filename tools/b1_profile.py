#!/usr/bin/env python
"""B=1, 4 s clip (BASELINE configs[1]) in a plain loop, for `rocprofv3 --kernel-trace --stats -- python tools/b1_profile.py`:
per-kernel durations of the latency path (profiles/r03/rocprofv3_b1_summary.txt).  GPU only."""
import importlib
import os
import sys

if len(sys.argv) == 3 and sys.argv[1] == "--summarise":     # python tools/b1_profile.py --summarise <kernel_stats.csv>
    import csv

    print("# rocprofv3 --kernel-trace --stats -- python tools/b1_profile.py : B=1, T=500 (4 s clip), FastNEWT, one stream, 220 forwards")
    tot = 0.0
    for r in list(csv.DictReader(open(sys.argv[2])))[:14]:
        if int(r["Calls"]) >= 200:
            tot += float(r["AverageNs"]) * int(r["Calls"]) / 220
        print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:8.1f} us  "
              f"min {float(r['MinNs']) / 1e3:7.1f}  max {float(r['MaxNs']) / 1e3:7.1f}")
    print(f"# sum of the per-forward kernels: {tot / 1e3:.1f} us")
    sys.exit(0)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nws = importlib.import_module("neural-waveshaping-synthesis_amd")
nws.ensure_default_config()
m = nws.NeuralWaveshaping.load_from_checkpoint(os.path.join(ROOT, "tests/golden/weights_vn.npz")).cuda().eval()
m.newt = nws.FastNEWT(m.newt)
torch.manual_seed(0)
f0, control = torch.rand(1, 1, 500, device="cuda"), torch.rand(1, 2, 500, device="cuda")
with torch.no_grad():
    for _ in range(20):
        m(f0, control)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        m(f0, control)
    e1.record()
    e1.synchronize()
print(f"B=1, T=500: {e0.elapsed_time(e1) / 200:.4f} ms per forward (back to back, one stream)")
