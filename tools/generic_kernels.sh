#!/bin/bash
# Per-kernel breakdown of the runtime-size path (csrc/generic.hip) under rocprofv3 ON THE GPU BOX:
#   bash tools/generic_kernels.sh B T   ->  stdout: the end-to-end lines of tools/generic_profile.py + average kernel durations
export TMPDIR=/tmp
OUT=gpurun_out/generic_kernels
rm -rf "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python tools/generic_profile.py $1 $2 > "$OUT.log" 2>&1
grep -a " T $2:" "$OUT.log"
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
print(f'{"kernel":72s} calls   avg us')
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("g_", "td_mlp", "col125", "row512", "row_kernel", "control_gru")):
        n = n.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f'{n[:72]:72s} {r["Calls"]:>5} {float(r["AverageNs"]) / 1e3:8.1f}')
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*trace.csv" -delete
