export TMPDIR=/tmp
mkdir -p gpurun_out/s4
timeout 300 python -m pytest tests/test_gpu_streaming.py -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 600 bash tools/stream_hop_ab.sh 2>&1 | grep -E "hop |outputs" | grep -E "seven|five|default|outputs" | tee gpurun_out/s4/stream_hop_ab.txt
for m in 0 1; do
NWS_STREAM_FUSE_MLP=$m timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s4/hop_trace -- python scripts/time_streaming.py --batch-size 1 --num-hops 300 --no-graph > gpurun_out/s4/hop_trace.log 2>&1
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/s4/hop_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-7:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  {r['Kernel_Name'][:50]}  grid {r.get('Grid_Size_X','')} wg {r.get('Workgroup_Size_X','')}")
PY
rm -rf gpurun_out/s4/hop_trace
done
python scripts/time_buffer_sizes.py --use-fast-newt --checkpoint tests/golden/weights_vn.npz 2>/dev/null | grep '^buffer' | head -2
