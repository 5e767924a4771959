export TMPDIR=/tmp
mkdir -p gpurun_out/s4
for mode in 0 1; do
NWS_STREAM_SPLIT_REVERB=$mode timeout 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s4/hop_trace$mode -- python scripts/time_streaming.py --batch-size 1 --num-hops 300 --no-graph > gpurun_out/s4/hop_trace.log 2>&1
tail -1 gpurun_out/s4/hop_trace.log
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/s4/hop_trace$mode/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-24:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f}  {r['Kernel_Name'][:50]}  grid {r.get('Grid_Size_X','')} wg {r.get('Workgroup_Size_X','')}")
PY
rm -rf gpurun_out/s4/hop_trace$mode
done
