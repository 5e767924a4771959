# Same-box A/B of the streaming hop's launch structure: three-launch reverb (NWS_STREAM_SPLIT_REVERB=0) against the history
# parts riding on the recurrence launch (default); outputs compared bit for bit.  -> profiles/r06/stream_hop_ab.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/hop
for B in 1 16; do
  for rep in 1 2; do
    NWS_AB_LABEL=three-launch NWS_STREAM_SPLIT_REVERB=0 python tools/stream_hop_ab.py gpurun_out/hop/old_$B.npy $B 2>&1 | grep -v Warning
    NWS_AB_LABEL=split-reverb python tools/stream_hop_ab.py gpurun_out/hop/new_$B.npy $B 2>&1 | grep -v Warning
  done
  python - <<PY
import numpy as np
a, b = np.load("gpurun_out/hop/old_$B.npy"), np.load("gpurun_out/hop/new_$B.npy")
print("B=$B outputs", a.shape, "bit-identical:", bool(np.array_equal(a, b)), "max |d|", float(np.abs(a - b).max()), "rms", float(np.sqrt((a.astype(np.float64) ** 2).mean())))
PY
done
rm -f gpurun_out/hop/*.npy
