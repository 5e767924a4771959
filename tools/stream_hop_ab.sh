# Same-box A/B of the streaming hop's launch structure, outputs compared bit for bit:
#   seven launches   NWS_STREAM_SPLIT_REVERB=0   recurrence, frame MLPs, prep, oscillator, noise, reverb partial, reduce
#   six launches     NWS_STREAM_FUSE_HEAD=0      the reverb's history parts ride on the recurrence launch
#   five launches    (default)                   + no prep launch: head roles in the recurrence / frame-MLP launches, closing kernel hands over
# -> profiles/r06/stream_hop_ab.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/hop
for B in 1 16; do
  for rep in 1 2; do
    NWS_AB_LABEL=seven-launch NWS_STREAM_SPLIT_REVERB=0 python tools/stream_hop_ab.py gpurun_out/hop/v7_$B.npy $B 2>&1 | grep "p50"
    NWS_AB_LABEL=six-launch NWS_STREAM_FUSE_HEAD=0 python tools/stream_hop_ab.py gpurun_out/hop/v6_$B.npy $B 2>&1 | grep "p50"
    NWS_AB_LABEL=five-launch python tools/stream_hop_ab.py gpurun_out/hop/v5_$B.npy $B 2>&1 | grep "p50"
  done
  python - <<PY
import numpy as np
a, b, c = (np.load(f"gpurun_out/hop/v{k}_$B.npy") for k in (7, 6, 5))
print("B=$B outputs", a.shape, "six == seven:", bool(np.array_equal(a, b)), " five == seven:", bool(np.array_equal(a, c)), " max |d|", float(np.abs(a - c).max()),
      "rms", float(np.sqrt((a.astype(np.float64) ** 2).mean())))
PY
done
rm -f gpurun_out/hop/*.npy
