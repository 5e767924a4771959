# Same-box A/B of the streaming hop's launch structure, outputs compared bit for bit:
#   seven launches   NWS_STREAM_SPLIT_REVERB=0   recurrence, frame MLPs, prep, oscillator, noise, reverb partial, reduce
#   six launches     NWS_STREAM_FUSE_HEAD=0      the reverb's history parts ride on the recurrence launch
#   five launches    NWS_MLP_FEW=0               + no prep launch: head roles in the recurrence / frame-MLP launches, closing kernel hands over
#   five + few-mlp   NWS_STREAM_FUSE_MLP=0       + the frame MLPs of the hop's two frames in matrix-vector form (csrc/mlp_few.h): fp32-equal, not bit-equal
#   default (four launches)                      + those frame MLPs as workgroups of the recurrence launch (they wait for its rows)
# -> profiles/r06/stream_hop_ab.txt
export TMPDIR=/tmp
mkdir -p gpurun_out/hop
for B in 1 16; do
  for rep in 1 2; do
    NWS_AB_LABEL=seven-launch NWS_STREAM_SPLIT_REVERB=0 NWS_MLP_FEW=0 python tools/stream_hop_ab.py gpurun_out/hop/v7_$B.npy $B 2>&1 | grep "p50"
    NWS_AB_LABEL=six-launch NWS_STREAM_FUSE_HEAD=0 NWS_MLP_FEW=0 python tools/stream_hop_ab.py gpurun_out/hop/v6_$B.npy $B 2>&1 | grep "p50"
    NWS_AB_LABEL=five-launch NWS_MLP_FEW=0 python tools/stream_hop_ab.py gpurun_out/hop/v5_$B.npy $B 2>&1 | grep "p50"
    NWS_AB_LABEL=five+few-mlp NWS_STREAM_FUSE_MLP=0 python tools/stream_hop_ab.py gpurun_out/hop/v5f_$B.npy $B 2>&1 | grep "p50"
    NWS_AB_LABEL=default python tools/stream_hop_ab.py gpurun_out/hop/v4_$B.npy $B 2>&1 | grep "p50"
  done
  python - <<PY
import numpy as np
a, b, c, e, d = (np.load(f"gpurun_out/hop/v{k}_$B.npy") for k in (7, 6, 5, "5f", 4))
r = lambda x: float(np.sqrt((x.astype(np.float64) ** 2).mean()))
print("B=$B outputs", a.shape, "six == seven:", bool(np.array_equal(a, b)), " five == seven:", bool(np.array_equal(a, c)), " default (four launches) == five + few-mlp:",
      bool(np.array_equal(e, d)), " default - seven: rms", r(d - a), "of", r(a))
PY
done
rm -f gpurun_out/hop/*.npy
