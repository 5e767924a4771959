# Differential run of the streaming step's launch structures over mixed chunk sequences (hops of one / two frames between longer
# chunks, first and final chunks of every size; 1 .. 17 streams): the emitted samples must be bit-identical between structures of
# equal frame-MLP arithmetic - seven launches against five (tile kernel both), five against four (matrix-vector form both).
export TMPDIR=/tmp
mkdir -p gpurun_out/hop
i=0
for spec in "1:2,2,1,2,7,2,2,1,1,2,3,2,2" "2:1,1,2,2,2,16,2,1,2,2" "5:2,2,2,33,1,2,2,2,1" "16:2,1,2,2,5,2,2" "17:1,2,2,2,4,2,1,2" "3:2" "3:1,1" "2:1,2"; do
  B=${spec%%:*}; C=${spec#*:}; i=$((i+1))
  NWS_AB_CHUNKS=$C NWS_STREAM_SPLIT_REVERB=0 NWS_MLP_FEW=0 python tools/stream_hop_ab.py gpurun_out/hop/d7.npy $B dump-only > /dev/null 2>&1
  NWS_AB_CHUNKS=$C NWS_MLP_FEW=0 python tools/stream_hop_ab.py gpurun_out/hop/d5.npy $B dump-only > /dev/null 2>&1
  NWS_AB_CHUNKS=$C NWS_STREAM_FUSE_MLP=0 python tools/stream_hop_ab.py gpurun_out/hop/d5f.npy $B dump-only > /dev/null 2>&1
  NWS_AB_CHUNKS=$C python tools/stream_hop_ab.py gpurun_out/hop/d4.npy $B dump-only > /dev/null 2>&1
  python - <<PY
import numpy as np
a, b, c, d = (np.load(f"gpurun_out/hop/d{k}.npy") for k in (7, 5, "5f", 4))
r = lambda x: float(np.sqrt((x.astype(np.float64) ** 2).mean()))
print("B=$B chunks $C:", a.shape, "five == seven:", bool(np.array_equal(a, b)), " four == five (matrix-vector MLPs):", bool(np.array_equal(c, d)),
      " finite:", bool(np.isfinite(d).all()), " four - seven rms", "%.2e" % r(d - a), "of", "%.3f" % r(a))
PY
done
rm -f gpurun_out/hop/d*.npy
