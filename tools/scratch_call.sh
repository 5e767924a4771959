export TMPDIR=/tmp
for B in 1; do
NWS_AB_EAGER=1 NWS_AB_LABEL=default python tools/stream_hop_ab.py /tmp/x.npy $B 2>&1 | grep "p50"
NWS_BACKEND=ctypes NWS_AB_EAGER=1 NWS_AB_LABEL=ctypes python tools/stream_hop_ab.py /tmp/x.npy $B 2>&1 | grep "p50"
done
timeout 900 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_coexec.py -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed"
