export TMPDIR=/tmp
mkdir -p gpurun_out/s4
timeout 300 python -m pytest tests/test_gpu_streaming.py -x -q -p no:cacheprovider 2>&1 | tail -3
NWS_STREAM_FUSE_HEAD=0 timeout 300 python -m pytest tests/test_gpu_streaming.py -x -q -p no:cacheprovider 2>&1 | tail -2
bash tools/stream_hop_ab.sh 2>&1 | tee gpurun_out/s4/stream_hop_ab.txt
