export TMPDIR=/tmp
mkdir -p gpurun_out/s4
for i in 1 2; do
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s4/pytest_gpu_g$i.txt 2>&1; grep -E "passed|failed" gpurun_out/s4/pytest_gpu_g$i.txt | tail -1
done
