set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/ev
timeout 200 python bench.py > gpurun_out/ev/bench_default.json 2> gpurun_out/ev/bench_default.err
timeout 400 bash tools/collect_profiles.sh r04 > gpurun_out/ev/collect.log 2>&1
ls gpurun_out/prof_r04 | head -30
