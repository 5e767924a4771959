set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/ev
python -m pytest tests -m gpu -x -q > gpurun_out/ev/pytest_gpu.txt 2>&1; tail -2 gpurun_out/ev/pytest_gpu.txt
python bench.py > gpurun_out/ev/bench_default.json 2> gpurun_out/ev/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ev/bench_driver_k20.json 2>/dev/null
python bench.py --no-cpu-baseline --exciter-opts 0 > gpurun_out/ev/bench_two_term.json 2>/dev/null
python bench.py --no-cpu-baseline --inputs realistic > gpurun_out/ev/bench_realistic_inputs.json 2>/dev/null
python bench.py --no-cpu-baseline --exact --steps 50 > gpurun_out/ev/bench_exact_shapers.json 2>/dev/null
NWS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --gather rccl > gpurun_out/ev/bench_world1_rccl.json 2>/dev/null
NWS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --gather copy > gpurun_out/ev/bench_world1_copy.json 2>/dev/null
VARIANTS=12,20,36 python tools/exciter_variants.py > gpurun_out/ev/exciter_variants.txt 2>&1
python tools/gru_variants.py > gpurun_out/ev/gru_variants.txt 2>&1
python scripts/time_buffer_sizes.py --use-fast-newt --checkpoint tests/golden/weights_vn.npz 2>/dev/null | grep '^buffer' > gpurun_out/ev/buffer_fast.txt
python scripts/time_buffer_sizes.py --checkpoint tests/golden/weights_vn.npz 2>/dev/null | grep '^buffer' > gpurun_out/ev/buffer_exact.txt
python scripts/time_streaming.py 2>/dev/null | grep '^stateful' > gpurun_out/ev/streaming_stateful.txt
bash tools/collect_profiles.sh r02 > gpurun_out/ev/collect.log 2>&1
ls gpurun_out/prof_r02 | head -30
du -sh gpurun_out
