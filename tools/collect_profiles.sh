#!/bin/bash
# Round evidence set, run ON THE GPU BOX from the repo root:  bash tools/collect_profiles.sh r01
# Writes gpurun_out/prof_<round>/ (kernel traces with --stats, separate --pmc passes, bench lines, parity report);
# tools/pmc_digest.py turns that into the files committed under profiles/<round>/.
set -u
ROUND=${1:-r01}
R=$PWD
OUT=$R/gpurun_out/prof_$ROUND
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --no-cpu-baseline --batch1-iters 0"
# 1. the default bench line (with cpu baseline, batch-1 and streaming extras)
(cd $R && python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err")
# 2. kernel traces + stats: default issue (pipeline, 2 audio streams) and one stream, no pipeline (undisturbed kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -o bench -- $BENCH --steps 200 --warmup 20 > "$OUT/trace_default.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1stream" -o bench -- $BENCH --steps 200 --warmup 20 --pipeline 0 --streams 1 > "$OUT/trace_1stream.log" 2>&1
# 3. counters, one stream, separate passes (never together with trace domains other than the kernel trace)
P="$BENCH --steps 5 --warmup 2 --pipeline 0 --streams 1"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -o bench -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_mfma" -o bench -- $P > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o bench -- $P > /dev/null 2>&1
cd $R
python tools/pmc_digest.py "$OUT" > "$OUT/digest.log" 2>&1
tail -30 "$OUT/digest.log"
# 4. the co-execution hazard matrix (csrc/coexec_probe.hip)
python tools/coexec_probe.py --json "$OUT/coexec_matrix.json" > "$OUT/coexec_matrix.txt" 2>&1
tail -12 "$OUT/coexec_matrix.txt"
# 5. parity report of the GPU suite
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1; tail -3 "$OUT/pytest_gpu.log"
cp gpurun_out/parity_report.json "$OUT/parity_report.json" 2>/dev/null
# 6. other bench configurations (one line each) and a long bit-exact soak of the default pipeline
(python bench.py --no-cpu-baseline --inputs realistic | grep '^{' > "$OUT/bench_realistic_inputs.json") 2>/dev/null
(python bench.py --no-cpu-baseline --exact | grep '^{' > "$OUT/bench_exact_shapers.json") 2>/dev/null
(NWS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline | grep '^{' > "$OUT/bench_world1_rccl.json") 2>/dev/null
python tools/soak_pipeline.py --rounds 250 > "$OUT/soak_pipeline.json" 2>/dev/null; cat "$OUT/soak_pipeline.json"
# 7. exact-shaper mode, one stream: kernel stats (the shaper-bank kernel)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_exact" -o bench -- $BENCH --exact --steps 50 --warmup 5 --pipeline 0 --streams 1 > "$OUT/trace_exact.log" 2>&1
cd $R
cp $(find "$OUT/trace_exact" -name "*kernel_stats.csv" | head -1) "$OUT/rocprofv3_kernel_stats_exact_1stream.csv"
head -4 "$OUT/rocprofv3_kernel_stats_exact_1stream.csv"
