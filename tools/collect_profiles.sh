#!/bin/bash
# Collect the round's rocprofv3 evidence ON THE GPU BOX:  bash tools/collect_profiles.sh r02
#   gpurun_out/prof_<round>/trace_default   --kernel-trace --stats of the default bench command (pipelined issue)
#   gpurun_out/prof_<round>/trace_1stream   same, --pipeline 0 --streams 1 (undisturbed per-kernel durations)
#   gpurun_out/prof_<round>/pmc_*           counter passes (separate runs: SQ has 8 slots, FETCH_SIZE and WRITE_SIZE do not
#                                           fit one pass; never combined with sys/hip/hsa traces)
# then tools/pmc_digest.py condenses them; copy the digest files into profiles/<round>/ and commit.
R=${1:-r04}
OUT=gpurun_out/prof_$R
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python bench.py --no-cpu-baseline --batch1-iters 0 --warmup 5 --legs 0 --pmc off ${NWS_PROFILE_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- $BENCH --steps 50 > "$OUT/trace_default.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_1stream" -- $BENCH --steps 30 --pipeline 0 --streams 1 > "$OUT/trace_1stream.log" 2>&1
P="$BENCH --steps 6 --pipeline 0 --streams 1"
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  -d "$OUT/pmc_sq" -- $P > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  -d "$OUT/pmc_mfma" -- $P > "$OUT/pmc_mfma.log" 2>&1
rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_l2" -- $P > "$OUT/pmc_l2.log" 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -- $P > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/pmc_write" -- $P > "$OUT/pmc_write.log" 2>&1
python tools/pmc_digest.py "$OUT" > "$OUT/digest.log" 2>&1
python tools/rocprof_summary.py "$OUT/rocprofv3_kernel_stats_1stream.csv" "$OUT/rocprofv3_kernel_stats_default.csv" > "$OUT/rocprofv3_summary.txt" 2>> "$OUT/digest.log"
# keep the merge-back small: the raw per-dispatch csv files are dropped, the digests stay
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -size +8M -delete
ls -la "$OUT"
