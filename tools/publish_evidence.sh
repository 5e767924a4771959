#!/bin/bash
# Copy what tools/evidence_run.sh left under gpurun_out/ (merged back from the GPU box) into profiles/<round>/ (tracked).
R=${1:-r06}
D=profiles/$R
mkdir -p "$D"
for f in bench_default bench_hybrid_w_optin bench_realistic_inputs bench_exact_shapers bench_world1_rccl bench_world1_copy bench_driver_k20; do
  tail -1 gpurun_out/ev/$f.json > "$D/$f.json"
done
cp gpurun_out/ev/exciter_variants.txt gpurun_out/ev/gru_variants.txt "$D/"
cp gpurun_out/parity_report.json "$D/parity_report.json"
cp gpurun_out/prof_$R/pmc_kernels.json gpurun_out/prof_$R/pmc_digest.txt gpurun_out/prof_$R/pmc_traffic.json gpurun_out/prof_$R/rocprofv3_summary.txt "$D/"
cp gpurun_out/prof_$R/rocprofv3_kernel_stats_1stream.csv "$D/rocprofv3_kernel_stats_1stream.csv"
cp gpurun_out/prof_$R/rocprofv3_kernel_stats_default.csv "$D/rocprofv3_kernel_stats_default_pipeline.csv"
grep -E "passed|failed" gpurun_out/ev/pytest_gpu.txt | tail -1 > "$D/pytest_gpu_summary.txt"
cat gpurun_out/ev/buffer_fast.txt gpurun_out/ev/buffer_exact.txt gpurun_out/ev/streaming_stateful.txt > "$D/buffer_sizes_summary.txt"
cp gpurun_out/ev/streaming.jsonl "$D/streaming_stateful.jsonl"
python tools/buffer_sizes_digest.py gpurun_out/ev/buffer_fast.txt gpurun_out/ev/buffer_exact.txt > "$D/buffer_sizes.csv"
for f in reverb_lengths mlp_variants mlp_timeline generic_path generic_kernels world1_ab scale_check_dry_run range_proven_ab queue_pipe_map_final placement_ab fake_peers_ab cu_pressure film_dma_ab mlp_paths_ab stream_hop_ab mlp_few_timeline; do
  [ -f gpurun_out/ev/$f.txt ] && grep -v "amdgpu.ids" gpurun_out/ev/$f.txt > "$D/$f.txt"
done
