set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/ev
rm -f gpurun_out/ev/streaming_stateful.txt
python -m pytest tests -m gpu -x -q > gpurun_out/ev/pytest_gpu.txt 2>&1; tail -2 gpurun_out/ev/pytest_gpu.txt
python bench.py > gpurun_out/ev/bench_default.json 2> gpurun_out/ev/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ev/bench_driver_k20.json 2>/dev/null
python bench.py --no-cpu-baseline --exciter-opts 8 --legs 0 --pmc off > gpurun_out/ev/bench_hybrid_w_optin.json 2>/dev/null
python bench.py --no-cpu-baseline --inputs realistic --legs 0 --pmc off > gpurun_out/ev/bench_realistic_inputs.json 2>/dev/null
python bench.py --no-cpu-baseline --exact --steps 50 --legs 0 --pmc off > gpurun_out/ev/bench_exact_shapers.json 2>/dev/null
NWS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --gather rccl > gpurun_out/ev/bench_world1_rccl.json 2>/dev/null
NWS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --pmc off --legs 0 --batch1-iters 0 --gather copy > gpurun_out/ev/bench_world1_copy.json 2>/dev/null
# round 5: the N > 1 issue pattern at world size 1, same-box A/B (single / rccl / copy / sub-batches / the round-4 queued form)
bash tools/world1_check.sh > gpurun_out/ev/world1_ab.txt 2>&1
bash tools/scale_check.sh --dry-run --out gpurun_out/ev/scale_dry > gpurun_out/ev/scale_check_dry_run.txt 2>&1
bash tools/range_check.sh > gpurun_out/ev/range_proven_ab.txt 2>&1
VARIANTS=12,44,36,68 python tools/exciter_variants.py > gpurun_out/ev/exciter_variants.txt 2>&1
python tools/gru_variants.py > gpurun_out/ev/gru_variants.txt 2>&1
python scripts/time_buffer_sizes.py --use-fast-newt --checkpoint tests/golden/weights_vn.npz 2>/dev/null | grep '^buffer' > gpurun_out/ev/buffer_fast.txt
python scripts/time_buffer_sizes.py --checkpoint tests/golden/weights_vn.npz 2>/dev/null | grep '^buffer' > gpurun_out/ev/buffer_exact.txt
rm -f gpurun_out/ev/streaming.jsonl
for b in 1 16; do
  python scripts/time_streaming.py --batch-size $b --num-hops 2000 --json-out gpurun_out/ev/streaming.jsonl 2>/dev/null | grep '^stateful' >> gpurun_out/ev/streaming_stateful.txt
  python scripts/time_streaming.py --batch-size $b --num-hops 2000 --static-io --json-out gpurun_out/ev/streaming.jsonl 2>/dev/null | grep '^stateful' >> gpurun_out/ev/streaming_stateful.txt
  python scripts/time_streaming.py --batch-size $b --num-hops 500 --no-graph --json-out gpurun_out/ev/streaming.jsonl 2>/dev/null | grep '^stateful' >> gpurun_out/ev/streaming_stateful.txt
done
# round 4: the reverb at every length, the two frame-MLP kernel families (+ ablations, cycle timeline), the runtime-size path
python tools/reverb_lengths.py > gpurun_out/ev/reverb_lengths.txt 2>&1
MODES=1,2 python tools/mlp_variants.py 64 500 48 500 32 500 128 500 > gpurun_out/ev/mlp_variants.txt 2>&1
python tools/mlp_timeline.py > gpurun_out/ev/mlp_timeline.txt 2>&1
python tools/generic_profile.py > gpurun_out/ev/generic_path.txt 2>&1
bash tools/generic_kernels.sh 64 500 > gpurun_out/ev/generic_kernels.txt 2>&1
# round 6: queue placement by measurement, the 8-rank queue population on one GPU, CU pressure, the two measured-as-nothing kernel items
python tools/queue_pipe_map.py --layout nnnnhhhh --repeat 2 > gpurun_out/ev/queue_pipe_map_final.txt 2>&1
bash tools/placement_ab.sh > /dev/null 2>&1; cp gpurun_out/placement_ab.txt gpurun_out/ev/placement_ab.txt
bash tools/fake_peers_ab.sh > /dev/null 2>&1; cp gpurun_out/fake_peers_ab.txt gpurun_out/ev/fake_peers_ab.txt
python tools/cu_pressure.py > gpurun_out/ev/cu_pressure.txt 2>&1
python tools/film_dma_ab.py > gpurun_out/ev/film_dma_ab.txt 2>&1
python tools/mlp_paths_ab.py > gpurun_out/ev/mlp_paths_ab.txt 2>&1
# round 6, last session: the streaming hop's launch structures (bit-identity + p50 per form), the few-frame MLP kernel's cycle timeline
bash tools/stream_hop_ab.sh 2>&1 | grep -E "p50|outputs" > gpurun_out/ev/stream_hop_ab.txt
python tools/mlp_few_timeline.py 2>&1 | grep -E "^rep [345]" > gpurun_out/ev/mlp_few_timeline.txt
bash tools/collect_profiles.sh ${ROUND:-r06} > gpurun_out/ev/collect.log 2>&1
ls gpurun_out/prof_${ROUND:-r06} | head -30
du -sh gpurun_out
