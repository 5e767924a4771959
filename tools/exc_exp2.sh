export TMPDIR=/tmp
mkdir -p gpurun_out/x5
cd /tmp
KINDS=rand ONLY=44,50 ROUNDS=1 timeout 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/x5/pmc_a -- python $GRAFT_REPO_ROOT/tools/exciter_ablate_product.py > $GRAFT_REPO_ROOT/gpurun_out/x5/a.log 2>&1
KINDS=rand ONLY=44,50 ROUNDS=1 timeout 120 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $GRAFT_REPO_ROOT/gpurun_out/x5/pmc_b -- python $GRAFT_REPO_ROOT/tools/exciter_ablate_product.py > $GRAFT_REPO_ROOT/gpurun_out/x5/b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_by_variant.py gpurun_out/x5/pmc_a exciter_newt > gpurun_out/x5/digest_a.txt 2>&1
python tools/pmc_by_variant.py gpurun_out/x5/pmc_b exciter_newt > gpurun_out/x5/digest_b.txt 2>&1
cat gpurun_out/x5/digest_a.txt gpurun_out/x5/digest_b.txt
tail -3 gpurun_out/x5/b.log
find gpurun_out/x5 -name "*.db" -delete; find gpurun_out/x5 -name "*counter_collection.csv" -delete
