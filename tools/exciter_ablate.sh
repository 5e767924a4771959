# Where the oscillator kernel's time goes, and at which CLOCK: timing ablations of the product configuration (un-profiled), the same launches
# under rocprofv3 with GRBM_GUI_ACTIVE (kernel cycles) + the kernel trace (durations) -> effective shader clock per variant, and the cost of a
# 64-lane table gather as a function of the lines it touches (tools/ubench/gather_rate.hip).  One gpurun call -> profiles/r06/exciter_ablations.txt
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/abl
mkdir -p $O
cd $GRAFT_REPO_ROOT
echo "== timing ablations, un-profiled (tools/exciter_ablate_product.py; B = 64, T = 500; rand = the timing script's inputs, real = F0 100-1000 Hz) ==" > $O/exciter_ablations.txt
timeout 300 python tools/exciter_ablate_product.py 2>&1 | grep -v amdgpu.ids >> $O/exciter_ablations.txt
cd /tmp
KINDS=rand ROUNDS=1 timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU -d $O/pmc -- python $GRAFT_REPO_ROOT/tools/exciter_ablate_product.py > $O/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - >> $O/exciter_ablations.txt <<'PY'
import csv, glob, re, collections, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/abl/pmc")
dur, ctr = collections.defaultdict(list), collections.defaultdict(lambda: collections.defaultdict(list))
key = lambda n: re.search(r"(\w+<[^>]*>)", n).group(1)
for p in glob.glob(O + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "exciter_newt" in r["Kernel_Name"]:
            dur[key(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for p in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "exciter_newt" in r["Kernel_Name"]:
            ctr[key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {"0": "product", "1": "no sines", "2": "no table gathers", "3": "no tail", "4": "no mixer MFMAs", "5": "prologue only", "6": "no loads before the barrier",
         "7": "no FiLM-row loads", "8": "no fragment DMA", "9": "no F0 / carry / shift loads"}
print("\n== the same launches under rocprofv3 (rand inputs): duration, GRBM_GUI_ACTIVE / 8 XCDs = kernel cycles, their ratio = effective shader clock ==")
for k in sorted(dur):
    m = {n: sum(v) / len(v) for n, v in ctr[k].items()}
    w = m.get("SQ_WAVES", 1.0)
    d = sum(dur[k]) / len(dur[k])
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    dbg = k.split(",")[1].strip()
    print(f"{k:34s} {names.get(dbg, ''):28s} {d:7.1f} us  {cyc / 1e3:7.1f} k cycles  {cyc / d / 1e3:5.2f} GHz | per wave: VALU {m.get('SQ_INSTS_VALU', 0) / w:6.1f} SALU {m.get('SQ_INSTS_SALU', 0) / w:6.1f} "
          f"VALU-active {4 * m.get('SQ_ACTIVE_INST_VALU', 0) / w:7.0f} MFMA-busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / w:6.0f} wave cycles {4 * m.get('SQ_WAVE_CYCLES', 0) / w:7.0f}")
PY
echo >> $O/exciter_ablations.txt
echo "== 64-lane 8-byte gathers from an L2-resident 2 MB table against LDS (tools/ubench/gather_rate.hip; three 8-wave workgroups per CU) ==" >> $O/exciter_ablations.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_rate.hip -o /tmp/gather_rate.out 2>/dev/null && timeout 120 /tmp/gather_rate.out >> $O/exciter_ablations.txt 2>&1
rm -rf $O/pmc
cat $O/exciter_ablations.txt
