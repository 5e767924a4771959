#!/usr/bin/env python
"""Condense gpurun_out/prof_<round>/ (tools/collect_profiles.sh) into the small files kept under profiles/<round>/:
   rocprofv3_kernel_stats_*.csv (copied), pmc_digest.txt, pmc_traffic.json, rocprofv3_summary.txt."""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:40] if name.startswith("exciter_newt_kernel<") else name[:24]


def counters(root, sub):
    agg = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in agg.items()}


def main(root):
    out = {}
    for sub in ("pmc_sq", "pmc_mfma", "pmc_l2", "pmc_fetch", "pmc_write"):
        for k, c in counters(root, sub).items():
            out.setdefault(k, {}).update(c)
    lines = ["# rocprofv3 --pmc digest, per-dispatch averages at B=64, T=500, one stream, whole forwards (separate passes: "
             "pmc_sq, pmc_mfma, pmc_l2, pmc_fetch, pmc_write).",
             "# SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are quad-cycles summed over waves, except SQ_VALU_MFMA_BUSY_CYCLES (cycles).",
             "# FETCH_SIZE / WRITE_SIZE in KB as reported (gfx950 x2 correction for wide coalesced reads NOT applied here)."]
    for k in sorted(out):
        m = out[k]
        w = m.get("SQ_WAVES", 0) or 1
        g = lambda n: m.get(n, float("nan"))  # noqa: E731
        wc = g("SQ_WAVE_CYCLES") or 1
        lines.append(
            f"{k:24s} waves {w:7.0f} | per wave: VALU {g('SQ_INSTS_VALU') / w:6.0f} MFMA {g('SQ_INSTS_MFMA') / w:4.0f} "
            f"TRANS {g('SQ_INSTS_VALU_TRANS_F32') / w:5.0f} wave-quad-cycles {wc / w:8.0f} | of wave-cycles: valu "
            f"{g('SQ_ACTIVE_INST_VALU') / wc:.2f} lds {g('SQ_ACTIVE_INST_LDS') / wc:.2f} wait_any {g('SQ_WAIT_ANY') / wc:.2f} "
            f"wait_inst {g('SQ_WAIT_INST_ANY') / wc:.2f} | MFMA busy cyc/wave {g('SQ_VALU_MFMA_BUSY_CYCLES') / w:7.0f} | "
            f"MOPS f16 {g('SQ_INSTS_VALU_MFMA_MOPS_F16'):.3g} f32 {g('SQ_INSTS_VALU_MFMA_MOPS_F32'):.3g} | LDS bank-conflict/"
            f"active {g('SQ_LDS_BANK_CONFLICT') / (g('SQ_LDS_IDX_ACTIVE') or 1):.3f} | L2 hit "
            f"{g('TCC_HIT_sum') / ((g('TCC_HIT_sum') + g('TCC_MISS_sum')) or 1):.3f} | FETCH_KB {g('FETCH_SIZE'):8.0f} "
            f"WRITE_KB {g('WRITE_SIZE'):8.0f} | GUI_ACTIVE {g('GRBM_GUI_ACTIVE'):.3g}")
    open(os.path.join(root, "pmc_digest.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    # per-kernel record for bench.py's roofline / roofline_all (profiles/<round>/pmc_kernels.json)
    stats = {}
    for path in glob.glob(os.path.join(root, "trace_1stream", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            stats[short(r["Name"])] = float(r["AverageNs"])
    kernels = {}
    opt_bits = None
    for k in out:
        mm = re.match(r"exciter_newt_kernel<4, 0, 2, (\d+)", k)
        if mm:
            opt_bits = int(mm.group(1))
    # template OPT bits of the hot-path kernel -> NwsWeights.exciter_opts value that selects it (csrc/exciter_newt.hip)
    eff_opts = None if opt_bits is None else (8 if opt_bits & 16 else 4 if opt_bits & 8 else 2 if opt_bits & 4 else 0 if opt_bits & 2 else 1)
    for k, m in out.items():
        w = m.get("SQ_WAVES", 0) or 1
        name = re.sub(r"<.*", "", k)
        e = {"waves": w, "valu_insts_per_wave": m.get("SQ_INSTS_VALU", 0) / w, "mfma_insts_per_wave": m.get("SQ_INSTS_MFMA", 0) / w,
             "trans_insts_per_wave": m.get("SQ_INSTS_VALU_TRANS_F32", 0) / w,
             "valu_active_quad_cycles_per_wave": m.get("SQ_ACTIVE_INST_VALU", 0) / w,
             "wave_quad_cycles_per_wave": m.get("SQ_WAVE_CYCLES", 0) / w,
             "mfma_busy_cycles_per_wave": m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / w,
             "wait_any_frac_of_wave_cycles": m.get("SQ_WAIT_ANY", 0) / (m.get("SQ_WAVE_CYCLES", 0) or 1),
             "wait_inst_frac_of_wave_cycles": m.get("SQ_WAIT_INST_ANY", 0) / (m.get("SQ_WAVE_CYCLES", 0) or 1)}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["fetch_kb"], e["write_kb"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
            e["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0   # gfx950: FETCH_SIZE x2 (MICROARCH guide)
        ns, gui = stats.get(k), m.get("GRBM_GUI_ACTIVE")
        if ns and gui:
            if gui / ns > 4.0:
                gui /= 8.0
            e.update(kernel_avg_ns_one_stream=ns)
            if 1.0 <= gui / ns <= 2.6:       # a plausible shader clock: GRBM_GUI_ACTIVE of short kernels includes idle gaps
                e.update(kernel_cycles=gui, clock_ghz_during_pass=gui / ns,
                         valu_busy_frac=w * e["valu_active_quad_cycles_per_wave"] * 4.0 / (1024.0 * gui))
        kernels[name] = e
    json.dump({"source": "rocprofv3 --pmc, separate passes, one stream, whole forwards (tools/collect_profiles.sh, tools/pmc_digest.py)",
               "batch_per_gpu": int(os.environ.get("NWS_PROFILE_BATCH", 64)), "frames": int(os.environ.get("NWS_PROFILE_FRAMES", 500)),
               "exciter_opts": eff_opts if eff_opts is not None else int(os.environ.get("NWS_EXCITER_OPTS", 0)),
               "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM "
                             "section); WRITE_SIZE as reported", "kernels": kernels},
              open(os.path.join(root, "pmc_kernels.json"), "w"), indent=1)
    ex = next((v for k, v in out.items() if k.startswith("exciter_newt_kernel")), None)
    if ex and "FETCH_SIZE" in ex and "WRITE_SIZE" in ex:
        w = ex.get("SQ_WAVES", 0) or 1
        traffic = {
            "source": "pmc_digest.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, B=64 T=500)",
            "kernel": "exciter_newt_kernel", "batch_per_gpu": 64, "frames": 500,
            "fetch_kb": ex["FETCH_SIZE"], "write_kb": ex["WRITE_SIZE"],
            "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "
                          "§HBM); WRITE_SIZE as reported",
            "hbm_bytes_per_launch": (2.0 * ex["FETCH_SIZE"] + ex["WRITE_SIZE"]) * 1024.0,
            "valu_insts_per_wave": ex.get("SQ_INSTS_VALU", 0) / w, "mfma_insts_per_wave": ex.get("SQ_INSTS_MFMA", 0) / w,
            "trans_insts_per_wave": ex.get("SQ_INSTS_VALU_TRANS_F32", 0) / w, "waves": w,
            "valu_active_quad_cycles_per_wave": ex.get("SQ_ACTIVE_INST_VALU", 0) / w,
            "wave_quad_cycles_per_wave": ex.get("SQ_WAVE_CYCLES", 0) / w,
            "gui_active_cycles": ex.get("GRBM_GUI_ACTIVE"),
        }
        # VALU-busy share of the kernel: VALU-active cycles summed over waves / (1024 SIMDs x kernel cycles).  Kernel cycles
        # from GRBM_GUI_ACTIVE; rocprofv3 reports it summed over the 8 XCDs in some passes: normalise with the trace's duration.
        ns = None
        for path in glob.glob(os.path.join(root, "trace_1stream", "**", "*kernel_stats.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                if "exciter_newt_kernel" in r["Name"]:
                    ns = float(r["AverageNs"])
        gui = ex.get("GRBM_GUI_ACTIVE")
        if ns and gui:
            if gui / ns > 4.0:          # > 4 GHz: the counter is a sum over XCDs
                gui /= 8.0
            traffic["kernel_avg_ns_one_stream"] = ns
            traffic["kernel_cycles"] = gui
            traffic["clock_ghz_during_pass"] = gui / ns
            traffic["valu_busy_frac"] = w * traffic["valu_active_quad_cycles_per_wave"] * 4.0 / (1024.0 * gui)
        json.dump(traffic, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
    for tag in ("default", "1stream"):
        for path in glob.glob(os.path.join(root, "trace_" + tag, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(path, os.path.join(root, f"rocprofv3_kernel_stats_{tag}.csv"))


if __name__ == "__main__":
    main(sys.argv[1])
