#!/usr/bin/env python
"""Print the MI355X co-execution hazard matrix (csrc/coexec_probe.hip, DESIGN.md 5.3, LABBOOK.md '5.2').

A probe kernel evaluates eight packed-fp32 instruction forms against scalar arithmetic while, on a second stream, either
nothing, a bare MFMA loop of one flavour, or one of the product's own MFMA kernels runs.  Columns = wrong results per form
(out of blocks * 256 * iters * rounds evaluations each).

    python tools/coexec_probe.py [--json out.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

FORMS = ["pk_fma plain", "pk_add src0 swap", "pk_add src1 lo-bcast", "pk_fma src2 swap+neg",
         "pk_add src1 swap", "pk_add src1 hi-bcast", "pk_mul src1 swap", "pk_fma src1 swap"]
FORMS2 = ["pk_add_f16 src1 swap", "pk_fma_f16 src1 swap", "pk_mul_f16 src1 hi-bcast", "fma_mix_f32 src1 hi", "fma_mixlo_f16 src1 hi",
          "pk_add_f32 SGPR hi-bcast", "pk_mul_f32 SGPR swap", "pk_add_f16 plain", "v_add_f64", "v_mul_f64", "v_fma_f64"]
LOADS = {"none": None, "v_mfma_f32_32x32x16_f16": 0, "v_mfma_f32_16x16x32_f16": 1, "v_mfma_f32_32x32x8f16": 2,
         "v_mfma_f32_32x32x2f32": 3}


def run(blocks=4096, iters=2000, rounds=5, product_kernels=True):
    import torch
    import nws_amd
    _lib = nws_amd._lib
    L = _lib.probe_lib()         # libnws_probe.so (include/nws_probe.h): the hazard kernels are not part of the product library
    s_probe, s_load = torch.cuda.Stream(), torch.cuda.Stream()
    report = torch.zeros(16, dtype=torch.int32, device="cuda")
    sink = torch.zeros(256, device="cuda")
    loads = {k: ((lambda kind=v: _lib.check(L.nws_coexec_mfma_load(kind, 8192, 3000, sink.data_ptr(), s_load.cuda_stream), "load"))
                 if v is not None else (lambda: None)) for k, v in LOADS.items()}
    if product_kernels:
        from gpu_util import build_model
        m = build_model(True)
        eng = m._engine
        B, T = 64, 500
        g = torch.Generator(device="cuda").manual_seed(0)
        c = torch.randn(B, 2, T, device="cuda", generator=g)
        nz = torch.rand(128 * T - 1, device="cuda", generator=g)
        gru = eng.control_gru(c)
        _, film, _, fir = eng.frame_mlps(gru)
        newt = torch.zeros(B, 128 * T, device="cuda")

        def with_stream(fn):
            def go():
                with torch.cuda.stream(s_load):
                    fn()
            return go
        loads["product: frame_mlps16_kernel x3"] = with_stream(lambda: [eng.frame_mlps(gru) for _ in range(3)])
        loads["product: fir_noise_mfma_kernel x4"] = with_stream(lambda: [eng.fir_noise(fir, nz, add_in=newt) for _ in range(4)])
        loads["product: control_gru_kernel"] = with_stream(lambda: eng.control_gru(c))
    out, out2 = {}, {}
    for probe, dst in ((L.nws_coexec_pk_probe, out), (L.nws_coexec_pk_probe2, out2)):
        for name, load in loads.items():
            report.zero_()
            torch.cuda.synchronize()
            for _ in range(rounds):
                _lib.check(probe(blocks, iters, report.data_ptr(), s_probe.cuda_stream), "probe")
                load()
                torch.cuda.synchronize()
            dst[name] = [int(v) for v in report.cpu().numpy().astype("uint32")[:8 if dst is out else 11]]
    # one kernel: probe in waves 0-1, v_mfma_f32_16x16x32_f16 loop in waves 2-3 of the same workgroups (half the probe lanes)
    report.zero_()
    torch.cuda.synchronize()
    for _ in range(rounds):
        _lib.check(L.nws_coexec_pk_probe_mixed(blocks, iters, 10 * iters, report.data_ptr(), sink.data_ptr(),
                                               s_probe.cuda_stream), "mixed")
    torch.cuda.synchronize()
    out["same kernel: MFMA in waves 2-3 (x0.5 evals)"] = [int(v) for v in report.cpu().numpy().astype("uint32")[:8]]
    return {"forms": FORMS, "evaluations_per_form": blocks * 256 * iters * rounds, "wrong_results": out,
            "forms_other_families": FORMS2, "wrong_results_other_families": out2}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    a = ap.parse_args()
    r = run()
    print(f"evaluations per form and row: {r['evaluations_per_form']:.3g}")
    print(f"{'running beside':36s} " + " ".join(f"{f[:12]:>12s}" for f in r["forms"]))
    for k, v in r["wrong_results"].items():
        print(f"{k[:44]:44s} " + " ".join(f"{x:12d}" for x in v))
    print(f"{'other families, running beside':36s} " + " ".join(f"{f[:12]:>12s}" for f in r["forms_other_families"]))
    for k, v in r["wrong_results_other_families"].items():
        print(f"{k:36s} " + " ".join(f"{x:12d}" for x in v))
    if a.json:
        json.dump(r, open(a.json, "w"), indent=1)
