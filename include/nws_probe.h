/*
 * nws_probe.h - C-ABI of libnws_probe.so: the probes that demonstrate the MI355X co-execution hazard the build guards against
 * (csrc/coexec_probe.hip, DESIGN.md 5.3, LABBOOK.md "5.2").  Tools and tests only.  The product library (include/nws_hip.h ->
 * libnws_hip.so) does not contain these kernels: they hold, on purpose, the packed-fp32 form its build guard refuses.
 * The reference has no counterpart (no native code at all).
 */
#ifndef NWS_PROBE_H
#define NWS_PROBE_H
#ifdef __cplusplus
extern "C" {
#endif

/* nws_coexec_pk_probe evaluates eight packed-fp32 instruction forms `iters` times per thread and adds, per form, the
 * number of results that differ from scalar arithmetic on the same operands to report[0..7] (device uint32[8], zeroed by
 * the caller; forms 4..7 are the swizzled-src1 ones).  nws_coexec_mfma_load runs a bare MFMA loop beside it:
 * kind 0 v_mfma_f32_32x32x16_f16, 1 v_mfma_f32_16x16x32_f16, 2 v_mfma_f32_32x32x8f16, 3 v_mfma_f32_32x32x2f32. */
int nws_coexec_pk_probe(int blocks, int iters, unsigned* report, void* stream);
/* same for packed fp16, v_fma_mix*, scalar-register second operands and fp64 (11 forms listed in csrc/coexec_probe.hip;
 * report: device uint32[11]) */
int nws_coexec_pk_probe2(int blocks, int iters, unsigned* report, void* stream);
/* probe 1 in waves 0-1 and a v_mfma_f32_16x16x32_f16 loop in waves 2-3 of the SAME workgroups (one kernel) */
int nws_coexec_pk_probe_mixed(int blocks, int iters, int mfma_iters, unsigned* report, float* sink, void* stream);
int nws_coexec_mfma_load(int kind, int blocks, int iters, float* sink /* device float[256] */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NWS_PROBE_H */
