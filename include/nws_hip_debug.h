/*
 * Diagnostic entry points of libnws_hip.so - NOT part of the product ABI (include/nws_hip.h): timing ablations of the hot kernels,
 * candidate sine implementations, and the probes that demonstrate the MI355X co-execution hazard the build guards against.
 * Used by tools/ and by the tests that keep them alive; outputs of ablation variants are meaningless by construction.
 * A binding of the product path never needs this file.
 */
#ifndef NWS_HIP_DEBUG_H
#define NWS_HIP_DEBUG_H
#include "nws_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Measurements / tests: which frame-MLP kernel nws_frame_mlps launches - 0 automatic (wave-resident frames from 8192 frames up,
 * tile kernels below; env NWS_MLP_KERNEL=tiles|frames), 1 the tile kernels, 2 wave-resident frames at any size. */
int nws_debug_frame_mlps_kernel(int mode);
int nws_debug_frame_mlps_probe(void* buf /* device, 4096 B: cycle timeline written by mode 2 + (6 << 8) */);

/* Diagnostics only: ablation variants of the fused kernel for timing (1 no sin, 2 no LUT gather, 3 no shaper tail,
 * 4 no MFMA; 0 = product kernel).  Outputs of variants != 0 are meaningless. */
int nws_debug_exciter_newt(int variant, const NwsWeights* w, const float* f0, const double* carry, const float* phase_u,
                           const float* rand_phase, const float* film, int B, int T, float sample_rate,
                           float* newt_out, void* stream);

/* Diagnostics only: timing ablations of control_gru_kernel (0 product; 1 half the LDS reads of h, 2 half the FMAs, 3 no
 * transcendentals in the gates, 4 no per-step barrier, 5 no LDS reads of h).  Outputs of variants != 0 are meaningless. */
int nws_debug_control_gru(int variant, const NwsWeights* w, const float* control, int B, int C, int T, float* gru_out,
                          void* stream);

/* Diagnostics only: candidate sine implementations (0 = nws_sinf as shipped, 1 = v_sin_f32 after an exact-product
 * reduction to turns, 2 = single odd polynomial after the same reduction); y[i] = sum of `reps` sines (reps = 1: sin(x[i])). */
int nws_debug_sin(int mode, const float* x, float* y, int64_t n, int reps, void* stream);

/* Diagnostics only: the MI355X co-execution hazard the build guards against (csrc/coexec_probe.hip, DESIGN.md 5.3, LABBOOK.md "5.2").
 * nws_coexec_pk_probe evaluates eight packed-fp32 instruction forms `iters` times per thread and adds, per form, the
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
#endif /* NWS_HIP_DEBUG_H */
