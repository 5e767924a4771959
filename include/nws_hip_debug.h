/*
 * Diagnostic entry points of libnws_hip.so - NOT part of the product ABI (include/nws_hip.h): timing ablations of the hot kernels
 * and candidate sine implementations.  (The probes that demonstrate the MI355X co-execution hazard live in a library of their
 * own, include/nws_probe.h -> libnws_probe.so: their kernels contain the instruction form the build refuses in this one.)
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

/* Diagnostics only (tools/cu_pressure.py): a resident load of `groups` 256-thread workgroups that keep their CUs' vector pipes
 * busy for `spin_us` microseconds of the wall clock - what a collective's ring kernels take from the oscillator kernel. */
int nws_debug_queue_busy(int groups, int spin_us, float* sink /* device float[256] */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NWS_HIP_DEBUG_H */
