// Shared device helpers for the gfx950 NEWT kernels.
//
// The library is compiled with -ffp-contract=off: every a*b+c written as plain arithmetic is two
// IEEE roundings (the reference's CPU chains for the oscillator phase and the LUT index must be
// reproduced rounding for rounding); fused multiply-adds are spelled
// fmaf() wherever they are wanted.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nws_hip.h"
#include "../../include/nws_hip_debug.h"

#define NWS_WAVE 64

typedef float f32x2 __attribute__((ext_vector_type(2)));  // one v_pk_*_f32 operand
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NWS_CHECK_LAUNCH()                      \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

// One-time-per-DEVICE guard for hipFuncSetAttribute (the attribute belongs to the kernel's image on ONE device: a flag per
// process would leave a second GPU of the same process without it).  `mask` is a static per call site; returns true the
// first time the calling thread's current device is seen.  (A benign race repeats the call, nothing worse.)
inline bool nws_first_use_on_device(unsigned long long& mask) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
  if ((mask >> d) & 1ull) return false;
  mask |= 1ull << d;
  return true;
}

// ---------------------------------------------------------------------------------------------
// F.upsample(x, T*128, mode="linear") of the reference (align_corners=False), bit-exact with
// ATen's CPU kernel as probed in the build container:  out = fmaf(1-l, x[i0], fl(l*x[i1])).
// (models/neural_waveshaping.py:75, models/modules/shaping.py:69.)
// ---------------------------------------------------------------------------------------------
struct NwsLerp {
  int i0, i1;
  float w0, w1;
};

__device__ __forceinline__ NwsLerp nws_lerp_coeff(int n, int T) {
  // scale = T/N = 1/128 exactly; (n + 0.5)/128 - 0.5 is exact in fp32 for n < 2^23
  float src = ((float)n + 0.5f) * (1.0f / (float)NWS_HOP) - 0.5f;
  src = src < 0.0f ? 0.0f : src;
  const int i0 = (int)src;  // src >= 0: trunc == floor
  NwsLerp c;
  c.i0 = i0;
  c.i1 = i0 + 1 < T ? i0 + 1 : T - 1;
  c.w1 = src - (float)i0;
  c.w0 = 1.0f - c.w1;
  return c;
}

__device__ __forceinline__ float nws_lerp(float a, float b, float w0, float w1) { return fmaf(w0, a, w1 * b); }

// ---------------------------------------------------------------------------------------------
// sinf with full-range argument reduction.  |error| <~ 1.5e-7 absolute for |x| <= 6e6 (two-constant
// Cody-Waite in fp32 with FMA: the product q*P1 is exact inside the fma), fp64 reduction above.
// The reference's torch.sin (Sleef u10) is itself within 1 ulp of the true value; the oscillator
// needs ~1e-6 (SURVEY.md §7 hard part 2).  Used where accuracy matters most (FastNEWT table construction,
// stand-alone shapers, |x| > 6e6); the oscillator uses nws_sinf_fast below.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float nws_sin_poly(float r) {
  const float z = r * r;
  float p = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
  p = fmaf(p, z, -1.6666654611e-1f);
  return fmaf(p * z, r, r);
}

__device__ __forceinline__ float nws_cos_poly(float r) {
  const float z = r * r;
  float p = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
  p = fmaf(p, z, 4.166664568298827e-2f);
  return fmaf(p, z * z, fmaf(-0.5f, z, 1.0f));
}

__device__ __noinline__ float nws_sinf_huge(float x) {
  // rare path: |x| > 6e6 (sine arguments this large are already phase-quantised in fp32)
  return (float)sin((double)x);
}

__device__ __forceinline__ float nws_sinf(float x) {
  if (__builtin_expect(fabsf(x) > 6.0e6f, 0)) return nws_sinf_huge(x);
  const float fq = rintf(x * 0.6366197723675814f);
  float r = fmaf(fq, -1.5707963705062866f, x);
  r = fmaf(fq, 4.371138828673793e-08f, r);
  int q = (int)fq;
  if (fabsf(x) > 32768.0f) {
    // the fp32 quotient estimate is only good to ~1e-7*|x|: r may sit up to ~0.5 outside [-pi/4, pi/4];
    // one more (exact) reduction step by -1/0/+1 quadrants brings it back
    const float fq2 = rintf(r * 0.6366197723675814f);
    r = fmaf(fq2, -1.5707963705062866f, r);
    r = fmaf(fq2, 4.371138828673793e-08f, r);
    q += (int)fq2;
  }
  const float s = nws_sin_poly(r);
  const float c = nws_cos_poly(r);
  float v = (q & 1) ? c : s;
  return (q & 2) ? -v : v;
}

// ---------------------------------------------------------------------------------------------
// Fast sine for the 6.5 M oscillator evaluations per utterance: the hardware v_sin_f32 (argument in turns)
// behind an exact-product reduction  p = x*C_hi, e = fma(x, C_hi, -p) (the rounding error of p, exact),
// t = (p - rint(p)) + (e + x*C_lo).  Measured on MI355X against float64 (tools/measure_sin.py): max abs
// error 2.4e-7 for |x| <= 5e6 (rms 4.6e-8), 2.2x the throughput of nws_sinf.  The oscillator needs ~1e-6.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float nws_sin_turns(float x) {  // caller guarantees |x| <= 6e6
  const float c_hi = 0.15915493667125702f;     // fl32(1/(2 pi))
  const float c_lo = 6.4206382432985265e-09f;   // 1/(2 pi) - c_hi
  const float p = x * c_hi;
  const float e = fmaf(x, c_hi, -p);
  const float t = (p - rintf(p)) + fmaf(x, c_lo, e);
  return __builtin_amdgcn_sinf(t);
}

// Any-magnitude sine without a function call (the oscillator's wave-uniform wide-argument path: phases beyond ~6e6 rad,
// i.e. minutes of audio or very high F0): the same exact-product reduction to turns carried out in fp64 with a two-term
// 1/(2 pi) (106 bits), then v_sin_f32.  ~1e-7 absolute up to |x| ~ 1e25; inf/NaN give NaN like sinf.
__device__ __forceinline__ float nws_sin_wide(float x) {
  const double c1 = 0x1.45f306dc9c883p-3, c2 = -0x1.6b01ec5417056p-57;
  const double xd = (double)x;
  const double p = xd * c1;
  const double e = __builtin_fma(xd, c1, -p);
  const double t = __builtin_amdgcn_fract(p) + __builtin_fma(xd, c2, e);
  return __builtin_amdgcn_sinf((float)t);
}

// nws_sinf without a device function call on any path (a call inside a kernel that keeps dozens of accumulators live spills
// them around the call): the polynomial form up to 6e6, the inline fp64 reduction + v_sin_f32 beyond
__device__ __forceinline__ float nws_sinf_nocall(float x) {
  if (__builtin_expect(fabsf(x) > 6.0e6f, 0)) return nws_sin_wide(x);
  const float fq = rintf(x * 0.6366197723675814f);
  float r = fmaf(fq, -1.5707963705062866f, x);
  r = fmaf(fq, 4.371138828673793e-08f, r);
  int q = (int)fq;
  if (fabsf(x) > 32768.0f) {
    const float fq2 = rintf(r * 0.6366197723675814f);
    r = fmaf(fq2, -1.5707963705062866f, r);
    r = fmaf(fq2, 4.371138828673793e-08f, r);
    q += (int)fq2;
  }
  const float sv = nws_sin_poly(r);
  const float cv = nws_cos_poly(r);
  const float v = (q & 1) ? cv : sv;
  return (q & 2) ? -v : v;
}

__device__ __forceinline__ float nws_sin_turns_checked(float x) {
  return __builtin_expect(fabsf(x) > 6.0e6f, 0) ? nws_sin_wide(x) : nws_sin_turns(x);
}

// no device function call on any path: a call inside the fused kernel costs ~30 VGPRs of ABI clobbers and a scratch frame
__device__ __forceinline__ float nws_sinf_fast(float x) { return nws_sin_turns_checked(x); }

__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }

// Scalar add/sub the compiler cannot fold into a packed instruction.  A v_pk_{add,mul,fma}_f32 whose LOW lane takes the
// HIGH half of src1 (op_sel[1] = 1: "a.x + a.y" written on a register pair, a broadcast of the second element of a pair)
// returns wrong results on MI355X while ANOTHER kernel executes v_mfma_f32_32x32x16_f16 / 16x16x32_f16 on the same CU
// (LABBOOK.md "5.2"; tools/coexec_probe.py reproduces it).  The build scans every kernel for that form and fails on
// it; where the compiler derives it from ordinary source these helpers keep the arithmetic scalar.
__device__ __forceinline__ float nws_add_scalar(float a, float b) {
  float d;
  asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float nws_fma_scalar(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float nws_sub_scalar(float a, float b) {
  float d;
  asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// Phase-carry pass of one utterance by one workgroup of NWAVES waves (phase_carry_kernel, and the prologue of the GRU kernel
// when the two control-rate launches of a forward are fused): carry[c] = sum_{n < 32 c} f0_up[n] in float64.  The sums are
// exact (fp32 addends, < 2^53 of dynamic range), so the order of summation does not matter.
template <int NWAVES>
__device__ __forceinline__ void nws_phase_carry_block(const float* __restrict__ f0, const float* __restrict__ f0_up, int T,
                                                      double* __restrict__ carry, int b, int tid, double* wave_tot) {
  const int N = T * NWS_HOP;
  const int nchunks = N / 32;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  double running = 0.0;
  for (int base = 0; base < nchunks; base += 64 * NWAVES) {
    const int c = base + tid;
    double s = 0.0;
    if (c < nchunks) {
      const int n0 = c * 32;
      if (f0_up != nullptr) {
        const float4* p = reinterpret_cast<const float4*>(f0_up + (size_t)b * N + n0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 v = p[i];
          s += (double)v.x;
          s += (double)v.y;
          s += (double)v.z;
          s += (double)v.w;
        }
      } else {
        const float* x = f0 + (size_t)b * T;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
          const NwsLerp L = nws_lerp_coeff(n0 + i, T);
          s += (double)nws_lerp(x[L.i0], x[L.i1], L.w0, L.w1);
        }
      }
    }
    double v = s;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double t = __shfl_up(v, off, 64);
      if (lane >= off) v += t;
    }
    if (lane == 63) wave_tot[wave] = v;
    double excl = __shfl_up(v, 1, 64);
    if (lane == 0) excl = 0.0;
    __syncthreads();
    double wp = 0.0, total = 0.0;
#pragma unroll
    for (int i = 0; i < NWAVES; ++i) {
      const double t = wave_tot[i];
      if (i < wave) wp += t;
      total += t;
    }
    if (c < nchunks) carry[(size_t)b * nchunks + c] = running + wp + excl;
    running += total;
    __syncthreads();
  }
}

// 32-lane-half exchange (lane l <-> lane l^32)
__device__ __forceinline__ float nws_swap_halves(float v) { return __shfl_xor(v, 32, 64); }

__device__ __forceinline__ float nws_leaky_relu(float x) { return x > 0.0f ? x : 0.01f * x; }

// ---------------------------------------------------------------------------------------------
// Stateful streaming (stream.hip): the time-domain reverb of a hop as partial sums over 256 taps.  Shared with
// control_gru.hip: in a hop of <= 256 emitted samples every part but the first reads reverb input of EARLIER hops only, so
// those 124 partial sums per utterance run as extra workgroups of the hop's first launch (the recurrence) instead of as a
// launch of their own behind the oscillator and noise kernels.
// ---------------------------------------------------------------------------------------------
#define NWS_STREAM_RING 65536   // reverb-input ring per utterance (>= 31 999 samples of history + the longest chunk)

// pre-reverb value of emitted sample i of this step: oscillator branch [lo, hi) + noise branch (64 samples of residue first,
// then this window's hops)
struct NwsPreSrc {
  const float* newt_w;
  const float* noise_w;
  const float* residue;
  int Nw, lo, R0, noise_off;
};
__device__ __forceinline__ float nws_pre_value(const NwsPreSrc& P, int b, int i) {
  const float nz = i < P.R0 ? P.residue[(size_t)b * 64 + i] : P.noise_w[(size_t)b * P.Nw + P.noise_off + i - P.R0];
  return P.newt_w[(size_t)b * P.Nw + P.lo + i] + nz;
}

// sum_{i < 256} ir[256 p + i] * x[pos + j0 + tid - 1 - 256 p - i]  (tap m = 1 + 256 p + i, ir_[m] = ir[m-1]) for output
// j0 + tid of utterance b.  x: the ring for samples of earlier steps, nws_pre_value() for this step's own (HIST: the caller
// knows that the part reaches none of them and P is not read).  All 256 threads of the workgroup call it; xs: 512 floats,
// hs: 256 floats of LDS, both 16-byte aligned.
template <bool HIST>
__device__ __forceinline__ float nws_stream_reverb_partial(const float* __restrict__ ring, const NwsPreSrc& P,
                                                           const float* __restrict__ ir, int ir_len, int M, int b, int p, int j0,
                                                           long long pos, int tid, float* xs, float* hs) {
  const long long lo = pos + j0 - 256 - 256ll * p;         // xs[i] = x[lo + i], i < 512  (x before the stream's start is zero)
  const float* rb = ring + (size_t)b * NWS_STREAM_RING;
  for (int i = tid; i < 512; i += 256) {
    const long long a = lo + i;
    float v = 0.0f;
    if (!HIST && a >= pos) v = a - pos < M ? nws_pre_value(P, b, (int)(a - pos)) : 0.0f;
    else if (a >= 0 && a < pos) v = rb[a & (NWS_STREAM_RING - 1)];
    xs[i] = v;
  }
  hs[tid] = 256 * p + tid < ir_len ? ir[256 * p + tid] : 0.0f;
  __syncthreads();
  // output j0 + tid: x index pos + j0 + tid - 1 - 256 p - i = lo + (tid + 255 - i)
  float acc0 = 0.0f, acc1 = 0.0f;
#pragma unroll 8
  for (int i = 0; i < 256; i += 4) {
    const float4 h4 = *reinterpret_cast<const float4*>(&hs[i]);
    // (scalar by construction: control_gru.hip keeps the SLP vectoriser, which packs the two chains into the swizzled
    // v_pk_fma_f32 form the build refuses)
    acc0 = nws_fma_scalar(h4.x, xs[tid + 255 - i], acc0);
    acc1 = nws_fma_scalar(h4.y, xs[tid + 254 - i], acc1);
    acc0 = nws_fma_scalar(h4.z, xs[tid + 253 - i], acc0);
    acc1 = nws_fma_scalar(h4.w, xs[tid + 252 - i], acc1);
  }
  return nws_add_scalar(acc0, acc1);
}

// The shared part of a hop's head, one workgroup of NT threads (an extra workgroup of the frame-MLP launch in the fused hop,
// block B of stream_prep_kernel otherwise): applies the previous step's pending sample / frame counts - nobody else in the
// launch reads the counters - and moves the noise window on.
struct NwsStreamNoiseWin {
  float* nzwin;
  const float* noise_new;
  const float* noise_all;
  long long* counters;
  int nz_shift, nz_keep, n_new, noise_all_len, first, K;
};
template <int NT, bool ADVANCE = true>
__device__ __forceinline__ void nws_stream_noise_window_block(const NwsStreamNoiseWin& Z, int tid) {
  const int Tw = Z.first ? Z.K : Z.K + 1;
  const long long F = Z.counters[1] + Z.counters[3];
  __syncthreads();   // every thread has read the old values
  // (!ADVANCE - the four-launch hop: this workgroup shares its launch with readers of the pending form; the closing kernel's last
  // workgroup applies the counts)
  if (ADVANCE && tid == 0) {
    Z.counters[0] += Z.counters[2];
    Z.counters[1] = F;
    Z.counters[2] = 0;
    Z.counters[3] = 0;
  }
  if (Z.noise_all != nullptr) {
    // injected stream (parity runs): the window starts at absolute sample max(0, 128 (A0 - 1)), A0 = first frame of the window
    const long long A0 = Z.first ? 0 : F - 1;
    const long long start = A0 <= 0 ? 0 : 128 * (A0 - 1);
    const int want = 128 * (Tw + 1) + 1;
    for (int i = tid; i < want; i += NT) Z.nzwin[i] = start + i < Z.noise_all_len ? Z.noise_all[start + i] : 0.0f;
  } else {
    // drawn stream: keep the last nz_keep samples (shifted down by nz_shift), append the n_new fresh draws
    for (int i0 = 0; i0 < Z.nz_keep; i0 += NT) {
      const int i = i0 + tid;
      const float v = i < Z.nz_keep ? Z.nzwin[i + Z.nz_shift] : 0.0f;
      __syncthreads();
      if (i < Z.nz_keep) Z.nzwin[i] = v;
      __syncthreads();
    }
    for (int i = tid; i < Z.n_new; i += NT) Z.nzwin[Z.nz_keep + i] = Z.noise_new[i];
  }
}

// What the extra workgroups of the recurrence launch need (nws_control_gru_stream).  The launch sits IN FRONT of the
// workgroup that applies the previous step's pending sample count (the step's second launch): the position of this step's first
// sample is counters[0] + counters[2].
// * parts 1 .. parts - 1 of the reverb of every utterance (ring != NULL; M <= 256);
// * the per-utterance HEAD of the hop (f0_w != NULL): F0 window [previous frame | new frames], row 0 of the FiLM and FIR-tap windows
//   from the previous hop's last frame (rows 1 .. K are written by the frame-MLP kernel itself), phase carries spliced with the
//   carried phase sum.  Everything it reads is an input of the hop or state of the previous one.
struct NwsStreamSide {
  // reverb history
  const float* ring;
  const float* ir;
  float* partial;               // [part][b][M]
  const long long* counters;
  int ir_len, M, B, parts;
  // head
  const float* f0_new;          // (B, K)
  float* prev_f0;
  const float* prev_film;
  const float* prev_fir;
  double* S;
  float* f0_w;
  float* film_w;
  float* fir_w;
  double* carry;
  int K, first, final;
  // frame MLPs of the K <= 2 new frames inside this launch (gru_flag != NULL): two workgroups per utterance (mlp_few.h) fetch
  // their first weight fragments, then wait for the recurrence workgroup of their utterance, which leaves
  // gru_flag[b] = frames seen through this hop (counters[1] + counters[3] + K: unique per hop) behind its last gru_out row;
  // + one workgroup for the shared noise window (the pending counters stay: see the closing kernel of stream.hip)
  long long* gru_flag;          // (B)
  long long* counters_rw;       // the same counters, writable: [5] = 1 if a frame-MLP workgroup gave up waiting
  const float* gru_out;         // (B, K, 128)
  int out_T, out_off;           // rows per utterance of film_w / fir_w, row of the first new frame
  NwsStreamNoiseWin win;
};

// 256 threads; wave_tot: 4 doubles of LDS
__device__ __forceinline__ void nws_stream_head_block(const NwsStreamSide& H, int b, int tid, double* wave_tot) {
  const int K = H.K, Tw = H.first ? K : K + 1, off = H.first ? 0 : 1;
  for (int i = tid; i < Tw; i += 256) H.f0_w[(size_t)b * Tw + i] = (i < off) ? H.prev_f0[b] : H.f0_new[(size_t)b * K + i - off];
  if (off) {
    for (int c = tid; c < NWS_FILM_CH; c += 256) H.film_w[(size_t)b * Tw * NWS_FILM_CH + c] = H.prev_film[(size_t)b * NWS_FILM_CH + c];
    for (int c = tid; c < NWS_FIR_HALF; c += 256) H.fir_w[(size_t)b * Tw * NWS_FIR_HALF + c] = H.prev_fir[(size_t)b * NWS_FIR_HALF + c];
  }
  __syncthreads();   // the F0 window is complete (global writes of this block are visible to it) and prev_f0 has been consumed
  if (tid == 0) H.prev_f0[b] = H.f0_new[(size_t)b * K + K - 1];
  // exclusive float64 prefix sums of the window's upsampled F0 at 32-sample granularity (the lerp clamps at the window edges;
  // those 64 + 64 samples are only emitted at the true ends of the stream)
  nws_phase_carry_block<4>(H.f0_w, nullptr, Tw, H.carry, b, tid, wave_tot);
  __syncthreads();
  const int nch = 4 * Tw;
  double* cb = H.carry + (size_t)b * nch;
  if (!H.first) {
    // sample 64 of the window is the first one not yet emitted: its running sum must continue the carried one.  Sums of
    // fp32 values in fp64 are exact, so the spliced carries are bit-identical to the one-shot forward's
    const double base = cb[2], s0 = H.S[b];
    __syncthreads();
    for (int c = tid; c < nch; c += 256) cb[c] = s0 + (cb[c] - base);
    __syncthreads();
  }
  if (!H.final && tid == 0) H.S[b] = cb[nch - 2];    // through the last emitted sample (128 Tw - 64)
}

// control_gru.hip: nws_control_gru_state with the extra workgroups `side` asks for (NULL: none).  Internal to the library.
extern "C" __attribute__((visibility("hidden"))) int nws_control_gru_stream(const NwsWeights* w, const float* control, int B, int C,
                                                                            int T, const float* h0, float* gru_out, float* hT,
                                                                            const NwsStreamSide* side, void* stream);
// frame_mlps.hip: the 32-frame tile kernel on T <= 32 new frames per utterance, FiLM / FIR-tap rows written as rows out_off .. of
// windows of out_T rows per utterance, + one extra workgroup running nws_stream_noise_window_block.  NWS_ERR_UNSUPPORTED if the
// fragment tables are missing or T > 32.  Internal to the library.
extern "C" __attribute__((visibility("hidden"))) int nws_frame_mlps_stream(const NwsWeights* w, const float* gru_out, int B, int T,
                                                                           float* film_w, float* fir_w, int out_T, int out_off,
                                                                           const NwsStreamNoiseWin* win, void* stream);
