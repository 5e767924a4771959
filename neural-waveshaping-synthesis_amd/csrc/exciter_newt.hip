// Fused harmonic exciter + waveshaper bank for gfx950 (wave64, fp32 MFMA).
//
// Replaces, per utterance b and 128-sample hop j (one workgroup = 4 waves x 32 samples):
//   F.upsample(f0)                      models/neural_waveshaping.py:75
//   HarmonicOscillator.forward          models/modules/generators.py:58-66
//   harmonic_mixer  Conv1d(101->64,1)   models/neural_waveshaping.py:66
//   NEWT.forward / FastNEWT.shaping_fn  models/modules/shaping.py:67-79, :136-151
//
// Design: the (B,101,N) oscillator bank and the (B,64,N) exciter of the
// reference never exist in HBM.  Each lane evaluates 8 sines per MFMA K-step directly in the
// B-fragment layout of v_mfma_f32_32x32x16_f16 (lane l -> harmonics 16ks + 8(l>>5) + 1..8 of sample
// l&31).  The 101->64 mixer runs on the fp16 matrix pipe with BOTH operands split into two fp16
// terms (W = W_hi + W_lo staged once per workgroup in LDS in A-fragment order, v = v_hi + v_lo):
// W_hi v_hi + W_hi v_lo + W_lo v_hi keeps 22 bits per product with fp32 accumulation -- measured
// indistinguishable from the exact-fp32 MFMA it replaced, at 1/5 of its matrix-pipe time, and unlike
// the fp32 MFMA it overlaps with the VALU work.  The 64x32 accumulator tile then goes straight through
// FiLM -> LUT (or sin-MLP) -> FiLM -> 64->1 mix in registers; one coalesced 128 B store per wave.
#include <cstdlib>
#include <type_traits>

#include "nws_common.h"

namespace {

constexpr int kK = NWS_N_HARMONICS;         // 101
constexpr int kKSteps = 7;                  // MFMA K-steps of 16 harmonics
constexpr int kKPad = 16 * kKSteps;         // 112
constexpr int kS = NWS_N_SHAPERS;           // 64
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifndef NWS_LUT_GROUP
#define NWS_LUT_GROUP 4
#endif
constexpr int kLutGroup = NWS_LUT_GROUP;   // table gathers in flight together in the fused tail (one memory round trip per group)
constexpr int kTile = 128;                  // samples per workgroup (= control hop)
constexpr float kTau = 6.283185307179586f;  // fl32(math.tau)
constexpr float kPi = 3.141592653589793f;   // fl32(math.pi)

// OPT bits of exciter_newt_kernel (compile-time variants of the FastNEWT hot path)
enum Opt {
  // (bit 1 was kOptScalarSines, the main loop in scalar fp32: measured slower in round 2, selected by no path since; retired in round 5)
  kOptFilmMfma = 2,     // FiLM interpolation (3 parameter types x 64 shapers x 32 samples per wave) as six bf16 MFMAs
  kOptOneTerm = 4,      // sines as ONE fp16 term in EVERY K-step (drops W_hi * v_lo: 2 MFMAs per product; 11-bit activations)
  kOptHybrid = 8,       // two-term sines in K-step 0 (mixer bias + harmonics 1..15), one term in K-steps 1..6
  kOptHybridW = 16,     // with kOptHybrid: the mixer WEIGHTS of harmonics 16..101 as one fp16 term too (1 MFMA per product)
  kOptLowReg = 32,      // with kOptFilmMfma: the tail keeps at most two FiLM tiles live (80 VGPRs: three 8-wave workgroups per CU)
  // with kOptFilmMfma | kOptLowReg, two hops per workgroup: the FiLM rows arrive as per-FRAME bf16x3 fragment records
  // (NWS_FILM_REC_BYTES per frame, film_frag_record below) and go to LDS by the LDS-DMA path - no staging arithmetic in this
  // kernel.  Round 6, VERDICT r5 #3: built, 3.6e-9 RMS from the product kernel, and MEASURED AS NOTHING - prologue-only launch
  // 36.8 -> 34.7 us, whole kernel 266.8 -> 263.6 us (profiles/r06/film_dma_ab.txt; another box: 35.8 -> 34.4, 257.4 -> 255.0): per workgroup the prologue-only launch is one
  // memory round trip and one barrier (16 000 workgroups / 768 slots = 21 rounds x 1.6 us), not the staging arithmetic (~100 of a
  // wave's ~800 vector instructions on three of eight waves), and in the real kernel the CU's other two workgroups work meanwhile.
  // Not on the product path (the records would cost the frame-MLP kernel 17 MB more stores per step); kept behind
  // nws_debug_exciter_newt variants 6 / 108.
  kOptFilmDma = 64
};
static_assert((kOptFilmMfma ^ kOptOneTerm ^ kOptHybrid ^ kOptHybridW ^ kOptLowReg ^ kOptFilmDma) == 126, "Opt bits must be distinct");
enum Mode { kModeLut = 0, kModeExact = 1, kModeExciterOnly = 2, kModeLutPairs = 3, kModeLutPairsDiv6 = 4, kModeExactBank = 5,
            kModeExactBankNF = 6 };   // NF: no v_fract in front of the sines of the hidden and output layers (NWS_EXCITER_BANK_NOFRACT)
__host__ __device__ constexpr bool is_bank(int mode) { return mode == kModeExactBank || mode == kModeExactBankNF; }
__host__ __device__ constexpr bool is_lut(int mode) {
  return mode == kModeLut || mode == kModeLutPairs || mode == kModeLutPairsDiv6;
}

// ---------------------------------------------------------------------------------------------
// carry[b][c] = sum_{n<32c} f0_up[b][n] in float64 (torch's CPU cumsum accumulates in double and
// rounds each element to fp32).  One workgroup per utterance.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void phase_carry_kernel(const float* __restrict__ f0,
                                                           const float* __restrict__ f0_up, int T,
                                                           double* __restrict__ carry) {
  __builtin_amdgcn_s_setprio(3);  // one workgroup per utterance on the control stream: latency matters, throughput does not
  __shared__ double wave_tot[16];
  nws_phase_carry_block<16>(f0, f0_up, T, carry, blockIdx.x, threadIdx.x, wave_tot);
}

// ---------------------------------------------------------------------------------------------
// exact shaper s at scalar argument x (TrainableNonlinearity, models/modules/shaping.py:36-37;
// grouped 1x1 convs == 64 independent 1->8->8->8->1 sin-MLPs).  Weights in LDS.
// ---------------------------------------------------------------------------------------------
struct ShaperLds {
  float in_scale[64];
  float w0[512], b0[512];
  float w2t[4096], b2[512];  // w2t[s][i][o] = net.2.weight[s*8+o][i]: the 8 outputs of one input are contiguous
  float w4t[4096], b4[512];  //   -> two packed-fp32 operands per ds_read_b128
  float w6[512], b6[64];
};

// In-kernel exact shaper, fast form.  The weights of all four layers are staged in LDS pre-multiplied by 1/(2 pi)
// (load_shaper_lds<true>), so that every pre-activation is already in TURNS and a sine is v_fract_f32 + v_sin_f32: no
// range-reduction arithmetic, no magnitude test (the fract handles any magnitude an fp32 argument can resolve).  The scaled
// weights carry one extra rounding (2^-24 relative), i.e. an argument error of the size of the reference's own rounding of
// the same pre-activation; held to the same parity bar as every other stage (tests/test_gpu_parity.py).
__device__ __forceinline__ float sin_of_turns(float t) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t)); }

// one dense 8 -> 8 sine layer in packed fp32: out[o] = sin(2 pi (b[o] + sum_i w[o][i] h[i])), weights transposed in LDS
__device__ __forceinline__ void sine_layer8(const float* __restrict__ wt, const float* __restrict__ bias,
                                            const f32x2 (&h)[4], f32x2 (&out)[4]) {
  const float4 b0 = *reinterpret_cast<const float4*>(bias), b1 = *reinterpret_cast<const float4*>(bias + 4);
  f32x2 acc[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float4 wa = *reinterpret_cast<const float4*>(wt + 8 * i), wb = *reinterpret_cast<const float4*>(wt + 8 * i + 4);
    const f32x2 hi = splat2((i & 1) ? h[i >> 1].y : h[i >> 1].x);
    acc[0] = fma2(f32x2{wa.x, wa.y}, hi, acc[0]);
    acc[1] = fma2(f32x2{wa.z, wa.w}, hi, acc[1]);
    acc[2] = fma2(f32x2{wb.x, wb.y}, hi, acc[2]);
    acc[3] = fma2(f32x2{wb.z, wb.w}, hi, acc[3]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) out[q] = f32x2{sin_of_turns(acc[q].x), sin_of_turns(acc[q].y)};
}

// (TrainableNonlinearity, models/modules/shaping.py:36-37) on the packed-fp32 / v_sin_f32 path; W holds the SCALED weights.
// Not inlined: 32 inlined copies in the unrolled tail push the kernel past 256 VGPRs into scratch.
__device__ __noinline__ float exact_shaper(const ShaperLds& W, int s, float x) {
  const float a = W.in_scale[s] * x;
  const float4 wa = *reinterpret_cast<const float4*>(&W.w0[s * 8]), wb = *reinterpret_cast<const float4*>(&W.w0[s * 8 + 4]);
  const float4 ba = *reinterpret_cast<const float4*>(&W.b0[s * 8]), bb = *reinterpret_cast<const float4*>(&W.b0[s * 8 + 4]);
  f32x2 h1[4] = {{sin_of_turns(fmaf(wa.x, a, ba.x)), sin_of_turns(fmaf(wa.y, a, ba.y))},
                 {sin_of_turns(fmaf(wa.z, a, ba.z)), sin_of_turns(fmaf(wa.w, a, ba.w))},
                 {sin_of_turns(fmaf(wb.x, a, bb.x)), sin_of_turns(fmaf(wb.y, a, bb.y))},
                 {sin_of_turns(fmaf(wb.z, a, bb.z)), sin_of_turns(fmaf(wb.w, a, bb.w))}};
  f32x2 h2[4];
  sine_layer8(&W.w2t[s * 64], &W.b2[s * 8], h1, h2);
  sine_layer8(&W.w4t[s * 64], &W.b4[s * 8], h2, h1);
  const float4 va = *reinterpret_cast<const float4*>(&W.w6[s * 8]), vb = *reinterpret_cast<const float4*>(&W.w6[s * 8 + 4]);
  f32x2 acc = f32x2{va.x, va.y} * h1[0];
  acc = fma2(f32x2{va.z, va.w}, h1[1], acc);
  acc = fma2(f32x2{vb.x, vb.y}, h1[2], acc);
  acc = fma2(f32x2{vb.z, vb.w}, h1[3], acc);
  return sin_of_turns(W.b6[s] + nws_add_scalar(acc.x, acc.y));
}

// reference-grade variant (Cody-Waite polynomial sine, scalar): used to BUILD the FastNEWT table and by the stand-alone
// shaper entry point, where accuracy matters more than speed
__device__ __forceinline__ float exact_shaper_precise(const ShaperLds& W, int s, float x) {
  const float a = W.in_scale[s] * x;
  float h1[8], h2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h1[i] = nws_sinf(fmaf(W.w0[s * 8 + i], a, W.b0[s * 8 + i]));
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float acc = W.b2[s * 8 + o];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(W.w2t[s * 64 + i * 8 + o], h1[i], acc);
    h2[o] = nws_sinf(acc);
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float acc = W.b4[s * 8 + o];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(W.w4t[s * 64 + i * 8 + o], h2[i], acc);
    h1[o] = nws_sinf(acc);
  }
  float acc = W.b6[s];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = fmaf(W.w6[s * 8 + i], h1[i], acc);
  return nws_sinf(acc);
}

// TURNS: weights and biases of every layer times fl32(1 / (2 pi)) (exact_shaper above); in_scale stays as it is
template <bool TURNS>
__device__ __forceinline__ void load_shaper_lds(ShaperLds& L, const NwsWeights& w, int tid, int nthreads) {
  const float c = TURNS ? 0.15915493667125702f : 1.0f;
  for (int i = tid; i < 64; i += nthreads) {
    L.in_scale[i] = w.shaper_in_scale[i];
    L.b6[i] = w.shaper_b6[i] * c;
  }
  for (int i = tid; i < 512; i += nthreads) {
    L.w0[i] = w.shaper_w0[i] * c;
    L.b0[i] = w.shaper_b0[i] * c;
    L.b2[i] = w.shaper_b2[i] * c;
    L.b4[i] = w.shaper_b4[i] * c;
    L.w6[i] = w.shaper_w6[i] * c;
  }
  for (int i = tid; i < 4096; i += nthreads) {  // i = (s*8 + o)*8 + in  ->  [s][in][o]
    const int sidx = i >> 6, o = (i >> 3) & 7, in = i & 7;
    L.w2t[sidx * 64 + in * 8 + o] = w.shaper_w2[i] * c;
    L.w4t[sidx * 64 + in * 8 + o] = w.shaper_w4[i] * c;
  }
}

// ---------------------------------------------------------------------------------------------
// Exact shapers, bank form (kModeExactBank).  In the fused tail a lane holds 32 DIFFERENT shapers of one sample, so every
// (sample, shaper) evaluation fetched its own 170 weights from LDS: 42 ds_read_b128 per evaluation, and the LDS return
// path (not the sines) bounded the kernel.  Here the FiLM'ed shaper inputs of the workgroup's 128 samples go through LDS
// once (xi[shaper][sample]); then every wave evaluates its shapers with lane = sample, two samples per lane:
// the shaper index is wave-uniform, its weights come from the nws_shaper_turns() table by SCALAR loads and feed the
// VALU as SGPR operands - no LDS traffic for weights, no VGPRs for them either.  Per-wave partial sums of the 64 -> 1
// mix meet in LDS.  Table row of shaper s (176 floats, every layer already times 1 / (2 pi)):
//   [0,8) w0  [8,16) b0  [16,80) w2t[in][out]  [80,88) b2  [88,152) w4t[in][out]  [152,160) b4  [160,168) w6  168 b6  169 in_scale
// LDS: the planes go where the mixer's fragment tables were (dead once the K loop is over) and hold ONE M-tile of the
// accumulators at a time (32 shapers: 16.5 KB + the partial sums < the 28 KB of whi / wlo), two passes: the workgroup needs
// the 38.8 KB of ExcLds and nothing else - 4 workgroups = 4 waves per SIMD on a CU.  (Round 4, up to here: all 64 shapers at
// once in 35 KB BEHIND ExcLds = 74.6 KB per workgroup, i.e. two workgroups = 2 waves per SIMD for a kernel that is a chain
// of quarter-rate sines and dependent packed FMAs.)
// ---------------------------------------------------------------------------------------------
constexpr int kBankRow = NWS_SHAPER_TURNS_ROW;
struct BankLds {
  float xi[32][kTile + 4];   // +4: the two M-tile halves of a store instruction land in different banks
  float red[4][kTile];
};

// FRACT = false: the pre-activation goes to v_sin_f32 as it is (in turns).  The instruction reduces arguments inside +-256 turns
// by itself; hidden and output layers of a sin-MLP see |pre| <= sum |W| + |b| (their inputs are sines), which the host checks
// against that domain once per weights version (engine.py: bank_nofract_safe; 0.6 turns for the shipped checkpoints).  One
// quarter-rate instruction per sine instead of two: 17 of the 25 sines of an evaluation (8 + 8 + 1; the 8 first-layer sines keep
// their v_fract, their argument is unbounded).
template <bool FRACT>
__device__ __forceinline__ float bank_sin(float t) { return FRACT ? sin_of_turns(t) : __builtin_amdgcn_sinf(t); }

template <bool FRACT>
__device__ __forceinline__ void bank_layer8(const float* __restrict__ wt, const float* __restrict__ bias,
                                            const float (&h)[8], float (&out)[8]) {
  f32x2 acc[4] = {{bias[0], bias[1]}, {bias[2], bias[3]}, {bias[4], bias[5]}, {bias[6], bias[7]}};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const f32x2 hi = splat2(h[i]);
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = fma2(f32x2{wt[8 * i + 2 * q], wt[8 * i + 2 * q + 1]}, hi, acc[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    out[2 * q] = bank_sin<FRACT>(acc[q].x);
    out[2 * q + 1] = bank_sin<FRACT>(acc[q].y);
  }
}

// W: wave-uniform pointer to the shaper's table row (scalar loads).  Every sine keeps its v_fract: a variant without it for
// shapers whose hidden pre-activations provably stay inside v_sin_f32's +-256-turn domain was measured SLOWER (1.69 vs
// 1.56 ms at B=64: two copies of the loop body, one uniform branch per shaper).
// (Round 4, measured and dropped: the two 8 x 8 hidden layers as 2 x 16 v_mfma_f32_4x4x1_16B_f32 per evaluation - lane 4 blk + j
// holds sample j's activation as B and its four output rows as the accumulator, the wave-uniform weight column comes from lane
// 4 blk + i; layout probed in tools/ubench/mfma4x4x1_layout.hip.  Same results, 1.71 against 1.49 ms per B = 64 launch: eight
// dependent two-pass MFMAs per accumulator with only four chains in flight, plus 32 per-lane weight loads per shaper.)
template <bool FRACT>
__device__ __forceinline__ float bank_shaper(const float* __restrict__ W, float x) {
  const float a = W[169] * x;
  float h1[8], h2[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) h1[o] = sin_of_turns(fmaf(W[o], a, W[8 + o]));    // the first layer's argument is unbounded: always reduced
  bank_layer8<FRACT>(W + 16, W + 80, h1, h2);
  bank_layer8<FRACT>(W + 88, W + 152, h2, h1);
  float acc = W[168];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = fmaf(W[160 + i], h1[i], acc);
  return bank_sin<FRACT>(acc);
}

__global__ void shaper_turns_kernel(NwsWeights w, float* __restrict__ out) {
  const int s = blockIdx.x, k = threadIdx.x;
  if (k >= kBankRow) return;
  const float c = 0.15915493667125702f;  // fl32(1 / (2 pi))
  float v = 0.0f;
  if (k < 8) v = w.shaper_w0[s * 8 + k] * c;
  else if (k < 16) v = w.shaper_b0[s * 8 + k - 8] * c;
  else if (k < 80) v = w.shaper_w2[(s * 8 + ((k - 16) & 7)) * 8 + ((k - 16) >> 3)] * c;   // [in][out] <- weight[s*8+out][in]
  else if (k < 88) v = w.shaper_b2[s * 8 + k - 80] * c;
  else if (k < 152) v = w.shaper_w4[(s * 8 + ((k - 88) & 7)) * 8 + ((k - 88) >> 3)] * c;
  else if (k < 160) v = w.shaper_b4[s * 8 + k - 152] * c;
  else if (k < 168) v = w.shaper_w6[s * 8 + k - 160] * c;
  else if (k == 168) v = w.shaper_b6[s] * c;
  else if (k == 169) v = w.shaper_in_scale[s];
  out[s * kBankRow + k] = v;
}

// FastNEWT.shaping_fn (models/modules/shaping.py:136-151), quirks kept: index scale size/(max-min)
// against a linspace grid of step (max-min)/(size-1); fract taken against the CLAMPED lower index.
struct LutParams {
  const float* table;
  const float2* pairs;
  int size;
  float tmin, trange, rcp_range;
};

// DIV6: trange == 6 (the reference's default table range): t/6 == fma(fma(-q,6,t), 1/6, q) with q = t*(1/6) for
// EVERY fp32 t (checked exhaustively over all mantissas), 3 instructions instead of the ~10 of an IEEE division.
// Compile-time on purpose: a runtime-uniform test makes hipcc unswitch the whole unrolled tail and spill it.
template <bool DIV6>
__device__ __forceinline__ float lut_index(const LutParams& P, float x) {
  const float t = (float)P.size * (x - P.tmin);
  if (DIV6) {
    const float q = t * P.rcp_range;
    const float r = fmaf(-q, P.trange, t);
    return fmaf(r, P.rcp_range, q);
  }
  return __fdiv_rn(t, P.trange);
}

// row_off = shaper * size (elements).  With the derived pair table {T[i], fl(T[min(i+1,size-1)] - T[i])}
// (nws_lut_pairs) one ALIGNED 8-byte gather serves the lookup; fl(upper - lower) is the reference's own
// intermediate, so the result stays bit-identical.  Without it: two 4-byte gathers.
template <bool PAIRS, bool DIV6>
__device__ __forceinline__ float lut_shaper(const LutParams& P, int row_off, float x) {
  const float idx = lut_index<DIV6>(P, x);
  float fl = floorf(idx);
  fl = fmaxf(fl, 0.0f);
  fl = fminf(fl, (float)(P.size - 1));
  const int lo = (int)fl;
  const float fract = idx - fl;
  if (PAIRS) {  // compile-time: a runtime test here makes hipcc unswitch the whole unrolled tail and spill it
    const float2 v = P.pairs[(unsigned)(row_off + lo)];
    return v.y * fract + v.x;
  }
  const int up = lo + 1 < P.size ? lo + 1 : P.size - 1;
  const float lv = P.table[(unsigned)(row_off + lo)];
  const float uv = P.table[(unsigned)(row_off + up)];
  return (uv - lv) * fract + lv;
}

// two adjacent shapers (row_off, row_off + size) at once; index chain in packed fp32, rounding for rounding as above
template <bool PAIRS, bool DIV6>
__device__ __forceinline__ f32x2 lut_shaper2(const LutParams& P, int row_off, f32x2 x) {
  // scalar on purpose: "x - splat(tmin)" with tmin in the second register of a pair is the swizzled-src1 packed form
  const f32x2 t = splat2((float)P.size) * f32x2{nws_sub_scalar(x.x, P.tmin), nws_sub_scalar(x.y, P.tmin)};
  f32x2 idx;
  if (DIV6) {
    const f32x2 q = t * splat2(P.rcp_range);
    const f32x2 r = fma2(-q, splat2(P.trange), t);
    idx = fma2(r, splat2(P.rcp_range), q);
  } else {
    idx = f32x2{__fdiv_rn(t.x, P.trange), __fdiv_rn(t.y, P.trange)};
  }
  const float top = (float)(P.size - 1);
  const f32x2 fl = {__builtin_amdgcn_fmed3f(floorf(idx.x), 0.0f, top), __builtin_amdgcn_fmed3f(floorf(idx.y), 0.0f, top)};
  const f32x2 fract = idx - fl;
  const int lo0 = (int)fl.x, lo1 = (int)fl.y;
  if (PAIRS) {
    const float2 a = P.pairs[(unsigned)(row_off + lo0)];
    const float2 b = P.pairs[(unsigned)(row_off + P.size + lo1)];
    return f32x2{a.y * fract.x + a.x, b.y * fract.y + b.x};
  }
  const int up0 = lo0 + 1 < P.size ? lo0 + 1 : P.size - 1, up1 = lo1 + 1 < P.size ? lo1 + 1 : P.size - 1;
  const float l0 = P.table[(unsigned)(row_off + lo0)], u0 = P.table[(unsigned)(row_off + up0)];
  const float l1 = P.table[(unsigned)(row_off + P.size + lo1)], u1 = P.table[(unsigned)(row_off + P.size + up1)];
  return f32x2{(u0 - l0) * fract.x + l0, (u1 - l1) * fract.y + l1};
}

// Hot-path form of lut_shaper2<true, true> (range 6, power-of-two size; chosen by the launcher):
//  * x arrives as x - min (the FiLM bias carries -min: one rounding where the reference has two, <= 1/2 ulp of x);
//  * size * (x - min) / 6 with the power of two folded into the division constants: q = y*(size/6), r = fma(-q, 6/size, y),
//    idx = fma(r, size/6, q) is the same correctly rounded quotient (every step is the DIV6 step scaled exactly by `size`);
//  * the gather address is  SGPR row base + 32-bit VGPR byte offset  (one v_add_lshl_u32 instead of 64-bit address math);
//  * the table lerp T[i] + d[i]*fract is ONE fma per shaper (a single rounding where the reference has two: <= 1/2 ulp of
//    the output, far below the sine error upstream); scalar on purpose, packing it would cost three v_mov per pair.
struct LutFast {
  const char* pairs;
  float c_r, c_d, top;
  unsigned row_bytes;
};

__device__ __forceinline__ LutFast make_lut_fast(const NwsWeights& w) {
  LutFast P;
  P.pairs = reinterpret_cast<const char*>(w.lut_pairs);
  P.c_r = (float)w.lut_size * (1.0f / 6.0f);  // fl(1/6) scaled exactly
  P.c_d = 6.0f / (float)w.lut_size;           // exact
  P.top = (float)(w.lut_size - 1);
  P.row_bytes = (unsigned)w.lut_size * 8u;
  return P;
}

__device__ __forceinline__ f32x2 lut_shaper2_fast(const LutFast& P, const char* __restrict__ row0, unsigned lane_off,
                                                  f32x2 x) {
  const f32x2 y = x;  // already x - lut_min: folded into the FiLM bias when it was staged
  const f32x2 q = y * splat2(P.c_r);
  const f32x2 r = fma2(-q, splat2(P.c_d), y);
  const f32x2 idx = fma2(r, splat2(P.c_r), q);
  const f32x2 fl = {__builtin_amdgcn_fmed3f(floorf(idx.x), 0.0f, P.top), __builtin_amdgcn_fmed3f(floorf(idx.y), 0.0f, P.top)};
  const f32x2 fract = idx - fl;
  const unsigned o0 = lane_off + ((unsigned)(int)fl.x << 3), o1 = lane_off + ((unsigned)(int)fl.y << 3);
  const float2 a = *reinterpret_cast<const float2*>(row0 + o0);
  const float2 b = *reinterpret_cast<const float2*>(row0 + P.row_bytes + o1);
  // asm: left to itself hipcc re-packs the two FMAs into one v_pk_fma_f32 behind three v_mov (4 instructions instead of 2)
  f32x2 out;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(out.x) : "v"(a.y), "v"(fract.x), "v"(a.x));
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(out.y) : "v"(b.y), "v"(fract.y), "v"(b.x));
  return out;
}

__device__ __forceinline__ LutParams make_lut_params(const NwsWeights& w) {
  LutParams P;
  P.table = w.lut;
  P.pairs = reinterpret_cast<const float2*>(w.lut_pairs);
  P.size = w.lut_size;
  P.tmin = w.lut_min;
  P.trange = w.lut_max - w.lut_min;
  P.rcp_range = 1.0f / P.trange;
  return P;
}

// Per-frame FiLM fragment record (kOptFilmDma): for type ty in {0 index gain, 1 index bias, 2 output gain} and shaper s, 8 bytes
// at (ty * 64 + s) * 8 = the value as THREE bf16 terms by truncation {t0, t1, t2, 0} (8 + 8 + 8 bits: exact for any fp32),
//   ty 0: (size / 6) g_idx        ty 1: (size / 6) (b_idx - lut_min)        ty 2: newt.mixer.weight[s] g_norm
// (the table-unit scaling and the folded 64 -> 1 mixer weight of the staging code above).  As the A operand of
// v_mfma_f32_32x32x16_bf16, lane (shaper i, half h) holds the record of frame f + h in K slots 8 h + 0..3 (slots 8 h + 4..7 zero);
// against B = {1 - w, 1 - w, 1 - w, 0, ...} in half 0 and {w, w, w, 0, ...} in half 1 (w = interpolation weight, a multiple of
// 1 / 256 below 1: w and 1 - w are exact in bf16) the instruction returns (1 - w) p[f] + w p[f + 1]: F.upsample's own formula
// (neural_waveshaping.py:75 semantics, shaping.py:69) for 32 shapers x 32 samples, from records that know nothing of their
// neighbours - which is what would let the frame-MLP kernel write them.  Beside the records, 16 bytes per frame (film_frag_aux):
// { sum_s newt.mixer.weight[s] b_norm[s] (float), 0, range-proof mask (64 bits: shaper s cannot leave the table at this frame) }.
struct FilmAux {
  float bsum, pad;
  unsigned long long mask;
};
static_assert(sizeof(FilmAux) == 16, "FilmAux");

// Mixer weights as two fp16 terms, W = W_hi + W_lo (22+ significant bits), in MFMA A-fragment order:
// fragment (ks, m, h, i) = 8 halfs = W[32m + i][16ks + 8h .. +7]; lane (i, h) reads it with one ds_read_b128.
struct ExcLds {
  f16x8 whi[kKSteps * 2 * 2 * 32];
  f16x8 wlo[kKSteps * 2 * 2 * 32];
  // FiLM rows of frames j-1, j, j+1 (clamped) as value + difference to the next frame, so that the reference's linear
  // upsampling (shaping.py:69) is ONE packed FMA per parameter: p(n) = fa + w1(n) * fd.  Types: 0 g_idx, 1 b_idx,
  // 2 out_w * g_norm (the 64->1 mixer weight folded in; its bias part  sum_s out_w[s] b_norm[s]  is the per-frame scalar bsum).
  // SoA: 4 consecutive shapers = one ds_read_b128 = two packed-fp32 operands.
  union {
    struct {
      float fa[3][3][kS];   // slots: frame pairs (jb-1, jb), (jb, jb+1) [, (jb+1, jb+2) when the workgroup covers two hops]
      float fd[3][3][kS];
    };
    // kOptFilmMfma: the same three parameter types as MFMA A fragments.  Row (slot, type, M-tile, shaper i) = 8 bf16 =
    // {a0, a1, a2, d0, d1, d2, 0, 0}: value a and frame difference d each as THREE bf16 terms (8 + 8 + 8 bits: exact for any
    // fp32, fp32 exponent range), against the B operand {1, 1, 1, w, w, w, 0, 0} (w = interpolation weight, a multiple of
    // 1/256: exact in bf16) one v_mfma_f32_32x32x16_bf16 returns a + w d for 32 shapers x 32 samples.
    uint4 ffrag[3][3][2][32];
    // kOptFilmDma: the fragment records of the workgroup's four frames (jb-1 .. jb+2, clamped), as they lie in memory
    unsigned char frec[4][NWS_FILM_REC_BYTES];
  };
  float bsum[4];
  unsigned long long fmask[4];   // kOptFilmDma: per-FRAME range-proof masks (a frame pair is proven where both frames are)
  // bit s of okmask[slot]: the table index of shaper s provably stays inside the table for every sample that interpolates
  // between the slot's two frames (staging, from NwsWeights.exciter_bound); 0 = unknown -> the clamped lookup
  unsigned long long okmask[3];
  // K slot c = 16ks + 8half + e: slot 0 is the mixer BIAS (its "sine" is the constant 1), slot c >= 1 is harmonic c
  // (16-byte aligned: the K-steps read shift and kf as ds_read_b128 with immediate offsets; behind okmask[3] the pair sat at 8 mod 16 and every
  // K-step spent four v_add_u32 on ds_read2_b64 addresses - no time in it, see LABBOOK 'Round 6, second half')
  alignas(16) float shift[kKPad];   // phase shift of slot c (0 for slot 0 and the padding slots 102..111)
  float kf[kKPad];              // (float)c: harmonic numbers as packed-FMA operands, read instead of computed
};

// LDS budget behind the occupancy figures: ExcLds is ALL the dynamic LDS of the hot kernel (three 8-wave workgroups
// per CU = 116 KB) and of the exact-shaper bank kernel (four 4-wave workgroups = 155 KB of the 160 KB)
static_assert(sizeof(ExcLds) <= 40960, "ExcLds above 40 KB: the bank kernel drops from four to three workgroups per CU");

// Where K slot kk of shaper s lives in the fragment tables (units: halfs).  K-steps 0..5 are v_mfma_f32_32x32x16_f16
// fragments (lane (i, h) holds slots 16 ks + 8 h + 0..7); the remainder - slots 96..103: harmonics 96..101 and two zero
// slots - is ONE v_mfma_f32_32x32x8_f16 step (lane (i, h) holds slots 96 + 4 h + 0..3, the first 8 bytes of a 16-byte
// row); slots 104..111 do not exist, their table positions (the unused upper halves of those rows) are written as zeros.
__host__ __device__ inline int mixer_frag_index(int s, int kk) {
  const int m = s >> 5, i = s & 31;
  if (kk < 96) return ((((kk >> 4) * 2 + m) * 2 + ((kk >> 3) & 1)) * 32 + i) * 8 + (kk & 7);
  const int r = kk - 96;   // 0..15
  return (((6 * 2 + m) * 2 + ((r >> 2) & 1)) * 32 + i) * 8 + (r & 3) + 4 * (r >> 3);
}

// harmonic_mixer as K-slot weights: slot 0 = bias, slots 1..101 = weight[:, c-1], padding slots = 0
__device__ __forceinline__ float mixer_slot_weight(const float* __restrict__ mixer_w, const float* __restrict__ mixer_b, int s,
                                                   int c) {
  return c == 0 ? mixer_b[s] : (c <= kK ? mixer_w[s * kK + c - 1] : 0.0f);
}

// ---- cross-lane helpers on the DPP path (no LDS round trips, no index arithmetic) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  // all rows enabled: lanes without a source read zero by bound_ctrl, no zero-initialised destination needed
  constexpr bool kBound = ROW_MASK == 0xf;
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, kBound);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, kBound);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// inclusive prefix sum inside each 32-lane half: row_shr 1/2/4/8 (zeros shifted in) scan the 16-lane rows, row_bcast15
// carries the lower row's total into the upper row
__device__ __forceinline__ double scan32_f64(double v) {
  v += dpp_f64<0x111, 0xf>(v);
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);
  return v;
}
// sum over the wave, valid in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_f32<0xb1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_f32<0x4e, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_f32<0x140, 0xf>(v);  // row_mirror: every lane holds its row's sum
  v += dpp_f32<0x142, 0xa>(v);  // row_bcast15 -> rows 1 and 3
  v += dpp_f32<0x143, 0xc>(v);  // row_bcast31 -> rows 2 and 3
  return v;
}

__device__ __forceinline__ uint2 bf16x3(float v) {
  auto top16 = [](float x) { return __builtin_bit_cast(unsigned, x) & 0xffff0000u; };
  const unsigned t0 = top16(v);
  const float r = v - __builtin_bit_cast(float, t0);
  const unsigned t1 = top16(r);
  const unsigned t2 = top16(r - __builtin_bit_cast(float, t1));
  return uint2{(t0 >> 16) | t1, t2 >> 16};
}

// the record and aux entry of one frame from its fp32 FiLM row [g_idx | b_idx | g_norm | b_norm] (lane = shaper)
__device__ __forceinline__ void film_frag_record(const NwsWeights& w, const float* __restrict__ row, int lane, unsigned char* __restrict__ rec,
                                                 FilmAux* __restrict__ aux) {
  const float c = (float)w.lut_size * (1.0f / 6.0f);
  const float ow = w.newt_out_w[lane];
  const float ga = row[lane] * c, ba = (row[kS + lane] - w.lut_min) * c, gn = ow * row[2 * kS + lane];
  uint2* out = reinterpret_cast<uint2*>(rec);
  out[lane] = bf16x3(ga);
  out[kS + lane] = bf16x3(ba);
  out[2 * kS + lane] = bf16x3(gn);
  const float bs = wave_sum_to_lane63(ow * row[3 * kS + lane]);
  // range proof at this frame (shaping.py:136-151 clamps `lower` into the table; inside it the clamp is the identity): idx = G x + B
  // with |x| <= X[s]; one table cell of margin plus 2^-9 |G| X for every rounding between here and the lookup.  Between two
  // frames idx is the convex combination of the frames' values for the same x, and so is the margin: a pair is proven where both
  // frames are.  NaN-safe by construction: every comparison with a NaN operand is false (no fmin / fmax in the chain).
  const float X = w.exciter_bound != nullptr ? w.exciter_bound[lane] : __builtin_inff();
  const float r = fabsf(ga) * X;
  const float e = 1.0f + r * (1.0f / 512.0f);
  const unsigned long long m = __ballot((ba - r) >= e && (ba + r) <= (float)(w.lut_size - 1) - e);
  if (lane == 63) *aux = FilmAux{bs, 0.0f, m};
}

__global__ __launch_bounds__(64) void film_frags_kernel(NwsWeights w, const float* __restrict__ film, unsigned char* __restrict__ recs,
                                                        FilmAux* __restrict__ aux) {
  const size_t f = blockIdx.x;
  film_frag_record(w, film + f * NWS_FILM_CH, threadIdx.x, recs + f * NWS_FILM_REC_BYTES, aux + f);
}

// (Round 4, measured and dropped: mask-and-subtract instead - hi = bits & 0xFFFFE000, lo = v - hi, two v_cvt_pk - is two
// instructions more per pair in cheaper classes and times the same (0.2605 against 0.2598 ms); LABBOOK.md.  Pitfall met on the
// way: hipcc 7.2 reads element 0 for __builtin_bit_cast(unsigned, v.y) of an ext_vector - tools/ubench/split_probe.hip.)
// lo = fp16(v - hi) for a packed pair, hi given as fp16: v_fma_mix{lo,hi}_f16 read the fp16 operand directly and round the
// (exact) difference to fp16 on the way out: 2 instructions per pair instead of 2 cvt + 1 pk_add + 1 cvt_pk.
// (hipcc folds fma(x,-1,y) to a subtraction before it can select the mixed-precision form, hence the asm.)
__device__ __forceinline__ f16x2 split_lo2(f16x2 hi, f32x2 v) {
  const unsigned hp = __builtin_bit_cast(unsigned, hi);
  unsigned lp;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(v.x));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(v.y));
  return __builtin_bit_cast(f16x2, lp);
}

// hot-loop form: n = rint(x C_hi), then t = fma(x, C_hi, -n) - the EXACT product minus an integer, rounded once: |t| <= 1/2,
// so the rounding costs <= 3e-8 turns - and t += x C_lo.  Three packed instructions and two v_rndne_f32 per pair; the
// v_fract form before it (p, its exact rounding error e, fract(p) + (x C_lo + e)) needed four and two v_fract_f32, and
// its reduced argument in [0,1) was one bit coarser (~1.9e-7 against ~0.9e-7 absolute on the sine, tools/measure_sin.py).
// Same box, B=64, all harmonics live: 0.2232 against 0.2279 ms (hybrid-W), 0.2872 against 0.2939 ms (two-term).
__device__ __forceinline__ f32x2 sin_turns2_fract(f32x2 x) {
  const f32x2 c_hi = splat2(0.15915493667125702f), c_lo = splat2(6.4206382432985265e-09f);
  const f32x2 p = x * c_hi;
  const f32x2 n = {__builtin_rintf(p.x), __builtin_rintf(p.y)};
  const f32x2 t = fma2(x, c_lo, fma2(x, c_hi, -n));
  return f32x2{__builtin_amdgcn_sinf(t.x), __builtin_amdgcn_sinf(t.y)};
}

// two sines at once: every step except rint and v_sin_f32 is a packed-fp32 instruction
__device__ __forceinline__ f32x2 sin_turns2(f32x2 x) {
  const f32x2 c_hi = splat2(0.15915493667125702f), c_lo = splat2(6.4206382432985265e-09f);
  const f32x2 p = x * c_hi;
  const f32x2 e = fma2(x, c_hi, -p);
  const f32x2 r = {rintf(p.x), rintf(p.y)};
  const f32x2 t = (p - r) + fma2(x, c_lo, e);
  return f32x2{__builtin_amdgcn_sinf(t.x), __builtin_amdgcn_sinf(t.y)};
}

// v = hi + lo with hi, lo fp16 (lo exact residual rounded to fp16: ~2^-22 relative)
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// DBG != 0 instantiations exist only for nws_debug_exciter_newt (ablation timing; results are wrong by design):
//   1: sin() replaced by its argument (scaled into the range proof's domain)   2: LUT gather skipped   3: whole FiLM/shaper tail skipped
//   4: MFMAs skipped   5: prologue only   6: no global load in front of the barrier (7 / 8 / 9: no FiLM rows / no fragment DMA / no F0, carry,
//   shifts): what the prologue's memory LATENCY costs the launch
// second launch-bound = minimum waves per SIMD: without it hipcc hoists all 32 LUT gathers of the tail, takes 256
// VGPRs and drops the kernel to 1 wave/SIMD (measured 2x slower); 4 waves/SIMD = 128 VGPRs, LDS allows 5 blocks/CU
// HPB = hops (128-sample tiles) per workgroup = 4 HPB waves.  Two hops share one copy of the 28 KB fragment table and one
// set of phase shifts: 33 KB of LDS per 8 waves instead of 31.5 KB per 4 -> 6 waves per SIMD instead of 5 (0.322 -> 0.304 ms).
// The bound of 7 waves/SIMD (72 VGPRs) is deliberate: at 8 (64 VGPRs) hipcc spills 24 B/lane to scratch, which is slower
// (0.328 ms).  (Wrong results seen with that build under two overlapping audio streams were first blamed on the scratch;
// the multi-stream problem turned out to be independent of it, see pipeline.py.  The build still rejects scratch.)
template <int MODE, int DBG = 0, int HPB = 1, int OPT = 0>
__global__ __launch_bounds__(256 * HPB, MODE == kModeExact ? 2 : (is_bank(MODE) ? 4 : ((OPT & kOptLowReg) ? 6 : (OPT & kOptFilmMfma) ? 5 : (HPB == 2 ? 7 : 5)))) void exciter_newt_kernel(NwsWeights w, const float* __restrict__ f0,
                                                           const float* __restrict__ f0_up,
                                                           const double* __restrict__ carry,
                                                           const float* __restrict__ phase_u,
                                                           const float* __restrict__ rand_phase,
                                                           const float* __restrict__ film, int T, float sample_rate,
                                                           float* __restrict__ exciter_out,
                                                           float* __restrict__ newt_out,
                                                           const float* __restrict__ bank = nullptr,
                                                           const float* __restrict__ add_in = nullptr,
                                                           const int xcd_groups = 0, const int xcd_aux = 0 /* kOptFilmDma: B T */) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ExcLds& L = *reinterpret_cast<ExcLds*>(smem_raw);
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw + ((sizeof(ExcLds) + 15) & ~size_t(15)));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int col = lane & 31;
  // Workgroups go to the eight XCDs round-robin in launch order (linear id % 8, MI355X_MICROARCH).  Neighbouring hop groups
  // share two of their four FiLM rows; as grid (groups, B) with jb = blockIdx.x HPB they always sat on different L2s and every
  // row came from HBM twice (105 MB per launch against 69 MB algorithmic).  xcd_groups > 0 (= hop groups per utterance): a 1-D
  // grid whose id p runs on XCD p % 8 as that XCD's (p / 8)-th workgroup; XCD k takes the k-th of eight contiguous, equally
  // long (+-1) ranges of the (utterance, hop group) sequence, so neighbours in time are neighbours in launch order on ONE L2.
  // Placement only: same results.
  int bx, b;
  if (xcd_groups > 0) {
    const unsigned p = blockIdx.x, total = gridDim.x, k = p & 7u, r = total & 7u;
    const unsigned l = k * (total >> 3) + (k < r ? k : r) + (p >> 3);
    b = __builtin_amdgcn_readfirstlane((int)(l / (unsigned)xcd_groups));   // (uniform: keep it in a scalar register)
    bx = (int)l - b * xcd_groups;
  } else {
    bx = (int)blockIdx.x;
    b = (int)blockIdx.y;
  }
  const int jb = bx * HPB;                 // first hop of the workgroup
  const int j = jb + (wave >> 2);          // this wave's hop
  const int w4 = wave & 3;                 // quarter of the hop
  const int N = T * NWS_HOP;
  constexpr int kThreads = 256 * HPB;

  // this lane's own inputs (two F0 frames or the upsampled sample, the 32-sample carry) are requested FIRST, in front of the
  // staging traffic below: the VM counter retires in order, so with these loads behind the staging ones the prologue paid
  // three dependent round trips to memory (staging -> F0 -> carry) before its barrier; now they are all in flight together
  const bool hop_live = HPB == 1 || j < T;
  const int n = (hop_live ? j : jb) * kTile + w4 * 32 + col;
  const NwsLerp lc = nws_lerp_coeff(n, T);
  float f0_a, f0_b = 0.0f;
  if (DBG == 6 || DBG == 9) {
    f0_a = 0.3f + 1.0e-4f * (float)col;
    f0_b = 0.31f;
  } else
  if (f0_up != nullptr) {
    f0_a = f0_up[(size_t)b * N + n];
  } else {
    const float* x = f0 + (size_t)b * T;
    f0_a = x[lc.i0];
    f0_b = x[lc.i1];
  }
  const double carry_in = (DBG == 6 || DBG == 9) ? 0.4 * (double)n : carry[(size_t)b * (N / 32) + (n >> 5)];

  // ---- stage the workgroup constants in LDS ----
  if (DBG == 6 || DBG == 8) {
    for (int e = tid; e < (int)(sizeof(L.whi) + sizeof(L.wlo)) / 16; e += kThreads) reinterpret_cast<uint4*>(L.whi)[e] = uint4{0x2e662e66u, 0x2e662e66u, 0x2e662e66u, 0x2e662e66u};
  } else
  if (w.mixer_frags != nullptr) {
    // pre-split fragment table (nws_mixer_frags): 28 KB = 28 pieces of 1 KB, copied by the LDS-DMA path
    // (global_load_lds_dwordx4: per-lane global address, wave-uniform LDS base + 16 B per lane; no VGPR round trip).
    // whi and wlo are contiguous; the workgroup barrier below drains the DMA (vmcnt) before anybody reads.
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // the instruction's immediate offset (< 4 KB) moves the global and the LDS address together: one address set-up serves
    // four loads.  4 waves: pieces 7w .. 7w+6;  8 waves: pieces 4w .. 4w+3 (w < 7)
    if (HPB == 1) {
      const char* src = static_cast<const char*>(w.mixer_frags) + wave * 7168 + lane * 16;
      char* dst = reinterpret_cast<char*>(L.whi) + wave * 7168;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 3072, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + 4096), (lptr_t)(dst + 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + 4096), (lptr_t)(dst + 4096), 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(src + 4096), (lptr_t)(dst + 4096), 16, 2048, 0);
    } else if (wave < ((OPT & kOptHybridW) ? 4 : 7)) {
      // (kOptHybridW reads W_hi and the K-step-0 part of W_lo only: the first 16 of the 28 KB)
      const char* src = static_cast<const char*>(w.mixer_frags) + wave * 4096 + lane * 16;
      char* dst = reinterpret_cast<char*>(L.whi) + wave * 4096;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 2048, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 3072, 0);
    }
  } else {
    _Float16* whi = reinterpret_cast<_Float16*>(L.whi);
    _Float16* wlo = reinterpret_cast<_Float16*>(L.wlo);
    for (int e = tid; e < kS * kKPad; e += kThreads) {
      const int s = e / kKPad, kk = e - s * kKPad;
      const float wv = mixer_slot_weight(w.mixer_w, w.mixer_b, s, kk);
      const int at = mixer_frag_index(s, kk);
      split_f16(wv, whi[at], wlo[at]);
    }
  }
  // FiLM slots (one shaper per lane), bias sums, phase shifts and harmonic numbers, spread over the waves:
  //   4 waves: 0/1 slots 0/1, 2 bsum[0..1], 3 bsum[2] + shifts;   8 waves: 0..2 slots, 3..6 bsum[0..3], 7 shifts
  constexpr int kSlots = HPB + 1;
  constexpr bool kFilmDma = (OPT & kOptFilmDma) != 0;
  static_assert(!kFilmDma || (HPB == 2 && (OPT & kOptFilmMfma) && (OPT & kOptLowReg) && MODE == kModeLutPairsDiv6),
                "kOptFilmDma: the default two-hop LUT kernel only");
  if (kFilmDma) {
    // the four frames' fragment records (`bank` carries their base here: records of all B T frames, then the aux entries): 6 KB =
    // six 1 KB pieces, one per wave 0..5, straight to LDS.  A piece straddles records, and clamped frames at the utterance's
    // ends break the contiguity: every lane finds its own 16 source bytes
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned char* recs = reinterpret_cast<const unsigned char*>(bank) + (size_t)b * T * NWS_FILM_REC_BYTES;
    auto frame_of = [&](int q) {
      const int f = jb - 1 + q;
      return f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
    };
    if (wave < 6) {
      const int o = wave * 1024 + lane * 16;
      const int q = (o >= NWS_FILM_REC_BYTES) + (o >= 2 * NWS_FILM_REC_BYTES) + (o >= 3 * NWS_FILM_REC_BYTES);
      const unsigned char* src = recs + (size_t)frame_of(q) * NWS_FILM_REC_BYTES + (o - q * NWS_FILM_REC_BYTES);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&L.frec[0][0] + wave * 1024), 16, 0, 0);
    } else if (wave == 6 && lane < 4) {
      const FilmAux* aux = reinterpret_cast<const FilmAux*>(reinterpret_cast<const unsigned char*>(bank) + (size_t)xcd_aux * NWS_FILM_REC_BYTES);
      const FilmAux a = aux[(size_t)b * T + frame_of(lane)];
      L.bsum[lane] = a.bsum;
      L.fmask[lane] = a.mask;
    }
  } else
  if (MODE != kModeExciterOnly) {
    const float* fb = film + (size_t)b * T * NWS_FILM_CH;
    auto frame_of = [&](int q) {
      const int f = jb - 1 + q;
      return f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
    };
    auto bias_sum = [&](int q) {  // sum_s out_w[s] * b_norm[frame q][s]
      const float v = wave_sum_to_lane63((DBG == 6 || DBG == 7) ? 1.0e-3f * (float)lane : w.newt_out_w[lane] * fb[(size_t)frame_of(q) * NWS_FILM_CH + 3 * kS + lane]);
      if (lane == 63) L.bsum[q] = v;
    };
    if (wave < kSlots) {
      const float* r0 = fb + (size_t)frame_of(wave) * NWS_FILM_CH + lane;
      const float* r1 = fb + (size_t)frame_of(wave + 1) * NWS_FILM_CH + lane;
      const float ow = (DBG == 6 || DBG == 7) ? 0.02f : w.newt_out_w[lane];
      if (OPT & kOptFilmMfma) {
        // index FiLM pre-scaled to table units: idx = (size/6) (g x + b - min) = g' x + b'  (a few ulp of idx away from the
        // reference's rounding chain, like the folded origin of the kModeLutPairsDiv6 path; held to the same e2e parity bar)
        const float c = (float)w.lut_size * (1.0f / 6.0f);
        float ga = 0.0f, gd = 0.0f, ba = 0.0f, bd = 0.0f;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          const float u0 = (DBG == 6 || DBG == 7) ? 0.3f + 0.001f * (float)(lane + ty) : r0[ty * kS], u1 = (DBG == 6 || DBG == 7) ? 0.31f + 0.001f * (float)(lane + ty) : r1[ty * kS];
          float a, d;
          if (ty == 0) {
            a = u0 * c;
            d = (u1 - u0) * c;
            ga = a;
            gd = d;
          } else if (ty == 1) {
            a = (u0 - w.lut_min) * c;
            d = (u1 - u0) * c;
            ba = a;
            bd = d;
          } else {
            a = ow * u0;
            d = ow * u1 - a;
          }
          // exact three-term bf16 split by truncation: t0 = top 16 bits, the remainder is exact in fp32
          auto top16 = [](float v) { return __builtin_bit_cast(unsigned, v) & 0xffff0000u; };
          const unsigned a0 = top16(a);
          const float ra = a - __builtin_bit_cast(float, a0);
          const unsigned a1 = top16(ra);
          const unsigned a2 = top16(ra - __builtin_bit_cast(float, a1));
          const unsigned d0 = top16(d);
          const float rd = d - __builtin_bit_cast(float, d0);
          const unsigned d1 = top16(rd);
          const unsigned d2 = top16(rd - __builtin_bit_cast(float, d1));
          L.ffrag[wave][ty][lane >> 5][lane & 31] = uint4{(a0 >> 16) | a1, (a2 >> 16) | d0, (d1 >> 16) | d2, 0u};
        }
        if ((OPT & kOptLowReg) && MODE == kModeLutPairsDiv6) {
          // Range proof for this slot's lookups (shaping.py:136-151 clamps `lower` into the table; inside the table the clamp
          // is the identity).  idx = G x + B with G, B linear in the interpolation weight between the slot's two frames and
          // |x| <= X[s] whatever the oscillator does, so idx lies between the extremes taken at the two frames:
          // B_f -+ |G_f| X.  One table cell of margin on either side plus 2^-9 of |G| X covers every rounding between here and
          // the tail (the three-term FiLM interpolation, the 22-bit - or, opted in, 11-bit - mixer products).  A NaN anywhere
          // must end as "not proven": fminf / fmaxf DROP a NaN operand (a NaN gain or bias of the right frame alone would vanish
          // from lo / hi), hence the explicit self-comparisons - a FiLM frame with a NaN always takes the clamped form.
          const float X = (DBG == 6 || DBG == 7) ? 2.0f : (w.exciter_bound != nullptr ? w.exciter_bound[lane] : __builtin_inff());
          const float r_a = fabsf(ga) * X, r_b = fabsf(ga + gd) * X;
          const float e = 1.0f + fmaxf(r_a, r_b) * (1.0f / 512.0f);
          const float lo = fminf(ba - r_a, (ba + bd) - r_b), hi = fmaxf(ba + r_a, (ba + bd) + r_b);
          const float right = (ba + bd) + r_b;     // NaN iff the right frame's gain, bias or bound is
          const unsigned long long okm = __ballot(lo >= e && hi <= (float)(w.lut_size - 1) - e && r_a == r_a && right == right);
          if (lane == 0) L.okmask[wave] = okm;
        }
      } else
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        const float sc = ty == 2 ? ow : 1.0f;
        const float v0 = sc * r0[ty * kS], v1 = sc * r1[ty * kS];
        // hot LUT path: the table origin (x - lut_min) rides on the first FiLM bias, one packed add less per shaper pair
        L.fa[wave][ty][lane] = (MODE == kModeLutPairsDiv6 && ty == 1) ? v0 - w.lut_min : v0;
        L.fd[wave][ty][lane] = v1 - v0;
      }
    } else if (HPB == 1) {
      if (wave == 2) {
        bias_sum(0);
        bias_sum(1);
      } else {
        bias_sum(2);
      }
    } else if (wave < 7) {
      bias_sum(wave - 3);
    }
  }
  if (wave == 4 * HPB - 1) {
    // _create_phase_shift (generators.py:54-56): fl(fl(u * rand_phase) - fl32(pi)) for harmonic c = slot c
    auto shift_of = [&](int c) { return (DBG == 6 || DBG == 9) ? 0.01f * (float)c : (c >= 1 && c <= kK ? phase_u[c - 1] * rand_phase[c - 1] - kPi : 0.0f); };
    L.shift[lane] = shift_of(lane);
    L.kf[lane] = (float)lane;
    if (lane < kKPad - 64) {
      L.shift[64 + lane] = shift_of(64 + lane);
      L.kf[64 + lane] = (float)(64 + lane);
    }
  }
  if (MODE == kModeExact) load_shaper_lds<true>(SH, w, tid, kThreads);

  // ---- per-sample phase: fp64 prefix sum -> fp32 rounding chain of the reference ----
  // (a second hop past the end of an odd-length utterance only helped with the staging above)
  const float f0n = f0_up != nullptr ? f0_a : nws_lerp(f0_a, f0_b, lc.w0, lc.w1);
  const double cs_local = scan32_f64((double)f0n);  // inclusive prefix sum over the wave's 32 samples (both halves alike)
  double cs = cs_local;
  cs += carry_in;
  const float csum = (float)cs;                                 // fl32 of the double prefix sum
  // math.tau * cumsum / sample_rate with a TRUE division: for sr = 16000 the reciprocal + one FMA correction below is the
  // correctly rounded quotient for every fp32 numerator (checked exhaustively over all mantissas), 3 instructions
  const float tc = kTau * csum;
  float phase;
  if (sample_rate == 16000.0f) {
    const float q = tc * 6.25e-05f;   // fl32(1/16000)
    phase = fmaf(fmaf(-q, 16000.0f, tc), 6.25e-05f, q);
  } else {
    phase = __fdiv_rn(tc, sample_rate);
  }
  const float nyquist = sample_rate * 0.5f;

  // the fragment table travels by LDS-DMA, which only the issuing wave's VM counter tracks: every wave drains its own
  // before the barrier (hipcc happens to place such a wait here anyway, for the carry load; this makes it a guarantee)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!hop_live) return;
  if (DBG == 5) {   // prologue only (timing): what staging + phase + the barrier cost per launch
    if (half == 0) newt_out[(size_t)b * N + n] = phase + L.shift[lane] + L.bsum[0];
    return;
  }

  // ---- 101 harmonics -> 64 shapers on the matrix cores (fp16 two-term split, fp32 accumulate) ----
  // K-step ks covers harmonics 16ks+1 .. 16ks+16; lane (col, half) evaluates the 8 sines of harmonics
  // 16ks + 8half + 1..8 for its sample: exactly the B fragment of v_mfma_f32_32x32x16_f16.
  f32x16 acc0, acc1;
  // no sine argument of this wave can exceed the fast reduction's range -> packed sines without range tests
  const bool small_args = __all(fabsf(phase) * (float)kKPad + 4.0f < 6.0e6f);
  // anti-alias mask (generators.py:50-52): harmonic k is live iff fl(f0*k) < sr/2.  fl(f0*k) is monotone in k for
  // f0 > 0, so the live set is a prefix 1..kmax; count it once per lane with the exact comparison.
  // every harmonic of every sample of the wave below Nyquist (e.g. the timing script's sub-1 Hz "F0"): no masks, no counting
  const bool all_live = __all((f0n * (float)kK) < nyquist);
  int kmax;
  if (all_live) {
    kmax = kK;
  } else if (!(f0n > 0.0f)) {
    kmax = f0n == f0n ? kK : 0;  // f0 <= 0: every product is <= 0 < sr/2;  NaN: nothing is live
  } else {
    // the quotient estimate is within one of the answer (both are small integers); settle it with the exact test
    const float q = nyquist * __builtin_amdgcn_rcpf(f0n);
    int kc = q > (float)kK ? kK : (int)q;
    kc += (kc < kK && (f0n * (float)(kc + 1)) < nyquist) ? 1 : 0;
    kc -= (kc > 0 && !((f0n * (float)kc) < nyquist)) ? 1 : 0;
    kc -= (kc > 0 && !((f0n * (float)kc) < nyquist)) ? 1 : 0;
    kmax = kc;
  }
  const int frag_lane = half * 32 + col;
  const f32x2 ph2 = splat2(phase);
  // rint(e P) for e = 0..7, P = phase / 2 pi = turns per harmonic number (shared integer parts of the sine reduction, below)
  // (the FiLM-on-the-matrix-pipe variants, i.e. the product path; the round-1 VALU-FiLM form at 72 registers has no room)
  constexpr bool kSharedTurns = (OPT & kOptFilmMfma) != 0;
  f32x2 turns_e[4];
  if (kSharedTurns) {
    const float P = phase * 0.15915493667125702f;
#pragma unroll
    for (int p = 0; p < 4; ++p) turns_e[p] = f32x2{__builtin_rintf((float)(2 * p) * P), __builtin_rintf((float)(2 * p + 1) * P)};
  }
  // one K-step; the first one starts the accumulators from the MFMA's inline-zero C operand (no 32 v_mov per wave)
  // how many fp16 terms the sines of K-step ks travel as (compile-time after unrolling): see Opt
  auto two_terms = [](const int ks) { return (OPT & kOptOneTerm) ? false : ((OPT & kOptHybrid) ? ks == 0 : true); };
  // weights: W_hi and W_lo everywhere, or (kOptHybridW) W_lo in K-step 0 only
  auto two_wterms = [](const int ks) { return (OPT & kOptHybridW) ? ks == 0 : true; };
  // sines of one pair of K slots: arg = fl(fl(k*phase) + shift), the reference's own rounding chain, then sin
  auto sine_pair = [&](const f32x2 kfp, const f32x2 shp) -> f32x2 {
    if (DBG == 1) return (kfp * ph2 + shp) * splat2(0x1p-40f);   // (bounded: the range-proven lookups trust |x| <= X)
    if (small_args) return sin_turns2_fract(kfp * ph2 + shp);
    const f32x2 arg2 = kfp * ph2 + shp;
    return f32x2{nws_sin_wide(arg2.x), nws_sin_wide(arg2.y)};
  };
  auto sines = [&](const int ks, auto first_tag, f16x8& vhi, f16x8& vlo) {
    constexpr bool kFirst = decltype(first_tag)::value;
    const int kk0 = 16 * ks + 8 * half;
    const int rem = kmax + 1 - kk0;    // this lane's live slots in the step: e < rem  (slot c = kk0 + e is live iff c <= kmax)
    const bool full = all_live || __all(rem >= 8);
    const float4 sh0 = *reinterpret_cast<const float4*>(&L.shift[kk0]);
    const float4 sh1 = *reinterpret_cast<const float4*>(&L.shift[kk0 + 4]);
    const float4 kf0 = *reinterpret_cast<const float4*>(&L.kf[kk0]);
    const float4 kf1 = *reinterpret_cast<const float4*>(&L.kf[kk0 + 4]);
    const f32x2 sh2[4] = {{sh0.x, sh0.y}, {sh0.z, sh0.w}, {sh1.x, sh1.y}, {sh1.z, sh1.w}};
    const f32x2 kf2[4] = {{kf0.x, kf0.y}, {kf0.z, kf0.w}, {kf1.x, kf1.y}, {kf1.z, kf1.w}};
    f32x2 v2[4];
    if (kSharedTurns && small_args && DBG != 1) {
      // the lane's 8 consecutive harmonics: turns(k0 + e) = turns(k0) + e P + (shift differences, |.| < 1 turn), so
      // n_e = rint(x_0 C_hi) + rint(e P) is within 2 of every x_e C_hi: one rint per K-step instead of eight, the reduced
      // argument t_e = fma(x_e, C_hi, -n_e) + x_e C_lo is still the exact product minus an integer, now |t_e| < 2 (fp32
      // spacing 2.4e-7 turns at worst; v_sin_f32 takes +-256 turns).  Against the per-sine rint of sin_turns2_fract:
      // 0.2061 instead of 0.2212 ms (hybrid-W), 0.2713 instead of 0.2889 (two-term), outputs 5e-9 RMS apart (signal 2.9e-3)
      const f32x2 c_hi = splat2(0.15915493667125702f), c_lo = splat2(6.4206382432985265e-09f);
      f32x2 a2[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) a2[p] = kf2[p] * ph2 + sh2[p];
      const f32x2 n0 = splat2(__builtin_rintf(a2[0].x * 0.15915493667125702f));
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const f32x2 t = fma2(a2[p], c_lo, fma2(a2[p], c_hi, -(n0 + turns_e[p])));
        v2[p] = f32x2{__builtin_amdgcn_sinf(t.x), __builtin_amdgcn_sinf(t.y)};
      }
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p) v2[p] = sine_pair(kf2[p], sh2[p]);
    }
    if (!full) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        v2[p].x = 2 * p < rem ? v2[p].x : 0.0f;
        v2[p].y = 2 * p + 1 < rem ? v2[p].y : 0.0f;
      }
    }
    if (kFirst && DBG != 1) v2[0].x = half == 0 ? 1.0f : v2[0].x;  // slot 0: the bias' constant input
    // v = hi + lo, both fp16 (lo = exact residual rounded to fp16): v_cvt_pk_f16_f32 for hi, one v_fma_mix{lo,hi}_f16 per lo
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f16x2 h2 = __builtin_convertvector(v2[p], f16x2);
      vhi[2 * p] = h2.x;
      vhi[2 * p + 1] = h2.y;
      if (two_terms(ks)) {
        const f16x2 l2 = split_lo2(h2, v2[p]);
        vlo[2 * p] = l2.x;
        vlo[2 * p + 1] = l2.y;
      }
    }
  };
  auto mix = [&](const int ks, auto first_tag, const f16x8& vhi, const f16x8& vlo) {
    constexpr bool kFirst = decltype(first_tag)::value;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f16x8 ahi = L.whi[(ks * 2 + m) * 64 + frag_lane];
      f32x16& acc = m == 0 ? acc0 : acc1;
      if (DBG == 4) {
        const f16x8 alo = L.wlo[(ks * 2 + m) * 64 + frag_lane];
        if (kFirst) acc = f32x16{};
        acc[ks] += (float)ahi[0] * (float)vhi[0] + (float)alo[1] * (float)vlo[1];
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, vhi, kFirst ? f32x16{} : acc, 0, 0, 0);
        if (two_terms(ks)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, vlo, acc, 0, 0, 0);
        if (two_wterms(ks)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(L.wlo[(ks * 2 + m) * 64 + frag_lane], vhi, acc, 0, 0, 0);
      }
    }
  };
  // K-step 6 = slots 96..103 (harmonics 96..101): ONE K=8 MFMA per term instead of a K=16 one whose upper half would be
  // padding - half the sines of a full step.  Lane (col, half) evaluates slots 96 + 4 half + 0..3.
  auto last_step = [&] {
    constexpr int ks = kKSteps - 1;
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    const int kk0 = 16 * ks + 4 * half;
    const int rem = kmax + 1 - kk0;
    const bool full = all_live || __all(rem >= 4);
    const float4 sh = *reinterpret_cast<const float4*>(&L.shift[kk0]);
    const float4 kf = *reinterpret_cast<const float4*>(&L.kf[kk0]);
    f32x2 v2[2];
    if (kSharedTurns && small_args && DBG != 1) {
      const f32x2 c_hi = splat2(0.15915493667125702f), c_lo = splat2(6.4206382432985265e-09f);
      const f32x2 a2[2] = {f32x2{kf.x, kf.y} * ph2 + f32x2{sh.x, sh.y}, f32x2{kf.z, kf.w} * ph2 + f32x2{sh.z, sh.w}};
      const f32x2 n0 = splat2(__builtin_rintf(a2[0].x * 0.15915493667125702f));
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const f32x2 t = fma2(a2[p], c_lo, fma2(a2[p], c_hi, -(n0 + turns_e[p])));
        v2[p] = f32x2{__builtin_amdgcn_sinf(t.x), __builtin_amdgcn_sinf(t.y)};
      }
    } else {
      v2[0] = sine_pair(f32x2{kf.x, kf.y}, f32x2{sh.x, sh.y});
      v2[1] = sine_pair(f32x2{kf.z, kf.w}, f32x2{sh.z, sh.w});
    }
    if (!full) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        v2[p].x = 2 * p < rem ? v2[p].x : 0.0f;
        v2[p].y = 2 * p + 1 < rem ? v2[p].y : 0.0f;
      }
    }
    f16x4 vhi, vlo;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const f16x2 h2 = __builtin_convertvector(v2[p], f16x2);
      vhi[2 * p] = h2.x;
      vhi[2 * p + 1] = h2.y;
      if (two_terms(ks)) {
        const f16x2 l2 = split_lo2(h2, v2[p]);
        vlo[2 * p] = l2.x;
        vlo[2 * p + 1] = l2.y;
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f16x4 ahi = *reinterpret_cast<const f16x4*>(&L.whi[(ks * 2 + m) * 64 + frag_lane]);   // first 8 bytes of the row
      f32x16& acc = m == 0 ? acc0 : acc1;
      if (DBG == 4) {
        acc[ks] += (float)ahi[0] * (float)vhi[0];
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ahi, vhi, acc, 0, 0, 0);
        if (two_terms(ks)) acc = __builtin_amdgcn_mfma_f32_32x32x8f16(ahi, vlo, acc, 0, 0, 0);
        if (two_wterms(ks))
          acc = __builtin_amdgcn_mfma_f32_32x32x8f16(*reinterpret_cast<const f16x4*>(&L.wlo[(ks * 2 + m) * 64 + frag_lane]), vhi, acc, 0, 0, 0);
      }
    }
  };
  // the first step always runs (it carries the bias); k*f0 only grows with k: once a step has no live lane, everything
  // above is masked too
  {
    auto kstep = [&](const int ks, auto first_tag) {
      f16x8 vhi, vlo;
      sines(ks, first_tag, vhi, vlo);
      mix(ks, first_tag, vhi, vlo);
    };
    kstep(0, std::true_type{});
    bool more = true;
#pragma unroll
    for (int ks = 1; ks < kKSteps - 1; ++ks) {
      if (!all_live && !__any(kmax + 1 - (16 * ks + 8 * half) > 0)) {
        more = false;
        break;
      }
      kstep(ks, std::false_type{});
    }
    if (more && (all_live || __any(kmax + 1 - (16 * (kKSteps - 1) + 4 * half) > 0))) last_step();
  }

  // accumulator element r of M-tile m: shaper 32m + (r&3) + 8(r>>2) + 4*half, sample `col`
  if (exciter_out != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s0 = (r & 3) + 8 * (r >> 2) + 4 * half;
      exciter_out[((size_t)b * kS + s0) * N + n] = acc0[r];
      exciter_out[((size_t)b * kS + s0 + 32) * N + n] = acc1[r];
    }
  }
  if (MODE == kModeExciterOnly) return;
  if (DBG == 3) {
    float t = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += acc0[r] + acc1[r];
    t += nws_swap_halves(t);
    if (half == 0) newt_out[(size_t)b * N + n] = t;
    return;
  }

  // ---- FiLM -> shaper -> FiLM -> 64->1 mix, all in registers ----
  const int q0 = lc.i0 - (jb - 1);  // slot of the left frame; fd[q0] is zero where the right frame is clamped
  LutParams LP;
  LutFast LF;
  if (MODE == kModeLutPairsDiv6) LF = make_lut_fast(w);
  else if (is_lut(MODE)) LP = make_lut_params(w);
  const int lane_row_off = is_lut(MODE) ? 4 * half * w.lut_size : 0;
  const unsigned lane_off_bytes = is_lut(MODE) ? (unsigned)lane_row_off * 8u : 0u;
  if (MODE == kModeLutPairsDiv6 && (OPT & kOptFilmMfma)) {
    // FiLM parameters of the wave's 64 shapers x 32 samples from the matrix pipe: a + w d per (shaper, sample) with the
    // accumulator layout of the exciter tile itself, so G[r], Bb[r], Gn[r] pair up with acc[r] register for register.
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const unsigned w1b = __builtin_bit_cast(unsigned, lc.w1);   // a multiple of 1/256 in [0, 1): exact in bf16
    uint4 bop;
    if (kFilmDma) {
      // half 0 weighs the left frame's three terms with 1 - w, half 1 the right frame's with w (K slots 8 h + 0..2); 1 - w is
      // a multiple of 1/256 in (0, 1] like w: exact in bf16
      const unsigned wb = (half == 0 ? __builtin_bit_cast(unsigned, 1.0f - lc.w1) : w1b) >> 16;
      bop = uint4{wb | (wb << 16), wb, 0u, 0u};
    } else {
      bop = half == 0 ? uint4{0x3f803f80u, 0x3f80u | (w1b & 0xffff0000u), (w1b >> 16) | (w1b & 0xffff0000u), 0u}
                      : uint4{0u, 0u, 0u, 0u};       // K slots 8..15 unused: zero B, whatever A holds there
    }
    const bf16x8 bfrag = __builtin_bit_cast(bf16x8, bop);
    // A fragment of parameter type ty, M-tile m
    auto ffrag_of = [&](const int ty, const int m) -> bf16x8 {
      if (kFilmDma) {
        const uint2 v = *reinterpret_cast<const uint2*>(&L.frec[0][0] + (q0 + half) * NWS_FILM_REC_BYTES + ((ty * 2 + m) * 32 + col) * 8);
        return __builtin_bit_cast(bf16x8, uint4{v.x, v.y, 0u, 0u});
      }
      return __builtin_bit_cast(bf16x8, L.ffrag[q0][ty][m][col]);
    };
    float part = 0.0f;
    unsigned ok_tile[2] = {0u, 0u};   // bit k of ok_tile[m]: shaper 32 m + k proven in range for this wave's samples
    if (OPT & kOptLowReg) {
      const int q0u = __builtin_amdgcn_readfirstlane(q0);
      if (__all(q0 == q0u)) {           // (a wave's 32 samples share their frame pair; anything else takes the clamped form)
        const unsigned long long okm = kFilmDma ? (L.fmask[q0u] & L.fmask[q0u + 1]) : L.okmask[q0u];
        ok_tile[0] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)okm);
        ok_tile[1] = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(okm >> 32));
      }
    }
    if (OPT & kOptLowReg) {
      // Register diet (93 -> <= 80 VGPRs: a third 8-wave workgroup fits a CU, 6 waves per SIMD instead of 4).  Same arithmetic,
      // other order: the index FiLM of a whole M-tile first, IN PLACE of the accumulator tile (acc, G, Bb live: 48 + the other
      // tile's 16), only then the gain tile Gn (its MFMA runs under the first gathers) and the lookups.
      // Two copies of a tile's sixteen lookups, chosen ONCE per tile by a wave-uniform test behind the tile's three MFMAs (a branch per
      // group of four lookups cost the register allocator the in-place tiles: 218 spilled registers; a branch around the whole tile
      // body had the FiLM stage hoisted out of both arms into fresh registers: 12 spilled): `proven` = all 32 shapers of the tile
      // provably index inside the table for this wave's samples -> floor(idx) is the table cell and idx - floor(idx) = fract(idx)
      // exactly (below 2^23): v_fract + v_cvt_u32 + v_lshl_add where the clamped form of shaping.py:136-151 needs floor, med3,
      // cvt, sub, lshl_add.  Same bits wherever both apply.
      auto lookups = [&](auto m_tag, auto proven_tag, const f32x16& acc, const f32x16& Gn) {
        constexpr int m = decltype(m_tag)::value;
        constexpr bool proven = decltype(proven_tag)::value;
#pragma unroll
        for (int g = 0; g < 16 / kLutGroup; ++g) {
          float fr[kLutGroup];
          float2 tv[kLutGroup];
#pragma unroll
          for (int e = 0; e < kLutGroup; ++e) {
            const int r = kLutGroup * g + e;
            const float idx = acc[r];
            unsigned o;
            if (proven) {
              o = lane_off_bytes + ((unsigned)idx << 3);
              fr[e] = __builtin_amdgcn_fractf(idx);
            } else {
              const float fl = __builtin_amdgcn_fmed3f(floorf(idx), 0.0f, LF.top);
              o = lane_off_bytes + ((unsigned)(int)fl << 3);
              fr[e] = idx - fl;
            }
            tv[e] = DBG == 2 ? float2{__builtin_bit_cast(float, o), fr[e]}
                             : *reinterpret_cast<const float2*>(LF.pairs + (size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * LF.row_bytes + o);
          }
#pragma unroll
          for (int e = 0; e < kLutGroup; ++e) part = fmaf(Gn[kLutGroup * g + e], fmaf(tv[e].y, fr[e], tv[e].x), part);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto tile = [&](auto m_tag) {
        constexpr int m = decltype(m_tag)::value;
        f32x16& acc = m == 0 ? acc0 : acc1;
        {
          const f32x16 G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag_of(0, m), bfrag, f32x16{}, 0, 0, 0);
          const f32x16 Bb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag_of(1, m), bfrag, f32x16{}, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = fmaf(G[r], acc[r], Bb[r]);   // FiLM in table units (bias and origin folded at staging)
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 Gn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ffrag_of(2, m), bfrag, f32x16{}, 0, 0, 0);
        // the tile's sixteen lookups per lane in one of two forms, chosen by ONE wave-uniform branch
        if (ok_tile[m] == 0xffffffffu) lookups(m_tag, std::true_type{}, acc, Gn);
        else lookups(m_tag, std::false_type{}, acc, Gn);
      };
      tile(std::integral_constant<int, 0>{});
      tile(std::integral_constant<int, 1>{});
    } else
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const f32x16& acc = m == 0 ? acc0 : acc1;
      const f32x16 G = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, L.ffrag[q0][0][m][col]), bfrag, f32x16{}, 0, 0, 0);
      const f32x16 Bb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, L.ffrag[q0][1][m][col]), bfrag, f32x16{}, 0, 0, 0);
      const f32x16 Gn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, L.ffrag[q0][2][m][col]), bfrag, f32x16{}, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 16 / kLutGroup; ++g) {
        float fr[kLutGroup];
        float2 tv[kLutGroup];
#pragma unroll
        for (int e = 0; e < kLutGroup; ++e) {
          const int r = kLutGroup * g + e;
          const float idx = fmaf(G[r], acc[r], Bb[r]);   // FiLM in table units (bias and origin folded at staging)
          // floor + clamp + integer index without a conversion: idx + (2^23 - 1/2) rounds to 2^23 + floor(idx) (at exact
          // integers k possibly k - 1 with fraction 1: the same point of the piecewise-linear table up to one rounding of
          // T), the clamp works on that float, and its low mantissa bits ARE the index
          // (the add-2^23 trick for floor + index was measured: no faster than floor / med3 / convert, 0.2406 vs 0.2384 ms)
          const float fl = __builtin_amdgcn_fmed3f(floorf(idx), 0.0f, LF.top);
          const unsigned o = lane_off_bytes + ((unsigned)(int)fl << 3);
          fr[e] = idx - fl;
          tv[e] = DBG == 2 ? float2{idx, 0.0f}
                           : *reinterpret_cast<const float2*>(LF.pairs + (size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * LF.row_bytes + o);
        }
#pragma unroll
        for (int e = 0; e < kLutGroup; ++e) part = fmaf(Gn[kLutGroup * g + e], fmaf(tv[e].y, fr[e], tv[e].x), part);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const float bias_n = fmaf(lc.w1, L.bsum[q0 + 1] - L.bsum[q0], L.bsum[q0]);
    const float total = part + nws_swap_halves(part) + (bias_n + w.newt_out_b[0]);
    if (half == 0) newt_out[(size_t)b * N + n] = add_in != nullptr ? add_in[(size_t)b * N + n] + total : total;
    return;
  }
  if (is_bank(MODE)) {
    static_assert(sizeof(BankLds) <= sizeof(L.whi) + sizeof(L.wlo), "the shaper-input planes take the place of the fragment tables");
    BankLds& BK = *reinterpret_cast<BankLds*>(smem_raw);
    const f32x2 w1_b = splat2(lc.w1);
    // bank passes: lane = samples `lane` and 64 + `lane` of the hop (frame pairs (j-1, j) and (j, j+1))
    const int n_a = jb * kTile + lane, n_b = n_a + 64;
    const NwsLerp la = nws_lerp_coeff(n_a, T), lb = nws_lerp_coeff(n_b, T);
    const int qa = la.i0 - (jb - 1), qb = lb.i0 - (jb - 1);
    float pa = 0.0f, pb = 0.0f;
    __syncthreads();   // every wave is through its K loop: whi / wlo are dead, the planes may be written
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      // FiLM'ed inputs of M-tile m (shapers 32 m .. 32 m + 31): accumulator registers 4g..4g+3 are shapers 32m + 8g + 4half + 0..3
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int s4 = 32 * m + 8 * g + 4 * half;
        const float4 fa0 = *reinterpret_cast<const float4*>(&L.fa[q0][0][s4]), fd0 = *reinterpret_cast<const float4*>(&L.fd[q0][0][s4]);
        const float4 fa1 = *reinterpret_cast<const float4*>(&L.fa[q0][1][s4]), fd1 = *reinterpret_cast<const float4*>(&L.fd[q0][1][s4]);
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const int r0 = 4 * g + 2 * h2;
          const f32x2 x2 = m == 0 ? f32x2{acc0[r0], acc0[r0 + 1]} : f32x2{acc1[r0], acc1[r0 + 1]};   // harmonic_mixer output, bias included
          const f32x2 g_i = fma2(w1_b, h2 == 0 ? f32x2{fd0.x, fd0.y} : f32x2{fd0.z, fd0.w}, h2 == 0 ? f32x2{fa0.x, fa0.y} : f32x2{fa0.z, fa0.w});
          const f32x2 b_i = fma2(w1_b, h2 == 0 ? f32x2{fd1.x, fd1.y} : f32x2{fd1.z, fd1.w}, h2 == 0 ? f32x2{fa1.x, fa1.y} : f32x2{fa1.z, fa1.w});
          const f32x2 xi = fma2(g_i, x2, b_i);  // FiLM (models/modules/dynamic.py:8)
          BK.xi[8 * g + 4 * half + 2 * h2][32 * w4 + col] = xi.x;
          BK.xi[8 * g + 4 * half + 2 * h2 + 1][32 * w4 + col] = xi.y;
        }
      }
      __syncthreads();
      // wave v: shapers 32 m + 8 v .. + 7 of this pass
      const int lv = __builtin_amdgcn_readfirstlane(8 * wave);
#pragma unroll 1
      for (int k = 0; k < 8; ++k) {
        const int sh_idx = 32 * m + lv + k;
        const float* __restrict__ W = bank + (size_t)sh_idx * kBankRow;   // a __restrict__ kernel argument: scalar loads
        const float ya = bank_shaper<MODE == kModeExactBank>(W, BK.xi[lv + k][lane]);
        const float yb = bank_shaper<MODE == kModeExactBank>(W, BK.xi[lv + k][64 + lane]);
        pa = fmaf(fmaf(la.w1, L.fd[qa][2][sh_idx], L.fa[qa][2][sh_idx]), ya, pa);   // normalising FiLM gain x newt.mixer weight
        pb = fmaf(fmaf(lb.w1, L.fd[qb][2][sh_idx], L.fa[qb][2][sh_idx]), yb, pb);
      }
      if (m == 0) __syncthreads();   // the planes of pass 0 have been read by everybody
    }
    BK.red[wave][lane] = pa;
    BK.red[wave][64 + lane] = pb;
    __syncthreads();
    if (tid < kTile) {
      const int n_o = jb * kTile + tid;
      const NwsLerp lo = nws_lerp_coeff(n_o, T);
      const int qo = lo.i0 - (jb - 1);
      const float sum = (BK.red[0][tid] + BK.red[1][tid]) + (BK.red[2][tid] + BK.red[3][tid]);
      const float bias_o = fmaf(lo.w1, L.bsum[qo + 1] - L.bsum[qo], L.bsum[qo]);
      const float tot_o = sum + (bias_o + w.newt_out_b[0]);
      newt_out[(size_t)b * N + n_o] = add_in != nullptr ? add_in[(size_t)b * N + n_o] + tot_o : tot_o;
    }
    return;
  }
  const f32x2 w1_2 = splat2(lc.w1);
  f32x2 part2 = {0.0f, 0.0f};
  // accumulator registers 4g..4g+3 of M-tile m are the 4 consecutive shapers 32m + 8g + 4half + 0..3
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int sb = 32 * m + 8 * g;  // compile-time part of the shaper index
      const int s4 = sb + 4 * half;
      float4 fa[3], fd[3];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) {
        fa[ty] = *reinterpret_cast<const float4*>(&L.fa[q0][ty][s4]);
        fd[ty] = *reinterpret_cast<const float4*>(&L.fd[q0][ty][s4]);
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {  // shaper pairs (s4, s4+1) and (s4+2, s4+3)
        const int r0 = 4 * g + 2 * h2;
        const f32x2 accp = m == 0 ? f32x2{acc0[r0], acc0[r0 + 1]} : f32x2{acc1[r0], acc1[r0 + 1]};
#define NWS_PAIR(v) (h2 == 0 ? f32x2{(v).x, (v).y} : f32x2{(v).z, (v).w})
        const f32x2 x2 = accp;  // harmonic_mixer output, bias included (K slot 0)
        const f32x2 g_i = fma2(w1_2, NWS_PAIR(fd[0]), NWS_PAIR(fa[0]));  // F.upsample of the FiLM parameters
        const f32x2 b_i = fma2(w1_2, NWS_PAIR(fd[1]), NWS_PAIR(fa[1]));
        const f32x2 g_n = fma2(w1_2, NWS_PAIR(fd[2]), NWS_PAIR(fa[2]));  // already times newt.mixer.weight
        const f32x2 xi = fma2(g_i, x2, b_i);  // FiLM (models/modules/dynamic.py:8)
        f32x2 sh;
        if (DBG == 2) {
          sh = xi;
        } else if (MODE == kModeLutPairsDiv6) {
          sh = lut_shaper2_fast(LF, LF.pairs + (size_t)(sb + 2 * h2) * LF.row_bytes, lane_off_bytes, xi);
        } else if (is_lut(MODE)) {
          sh = lut_shaper2<MODE != kModeLut, MODE == kModeLutPairsDiv6>(LP, (sb + 2 * h2) * LP.size + lane_row_off, xi);
        } else {
          sh = f32x2{exact_shaper(SH, s4 + 2 * h2, xi.x), exact_shaper(SH, s4 + 2 * h2 + 1, xi.y)};
        }
        part2 = fma2(g_n, sh, part2);  // normalising FiLM gain and 64->1 mix in one FMA
#undef NWS_PAIR
      }
      // fence the scheduler per group of 4 shapers: bounded number of gathers / FiLM operands live at once
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const float partial = part2.x + part2.y;
  // the normalising FiLM biases went through the mixer per FRAME: interpolate their sum like any other parameter
  const float bias_n = fmaf(lc.w1, L.bsum[q0 + 1] - L.bsum[q0], L.bsum[q0]);
  const float total = partial + nws_swap_halves(partial) + (bias_n + w.newt_out_b[0]);
  if (half == 0) newt_out[(size_t)b * N + n] = add_in != nullptr ? add_in[(size_t)b * N + n] + total : total;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone HarmonicOscillator.forward (models/modules/generators.py:58-66): (B, N) upsampled F0 -> (B, 101, N) sines
// times the anti-alias mask.  Same phase arithmetic as the fused kernel (fp64 prefix sum from the 32-sample carries, the
// reference's fp32 rounding chain, arguments fl(fl(k phase) + shift)); one thread per sample, 101 coalesced stores.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void oscillator_kernel(const float* __restrict__ f0_up, const double* __restrict__ carry,
                                                         const float* __restrict__ phase_u, const float* __restrict__ rand_phase,
                                                         int N, float sample_rate, float* __restrict__ out) {
  __shared__ float shift[kK];
  const int b = blockIdx.y;
  const int n = blockIdx.x * 128 + threadIdx.x;          // N is a multiple of 128: no partial blocks
  if (threadIdx.x < kK) shift[threadIdx.x] = phase_u[threadIdx.x] * rand_phase[threadIdx.x] - kPi;   // generators.py:54-56
  const float f0n = f0_up[(size_t)b * N + n];
  double cs = scan32_f64((double)f0n);
  cs += carry[(size_t)b * (N / 32) + (n >> 5)];
  const float tc = kTau * (float)cs;
  const float phase = __fdiv_rn(tc, sample_rate);
  const float nyquist = sample_rate * 0.5f;
  __syncthreads();
  float* o = out + (size_t)b * kK * N + n;
  for (int k = 1; k <= kK; ++k) {
    const float kf = (float)k;
    const float arg = kf * phase + shift[k - 1];           // two roundings (-ffp-contract=off)
    const float v = nws_sinf(arg);
    o[(size_t)(k - 1) * N] = (f0n * kf) < nyquist ? v : 0.0f;   // NaN F0: mask false, like the reference's comparison
  }
}

// ---------------------------------------------------------------------------------------------
// Stand-alone NEWT.forward / FastNEWT.forward on a materialised exciter (models/modules/shaping.py:67-79):
// film (B, 256, T) channel-major [g_idx | b_idx | g_norm | b_norm] as newt.mlp returns it, upsampled x128 on the fly;
// x = g_idx e + b_idx -> shaper -> g_norm s + b_norm -> Conv1d(64 -> 1).  One thread per sample; the reference's own
// rounding chains (FiLM as multiply then add, LUT index chain of FastNEWT._lookup), since nothing is fused here.
// ---------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(128) void newt_apply_kernel(NwsWeights w, const float* __restrict__ exciter,
                                                         const float* __restrict__ film, int T, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw);
  if (MODE == kModeExact) load_shaper_lds<false>(SH, w, threadIdx.x, 128);
  __syncthreads();
  const int b = blockIdx.y;
  const int N = T * NWS_HOP;
  const int n = blockIdx.x * 128 + threadIdx.x;          // one hop per workgroup
  const NwsLerp lc = nws_lerp_coeff(n, T);
  LutParams LP;
  if (MODE == kModeLut) LP = make_lut_params(w);
  const float* fb = film + (size_t)b * NWS_FILM_CH * T;
  const float* eb = exciter + (size_t)b * kS * N + n;
  float acc = w.newt_out_b[0];
  for (int s = 0; s < kS; ++s) {
    const float g_i = nws_lerp(fb[(size_t)s * T + lc.i0], fb[(size_t)s * T + lc.i1], lc.w0, lc.w1);
    const float b_i = nws_lerp(fb[(size_t)(kS + s) * T + lc.i0], fb[(size_t)(kS + s) * T + lc.i1], lc.w0, lc.w1);
    const float g_n = nws_lerp(fb[(size_t)(2 * kS + s) * T + lc.i0], fb[(size_t)(2 * kS + s) * T + lc.i1], lc.w0, lc.w1);
    const float b_n = nws_lerp(fb[(size_t)(3 * kS + s) * T + lc.i0], fb[(size_t)(3 * kS + s) * T + lc.i1], lc.w0, lc.w1);
    const float x = g_i * eb[(size_t)s * N] + b_i;                                     // FiLM: gamma * x + beta (dynamic.py:8)
    const float sh = MODE == kModeLut ? lut_shaper<false, false>(LP, s * LP.size, x) : exact_shaper_precise(SH, s, x);
    acc = fmaf(w.newt_out_w[s], g_n * sh + b_n, acc);
  }
  out[(size_t)b * N + n] = acc;
}

// element-wise shaper application on (B,64,N) (stage tests / FastNEWT table construction)
template <int MODE>
__global__ __launch_bounds__(256) void shaper_apply_kernel(NwsWeights w, const float* __restrict__ x, int64_t N,
                                                           float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw);
  if (MODE == kModeExact) load_shaper_lds<false>(SH, w, threadIdx.x, 256);
  __syncthreads();
  const int64_t row = blockIdx.y;  // b*64 + s
  const int s = (int)(row & 63);
  LutParams LP;
  if (MODE == kModeLut) LP = make_lut_params(w);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const float v = x[row * N + i];
    y[row * N + i] = MODE == kModeLut ? lut_shaper<false, false>(LP, s * LP.size, v) : exact_shaper_precise(SH, s, v);
  }
}

__global__ __launch_bounds__(256) void shaper_table_kernel(NwsWeights w, int size, float tmin, float tmax,
                                                           float* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw);
  load_shaper_lds<false>(SH, w, threadIdx.x, 256);
  __syncthreads();
  const int s = blockIdx.y;
  // torch.linspace(min, max, size) in fp32 (ATen RangeFactories, symmetric form; bit-exact with the
  // CPU kernel as probed in the build container): step = (max-min)/(size-1); first half
  // fma(step, i, min), second half fma(-step, size-1-i, max).
  const float step = __fdiv_rn(tmax - tmin, (float)(size - 1));
  const int halfway = size / 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < size; i += gridDim.x * 256) {
    const float xv = i < halfway ? fmaf(step, (float)i, tmin) : fmaf(-step, (float)(size - 1 - i), tmax);
    table[(size_t)s * size + i] = exact_shaper_precise(SH, s, xv);
  }
}

// Candidate cheaper sines, measured on hardware through nws_debug_sin before any of them may replace nws_sinf.
//  exact-product reduction to turns: p = x*C_hi, e = fma(x, C_hi, -p) (the product's rounding error),
//  t = (p - rint(p)) + (e + x*C_lo)  in [-0.53, 0.53] turns, good to ~3e-8 turns for |x| <= 6e6.
__device__ __forceinline__ float turns_reduce(float x) {
  const float c_hi = 0.15915493667125702f;     // fl32(1/(2 pi))
  const float c_lo = 6.4206382432985265e-09f;   // 1/(2 pi) - c_hi
  const float p = x * c_hi;
  const float e = fmaf(x, c_hi, -p);
  return (p - rintf(p)) + fmaf(x, c_lo, e);
}
__device__ __forceinline__ float sin_hw_turns(float x) { return __builtin_amdgcn_sinf(turns_reduce(x)); }
__device__ __forceinline__ float sin_poly_turns(float x) {
  float t = turns_reduce(x);
  t = t - rintf(t);                                   // [-0.5, 0.5]
  const float f = copysignf(0.5f, t) - t;             // sin(pi - a) = sin(a)
  t = fabsf(t) > 0.25f ? f : t;                       // [-0.25, 0.25] turns
  const float a = t * 6.283185307179586f;             // [-pi/2, pi/2]
  const float z = a * a;
  float q = fmaf(-2.4080531346726275e-08f, z, 2.753648004727438e-06f);  // odd degree-11 fit, |err| < 2e-8 on [-pi/2, pi/2]
  q = fmaf(q, z, -0.00019841086759697646f);
  q = fmaf(q, z, 0.00833333283662796f);
  q = fmaf(q, z, -0.1666666716337204f);
  return fmaf(a * z, q, a);
}

template <int MODE>
__global__ void sin_variant_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int reps) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    float acc = 0.0f;
    for (int r = 0; r < reps; ++r) {
      const float a = v + (float)r * 1.0e-3f * acc;
      acc += MODE == 0 ? nws_sinf(a) : MODE == 1 ? sin_hw_turns(a) : sin_poly_turns(a);
    }
    y[i] = acc;
  }
}

// mixer_w (64,101) fp32 -> [whi fragments | wlo fragments] exactly as ExcLds holds them
__global__ void mixer_frags_kernel(const float* __restrict__ mixer_w, const float* __restrict__ mixer_b,
                                   _Float16* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kS * kKPad) return;
  const int s = e / kKPad, kk = e - s * kKPad;
  const float wv = mixer_slot_weight(mixer_w, mixer_b, s, kk);
  const int at = mixer_frag_index(s, kk);
  _Float16 h, l;
  split_f16(wv, h, l);
  out[at] = h;
  out[kKSteps * 2 * 2 * 32 * 8 + at] = l;
}

// X[s] = (sum_k |W[s][k]| + |b[s]|) rounded up: one wave per shaper
__global__ void exciter_bound_kernel(const float* __restrict__ mixer_w, const float* __restrict__ mixer_b, float* __restrict__ out) {
  const int s = blockIdx.x, lane = threadIdx.x;
  float v = 0.0f;
  for (int k = lane; k < kK; k += 64) v += fabsf(mixer_w[(size_t)s * kK + k]);
  v = wave_sum_to_lane63(v);
  if (lane == 63) out[s] = (v + fabsf(mixer_b[s])) * (1.0f + 1.0f / 1024.0f);
}

__global__ void lut_pairs_kernel(const float* __restrict__ table, int size, float2* __restrict__ pairs) {
  const int s = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < size; i += gridDim.x * blockDim.x) {
    const float lv = table[(size_t)s * size + i];
    const float uv = table[(size_t)s * size + (i + 1 < size ? i + 1 : size - 1)];
    pairs[(size_t)s * size + i] = make_float2(lv, uv - lv);
  }
}

__global__ void sin_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = nws_sinf(x[i]);
}

// A = I-like / asymmetric-B check of the 32x32x2 f32 MFMA fragment maps used above:
//   A[i][k]: lane l holds A[l&31][l>>5];  B[k][j]: lane l holds B[l>>5][l&31];
//   D[i][j]: lane l, reg r holds D[(r&3)+8(r>>2)+4(l>>5)][l&31].
__global__ void selftest_mfma_kernel(int32_t* bad) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, col = lane & 31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  // D = sum over 16 steps of A_s (32x2) * B_s (2x32), with A[i][k] = (i == k) over K = 32, B[k][j] = 100k + j
  for (int s = 0; s < 16; ++s) {
    const int k = 2 * s + half;
    const float a = (col == k) ? 1.0f : 0.0f;
    const float bv = 100.0f * (float)k + (float)col;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc, 0, 0, 0);
  }
  int nbad = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
    if (acc[r] != 100.0f * (float)i + (float)col) ++nbad;
  }
  // same check for the 32x32x16 f16 form with the fragment convention used above: lane (i|j = l&31, h = l>>5),
  // element e <-> k = 8h + e for BOTH operands.  A[i][k] = (i % 16 == k), B[k][j] = 100k + j  ->  D[i][j] = B[i%16][j]
  f16x8 a16, b16;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * half + e;
    a16[e] = (_Float16)((col % 16) == k ? 1.0f : 0.0f);
    b16[e] = (_Float16)(100.0f * (float)k + (float)col);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a16, b16, acc, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
    if (acc[r] != 100.0f * (float)(i % 16) + (float)col) ++nbad;
  }
  if (nbad) atomicAdd(bad, nbad);
}

bool weights_ok(const NwsWeights* w) { return w != nullptr && w->mixer_w && w->mixer_b; }

}  // namespace

extern "C" {

int nws_abi_version(void) { return NWS_ABI_VERSION; }

const char* nws_error_string(int code) {
  if (code == NWS_OK) return "ok";
  if (code == NWS_ERR_UNSUPPORTED) return "nws: unsupported size (kernels are specialised for gin/models/newt.gin)";
  if (code == NWS_ERR_BAD_ARG) return "nws: bad argument";
  if (code == NWS_ERR_WORKSPACE) return "nws: workspace too small";
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "nws: unknown error";
}

int nws_selftest_mfma(int32_t* bad_out, void* stream) {
  if (!bad_out) return NWS_ERR_BAD_ARG;
  hipError_t e = hipMemsetAsync(bad_out, 0, sizeof(int32_t), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  selftest_mfma_kernel<<<1, 64, 0, (hipStream_t)stream>>>(bad_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_sin(const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y || n < 0) return NWS_ERR_BAD_ARG;
  if (n == 0) return NWS_OK;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  sin_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, y, n);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_mixer_frags(const float* mixer_w, const float* mixer_b, void* frags_out, void* stream) {
  if (!mixer_w || !mixer_b || !frags_out) return NWS_ERR_BAD_ARG;
  mixer_frags_kernel<<<(kS * kKPad + 255) / 256, 256, 0, (hipStream_t)stream>>>(mixer_w, mixer_b,
                                                                                 static_cast<_Float16*>(frags_out));
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_debug_film_frags(const NwsWeights* w, const float* film, int B, int T, void* frags_out, void* stream) {
  if (!w || !film || !frags_out || B <= 0 || T <= 0 || !w->newt_out_w || w->lut_size < 2) return NWS_ERR_BAD_ARG;
  const size_t frames = (size_t)B * T;
  if (frames >= (1ull << 31)) return NWS_ERR_UNSUPPORTED;
  unsigned char* recs = static_cast<unsigned char*>(frags_out);
  film_frags_kernel<<<(unsigned)frames, 64, 0, (hipStream_t)stream>>>(*w, film, recs, reinterpret_cast<FilmAux*>(recs + frames * NWS_FILM_REC_BYTES));
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_exciter_bound(const float* mixer_w, const float* mixer_b, float* bound_out, void* stream) {
  if (!mixer_w || !mixer_b || !bound_out) return NWS_ERR_BAD_ARG;
  exciter_bound_kernel<<<kS, 64, 0, (hipStream_t)stream>>>(mixer_w, mixer_b, bound_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_shaper_turns(const NwsWeights* w, float* table_out, void* stream) {
  if (!w || !table_out || !w->shaper_in_scale || !w->shaper_w0 || !w->shaper_b0 || !w->shaper_w2 || !w->shaper_b2 ||
      !w->shaper_w4 || !w->shaper_b4 || !w->shaper_w6 || !w->shaper_b6)
    return NWS_ERR_BAD_ARG;
  shaper_turns_kernel<<<NWS_N_SHAPERS, 192, 0, (hipStream_t)stream>>>(*w, table_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_lut_pairs(const float* table, int table_size, float* pairs_out, void* stream) {
  if (!table || !pairs_out || table_size < 2) return NWS_ERR_BAD_ARG;
  const dim3 grid((table_size + 255) / 256, NWS_N_SHAPERS);
  lut_pairs_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(table, table_size, reinterpret_cast<float2*>(pairs_out));
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_debug_sin(int mode, const float* x, float* y, int64_t n, int reps, void* stream) {
  if (!x || !y || n <= 0 || reps < 1) return NWS_ERR_BAD_ARG;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) sin_variant_kernel<0><<<blocks, 256, 0, st>>>(x, y, n, reps);
  else if (mode == 1) sin_variant_kernel<1><<<blocks, 256, 0, st>>>(x, y, n, reps);
  else if (mode == 2) sin_variant_kernel<2><<<blocks, 256, 0, st>>>(x, y, n, reps);
  else return NWS_ERR_BAD_ARG;
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_phase_carry(const float* f0, const float* f0_up, int B, int T, double* carry, void* stream) {
  if ((!f0 && !f0_up) || !carry || B <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  phase_carry_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(f0, f0_up, T, carry);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_exciter_newt(const NwsWeights* w, const float* f0, const float* f0_up, const double* carry,
                     const float* phase_u, const float* rand_phase, const float* film, int B, int T,
                     float sample_rate, float* exciter_out, float* newt_out, void* stream) {
  return nws_exciter_newt_add(w, f0, f0_up, carry, phase_u, rand_phase, film, nullptr, B, T, sample_rate, exciter_out, newt_out, stream);
}

int nws_exciter_newt_add(const NwsWeights* w, const float* f0, const float* f0_up, const double* carry,
                         const float* phase_u, const float* rand_phase, const float* film, const float* add_in, int B, int T,
                         float sample_rate, float* exciter_out, float* newt_out, void* stream) {
  if (!weights_ok(w) || (!f0 && !f0_up) || !carry || !phase_u || !rand_phase || B <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  if (!exciter_out && !newt_out) return NWS_ERR_BAD_ARG;
  if (add_in && !newt_out) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  const dim3 grid(T, B);
  const size_t base = (sizeof(ExcLds) + 15) & ~size_t(15);
  hipStream_t st = (hipStream_t)stream;
  if (!newt_out) {
    exciter_newt_kernel<kModeExciterOnly><<<grid, 256, base, st>>>(*w, f0, f0_up, carry, phase_u, rand_phase, film, T,
                                                                   sample_rate, exciter_out, newt_out);
  } else {
    if (!film || !w->newt_out_w || !w->newt_out_b) return NWS_ERR_BAD_ARG;
    if (w->lut != nullptr) {
      if (w->lut_size < 2 || !(w->lut_max > w->lut_min)) return NWS_ERR_BAD_ARG;
      const bool pow2 = (w->lut_size & (w->lut_size - 1)) == 0 && w->lut_size <= (1 << 20);
      if (w->lut_pairs != nullptr && w->lut_max - w->lut_min == 6.0f && pow2) {
        // FastNEWT hot path.  exciter_opts (nws_hip.h): round-1 FiLM interpolation on the VALU / one fp16 term per sine in
        // every K-step / in K-steps 1..6 only
        // XCD-contiguous hop groups (see the kernel); NWS_EXCITER_XCD=0 restores grid (groups, B) (measurements)
        static const bool xcd_map = [] { const char* e = getenv("NWS_EXCITER_XCD"); return !(e && e[0] == '0'); }();
        const int groups = (T + 1) / 2;
        const int xcd_groups = xcd_map && (long long)groups * B >= 64 && (long long)groups * B < (1ll << 31) ? groups : 0;
        const dim3 g2 = xcd_groups ? dim3((unsigned)(groups * B), 1) : dim3(groups, B);
        const int opts = w->exciter_opts;
        // the 80-register tail (kOptLowReg: three workgroups per CU) is the default; NWS_EXCITER_LOWREG=0 keeps the 93-register
        // one (measurements: same results, 3 % slower alone, 6 % slower on realistic F0)
        static const bool low_reg = [] { const char* e = getenv("NWS_EXCITER_LOWREG"); return !(e && e[0] == '0'); }();
        // (measurements: NWS_EXCITER_LDS_PAD=<bytes> of unused LDS per workgroup lowers the occupancy of the hot kernel)
        static const size_t hot_pad = [] { const char* e = getenv("NWS_EXCITER_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();
#define NWS_HOT(O) exciter_newt_kernel<kModeLutPairsDiv6, 0, 2, O><<<g2, 512, base + hot_pad, st>>>( \
            *w, f0, f0_up, carry, phase_u, rand_phase, film, T, sample_rate, exciter_out, newt_out, nullptr, add_in, xcd_groups)
        // (measurements, VERDICT r4 #4: NWS_EXCITER_HPB=1 launches the default kernel as 4-wave workgroups of one hop - three of them
        // fit beside a 251-register recurrence wave on a SIMD where one 8-wave workgroup = two waves per SIMD does; alone it is
        // LDS-bound at four workgroups = four waves per SIMD instead of six.  Same bits.)
        static const bool one_hop = [] { const char* e = getenv("NWS_EXCITER_HPB"); return e && e[0] == '1'; }();
        if (one_hop && opts == 0) {
          const int xg1 = xcd_map && (long long)T * B >= 64 && (long long)T * B < (1ll << 31) ? T : 0;
          const dim3 g1 = xg1 ? dim3((unsigned)(T * B), 1) : dim3(T, B);
          exciter_newt_kernel<kModeLutPairsDiv6, 0, 1, kOptFilmMfma | kOptLowReg><<<g1, 256, base + hot_pad, st>>>(
              *w, f0, f0_up, carry, phase_u, rand_phase, film, T, sample_rate, exciter_out, newt_out, nullptr, add_in, xg1);
        } else
        if (opts & NWS_EXCITER_VALU_FILM) NWS_HOT(0);
        else if (opts & NWS_EXCITER_ONE_TERM) NWS_HOT(kOptFilmMfma | kOptOneTerm | kOptLowReg);
        else if (opts & NWS_EXCITER_HYBRID_W) { if (low_reg) NWS_HOT(kOptFilmMfma | kOptHybrid | kOptHybridW | kOptLowReg); else NWS_HOT(kOptFilmMfma | kOptHybrid | kOptHybridW); }
        else if (opts & NWS_EXCITER_HYBRID) NWS_HOT(kOptFilmMfma | kOptHybrid | kOptLowReg);
        else if (low_reg) NWS_HOT(kOptFilmMfma | kOptLowReg);
        else NWS_HOT(kOptFilmMfma);
#undef NWS_HOT
      } else if (w->lut_pairs != nullptr)
        exciter_newt_kernel<kModeLutPairs><<<grid, 256, base, st>>>(*w, f0, f0_up, carry, phase_u, rand_phase, film,
                                                                    T, sample_rate, exciter_out, newt_out, nullptr, add_in);
      else
        exciter_newt_kernel<kModeLut><<<grid, 256, base, st>>>(*w, f0, f0_up, carry, phase_u, rand_phase, film, T,
                                                               sample_rate, exciter_out, newt_out, nullptr, add_in);
    } else {
      if (!w->shaper_w0 || !w->shaper_w2 || !w->shaper_w4 || !w->shaper_w6) return NWS_ERR_BAD_ARG;
      // (measurements: NWS_EXCITER_BANK_LDS_PAD=<bytes> of unused LDS per workgroup lowers the bank kernel's occupancy -
      // 35840 restores the two workgroups per CU it had while the shaper-input planes lived behind ExcLds)
      static const size_t bank_pad = [] { const char* e = getenv("NWS_EXCITER_BANK_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();
      if (w->shaper_turns != nullptr && (w->exciter_opts & NWS_EXCITER_BANK_NOFRACT))
        exciter_newt_kernel<kModeExactBankNF><<<grid, 256, base + bank_pad, st>>>(
            *w, f0, f0_up, carry, phase_u, rand_phase, film, T, sample_rate, exciter_out, newt_out, w->shaper_turns, add_in);
      else if (w->shaper_turns != nullptr)
        exciter_newt_kernel<kModeExactBank><<<grid, 256, base + bank_pad, st>>>(
            *w, f0, f0_up, carry, phase_u, rand_phase, film, T, sample_rate, exciter_out, newt_out, w->shaper_turns, add_in);
      else
        exciter_newt_kernel<kModeExact><<<grid, 256, base + sizeof(ShaperLds), st>>>(
            *w, f0, f0_up, carry, phase_u, rand_phase, film, T, sample_rate, exciter_out, newt_out, nullptr, add_in);
    }
  }
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_debug_exciter_newt(int variant, const NwsWeights* w, const float* f0, const double* carry, const float* phase_u,
                           const float* rand_phase, const float* film, int B, int T, float sample_rate,
                           float* newt_out, void* stream) {
  if (!weights_ok(w) || !f0 || !carry || !phase_u || !rand_phase || !film || !newt_out || !w->lut || !w->lut_pairs)
    return NWS_ERR_BAD_ARG;
  if (w->lut_max - w->lut_min != 6.0f || (w->lut_size & (w->lut_size - 1)) != 0) return NWS_ERR_UNSUPPORTED;
  const dim3 grid(T, B);
  const size_t base = (sizeof(ExcLds) + 15) & ~size_t(15);
  hipStream_t st = (hipStream_t)stream;
#define NWS_OPT_LAUNCH(O)                                                                                          \
  exciter_newt_kernel<kModeLutPairsDiv6, 0, 2, O><<<dim3((T + 1) / 2, B), 512, base, st>>>(*w, f0, nullptr, carry, phase_u, \
                                                            rand_phase, film, T, sample_rate, nullptr, newt_out)
  if (variant == 5) {   // prologue only, product configuration
    exciter_newt_kernel<kModeLutPairsDiv6, 5, 2, kOptFilmMfma | kOptLowReg><<<dim3((T + 1) / 2, B), 512, base, st>>>(
        *w, f0, nullptr, carry, phase_u, rand_phase, film, T, sample_rate, nullptr, newt_out);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  if ((variant >= 21 && variant <= 24) || (variant >= 26 && variant <= 29)) {
    // timing ablations of the PRODUCT configuration (results wrong by design): 21 no sines, 22 no table gathers, 23 no tail, 24 no mixer MFMAs;
    // 26 no global load in front of the barrier, 27 / 28 / 29 no FiLM rows / no fragment DMA / no F0, carry, shifts (tools/exciter_ablate_product.py)
#define NWS_ABL(D) exciter_newt_kernel<kModeLutPairsDiv6, D, 2, kOptFilmMfma | kOptLowReg><<<dim3((T + 1) / 2, B), 512, base, st>>>( \
        *w, f0, nullptr, carry, phase_u, rand_phase, film, T, sample_rate, nullptr, newt_out)
    switch (variant) {
      case 21: NWS_ABL(1); break;
      case 22: NWS_ABL(2); break;
      case 23: NWS_ABL(3); break;
      case 24: NWS_ABL(4); break;
      case 26: NWS_ABL(6); break;
      case 27: NWS_ABL(7); break;
      case 28: NWS_ABL(8); break;
      default: NWS_ABL(9); break;
    }
#undef NWS_ABL
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  if (variant == 6) {   // prologue only, fragment records by LDS-DMA (`film` = the records + aux entries of nws_debug_film_frags here)
    exciter_newt_kernel<kModeLutPairsDiv6, 5, 2, kOptFilmMfma | kOptLowReg | kOptFilmDma><<<dim3((T + 1) / 2, B), 512, base, st>>>(
        *w, f0, nullptr, carry, phase_u, rand_phase, film, T, sample_rate, nullptr, newt_out, film, nullptr, 0, B * T);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  if (variant >= 10) {   // 10 + OPT bits: the product kernel's compile-time options (two hops per workgroup)
    switch (variant - 10) {
      case 0: NWS_OPT_LAUNCH(0); break;
      case 2: NWS_OPT_LAUNCH(2); break;
      case 26: NWS_OPT_LAUNCH(26); break;
      case 34: NWS_OPT_LAUNCH(34); break;   // kOptFilmMfma | kOptLowReg: the default kernel
      case 58: NWS_OPT_LAUNCH(58); break;   // ... | kOptHybrid | kOptHybridW | kOptLowReg: the opt-in hybrid-W kernel
      case 98:                              // kOptFilmMfma | kOptLowReg | kOptFilmDma (`film` = records + aux entries of nws_debug_film_frags)
        exciter_newt_kernel<kModeLutPairsDiv6, 0, 2, 98><<<dim3((T + 1) / 2, B), 512, base, st>>>(
            *w, f0, nullptr, carry, phase_u, rand_phase, film, T, sample_rate, nullptr, newt_out, film, nullptr, 0, B * T);
        break;
      default: return NWS_ERR_BAD_ARG;
    }
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
#undef NWS_OPT_LAUNCH
#define NWS_DBG_LAUNCH(V)                                                                                         \
  exciter_newt_kernel<kModeLutPairsDiv6, V><<<grid, 256, base, st>>>(*w, f0, nullptr, carry, phase_u, rand_phase, film, T, \
                                                            sample_rate, nullptr, newt_out)
  switch (variant) {
    case 0: NWS_DBG_LAUNCH(0); break;
    case 1: NWS_DBG_LAUNCH(1); break;
    case 2: NWS_DBG_LAUNCH(2); break;
    case 3: NWS_DBG_LAUNCH(3); break;
    case 4: NWS_DBG_LAUNCH(4); break;
    default: return NWS_ERR_BAD_ARG;
  }
#undef NWS_DBG_LAUNCH
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_oscillator(const float* f0_up, const double* carry, const float* phase_u, const float* rand_phase, int B, int N,
                   float sample_rate, float* out, void* stream) {
  if (!f0_up || !carry || !phase_u || !rand_phase || !out || B <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (N % NWS_HOP != 0 || B > 65535) return NWS_ERR_UNSUPPORTED;
  oscillator_kernel<<<dim3(N / NWS_HOP, B), 128, 0, (hipStream_t)stream>>>(
      f0_up, carry, phase_u, rand_phase, N, sample_rate, out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_newt_apply(const NwsWeights* w, const float* exciter, const float* film, int B, int T, float* out, void* stream) {
  if (!w || !exciter || !film || !out || B <= 0 || T <= 0 || !w->newt_out_w || !w->newt_out_b) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  const dim3 grid(T, B);
  const int threads = 128;
  if (w->lut != nullptr) {
    if (w->lut_size < 2 || !(w->lut_max > w->lut_min)) return NWS_ERR_BAD_ARG;
    newt_apply_kernel<kModeLut><<<grid, threads, 16, (hipStream_t)stream>>>(*w, exciter, film, T, out);
  } else {
    if (!w->shaper_in_scale || !w->shaper_w0 || !w->shaper_w2 || !w->shaper_w4 || !w->shaper_w6) return NWS_ERR_BAD_ARG;
    newt_apply_kernel<kModeExact><<<grid, threads, sizeof(ShaperLds), (hipStream_t)stream>>>(*w, exciter, film, T, out);
  }
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_shaper_table(const NwsWeights* w, int table_size, float table_min, float table_max, float* table_out,
                     void* stream) {
  if (!w || !table_out || table_size < 2 || !w->shaper_w0) return NWS_ERR_BAD_ARG;
  const dim3 grid((table_size + 255) / 256, NWS_N_SHAPERS);
  shaper_table_kernel<<<grid, 256, sizeof(ShaperLds), (hipStream_t)stream>>>(*w, table_size, table_min, table_max,
                                                                             table_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_shaper_apply(const NwsWeights* w, const float* x, int64_t B, int64_t N, float* y, void* stream) {
  if (!w || !x || !y || B <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (B * NWS_N_SHAPERS > 65535) return NWS_ERR_UNSUPPORTED;
  const int gx = (int)((N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024);
  const dim3 grid(gx, (unsigned)(B * NWS_N_SHAPERS));
  if (w->lut != nullptr) {
    shaper_apply_kernel<kModeLut><<<grid, 256, 16, (hipStream_t)stream>>>(*w, x, N, y);
  } else {
    if (!w->shaper_w0) return NWS_ERR_BAD_ARG;
    shaper_apply_kernel<kModeExact><<<grid, 256, sizeof(ShaperLds), (hipStream_t)stream>>>(*w, x, N, y);
  }
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
