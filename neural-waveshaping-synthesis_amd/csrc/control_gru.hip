// Control encoder GRU(2 -> 128), persistent over the T control frames.
//
// Replaces aten::gru as called by ControlModule.forward (models/neural_waveshaping.py:24-25) on
// control[:, 0:2] (get_embedding, :69-72): gate order [r; z; n], h0 = 0,
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr),  z likewise,
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (h - n) * z + n.
//
// Design: the recurrence is strictly sequential, so one workgroup owns one utterance for all T steps and
// W_hh (384x128 fp32 = 192 KB, more than the 160 KB LDS) lives in VGPRs.  Wave w owns hidden units [32w, 32w+32); lane
// l = 32a + 16s + p holds the three gate rows of the unit PAIR (32w + 2p, +1) restricted to the K-quarter 2a + s (columns
// [32(2a+s), +32)): 192 weights per lane.  Per step a lane reads its 32 h values (8 broadcast ds_read_b128), does 96
// packed FMAs (v_pk_fma_f32; each h value serves six outputs), reduces over the four K-quarters with one
// v_permlane32_swap per gate - which also separates the two units: lanes a=0 keep the first, a=1 the second - and one
// self v_permlane16_swap, then evaluates the gates of ITS unit (two replicas) with v_exp/v_rcp (1 ulp) and writes h'
// into the other half of a ping-pong LDS buffer: ONE workgroup barrier per step that waits on LDS only (raw s_barrier +
// lgkmcnt(0); the 512 B h_t global store is never drained on the critical path), control values staged in LDS 1024
// frames at a time.
// Measured (MI355X, 500 steps, tools/gru_variants.py): this kernel 0.223 ms (0.446 us/step; 0.458 before round 3 pipelined
// the reads of h by hand).  Cycle timeline of a step (variant 6, s_memtime probes, ~1070 cycles): LDS reads + 96 packed FMAs
// ~640, cross-lane reduction ~125, gates ~125, h store + barrier ~110.  The FMA phase sits on what the vector pipe delivers:
// tools/ubench/valu_rate.hip measures 6.4 cycles per v_pk_fma_f32 for a lone wave and 5.3-5.7 per SIMD with 2-8 waves, and the
// same MAC rate from scalar v_fma_f32 (2.9-3.1 cycles each at >= 2 waves) - ~20-24 fp32 MACs per clock and SIMD whatever the
// form, so the 49 152 MACs of a step cost >= 510-610 cycles on the four SIMDs of one CU.  Earlier measurements: round 1's
// layout - lane = (unit, K-half), 16 ds_read_b128 per lane - 0.276 ms; 192 v_fmac_f32 instead of 96 packed FMAs 1.05
// us/step; four units per lane (4 reads, a longer reduction) 0.236 ms; eight waves (two per SIMD: reduction and gate math
// are issued twice per SIMD) 0.285 ms; K-quarters interleaved at 16 B (the four addresses of a read in one 64 B line) 0.238
// ms; 4 K-slices + LDS partial-sum exchange + libm gates 0.457 ms (round 1).
#include "nws_common.h"
#include "mlp_few.h"

namespace {

constexpr int kH = NWS_HIDDEN;  // 128
constexpr int kXChunk = 1024;   // control frames staged in LDS at a time

// sum over the 16-lane row pairs (lane l <-> l^16) on the VALU (v_permlane16_swap), no LDS round trip
__device__ __forceinline__ float sum_rows16(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// workgroup barrier that orders LDS traffic only: the h_t global stores (and nothing else in this loop)
// need not be drained before the next step may start, which __syncthreads() would force (vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DBG != 0: timing ablations for nws_debug_control_gru (results wrong by design): 1 half the LDS reads of h, 2 half the
// FMAs, 3 gates without transcendentals, 4 no workgroup barrier, 5 no LDS reads of h at all, 6 correct results + a cycle
// timeline of steps 200..207 of wave 0 (s_memtime after: barrier, FMA loop, reduction, gates; twice back to back at the
// top for the probe's own cost) written over the head of gru_out[0] as int64
__device__ __forceinline__ unsigned long long gru_tick(float pin) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(pin) : "memory");
  return t;
}
template <int DBG = 0>
__global__ __launch_bounds__(256, 1) void control_gru_kernel(NwsWeights w, const float* __restrict__ control, int C, int T,
                                                             const float* __restrict__ h0, float* __restrict__ gru_out,
                                                             float* __restrict__ hT, const float* __restrict__ f0,
                                                             double* __restrict__ carry, NwsStreamSide side) {
  const int tid = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float h_lds[2][kH];
  __shared__ __attribute__((aligned(16))) float x_lds[2][kXChunk];  // control[:, 0:2] of the current chunk of frames
  __shared__ unsigned long long ticks[DBG == 6 ? 8 * 6 : 1];
  // fused control-rate prologue of a forward (nws_control_gru_carry): the oscillator phase carries of utterance b are the
  // work of the EXTRA workgroup B + b of the same launch, on a CU of its own beside the recurrences (8 us there; as a
  // prologue of the recurrence's own workgroup they delayed its first step by 19 us, as a separate 64-workgroup launch in
  // front of it on the control stream by 70 us under load)
  if (carry != nullptr && blockIdx.x >= gridDim.x / 2) {
    nws_phase_carry_block<4>(f0, nullptr, T, carry, blockIdx.x - gridDim.x / 2, tid, reinterpret_cast<double*>(&x_lds[0][0]));
    return;
  }
  // streaming hop (nws_control_gru_stream): workgroups behind the B recurrences do what in the hop depends on nothing the hop
  // computes - the per-utterance head (windows' first rows, phase carries), then the reverb's history parts (256 taps x 256
  // outputs each, earlier hops' reverb input only) - beside the hop's critical path instead of on it (stream.hip, nws_common.h)
  if (DBG == 0 && (side.ring != nullptr || side.f0_w != nullptr) && blockIdx.x >= (unsigned)side.B) {
    int idx = blockIdx.x - side.B;
    if (side.f0_w != nullptr) {
      if (idx < side.B) {
        nws_stream_head_block(side, idx, tid, reinterpret_cast<double*>(&x_lds[0][0]));
        return;
      }
      idx -= side.B;
    }
    if (side.gru_flag != nullptr) {
      // frame MLPs of the hop's new frames: (path, utterance) = (idx & 1, idx >> 1).  Dispatched behind ALL recurrence workgroups
      // (lower block ids), so the wait below cannot keep one of them from starting; it is bounded anyway (a hop of NaNs instead of
      // a hung queue; counters[5] says so too)
      if (idx < 2 * side.B) {
        __shared__ __attribute__((aligned(16))) NwsFewLds FL;
        const int mb = idx >> 1;
        const long long target = side.counters[1] + side.counters[3] + side.K;
        __shared__ int gave_up;
        nws_mlp_few_path<2>(FL, w, side.gru_out, side.K, mb, idx & 1, side.film_w, side.fir_w, side.out_T, side.out_off, tid, [&] {
          if (tid == 0) {
            int spins = 0, lost = 0;
            while (__hip_atomic_load(&side.gru_flag[mb], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != target) {
              if (++spins > (1 << 18)) {      // ~0.3 s: the recurrence of a hop takes microseconds
                side.counters_rw[5] = 1;
                lost = 1;
                break;
              }
              __builtin_amdgcn_s_sleep(4);
            }
            gave_up = lost;
          }
          __syncthreads();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // gru_out as the recurrence workgroup left it
          return gave_up == 0;
        });
        return;
      }
      idx -= 2 * side.B;
      if (idx == 0) {
        nws_stream_noise_window_block<256, false>(side.win, tid);
        return;
      }
      idx -= 1;
    }
    __shared__ __attribute__((aligned(16))) float rv_xs[512];
    __shared__ __attribute__((aligned(16))) float rv_hs[256];
    const int sb = idx % side.B, p = 1 + idx / side.B;
    const long long pos = side.counters[0] + side.counters[2];
    const NwsPreSrc none{};
    const float v = nws_stream_reverb_partial<true>(side.ring, none, side.ir, side.ir_len, side.M, sb, p, 0, pos, tid, rv_xs, rv_hs);
    if (tid < side.M) side.partial[((size_t)p * side.B + sb) * side.M + tid] = v;
    return;
  }
  // latency-bound and usually sharing its SIMDs with throughput kernels of other streams (ForwardPipeline): ask for issue
  // priority, the recurrence is the critical path of the pipelined step
  __builtin_amdgcn_s_setprio(3);
  const int b = blockIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int a = lane >> 5;
  const int sl = 2 * a + ((lane >> 4) & 1);    // K-slice
  const int ua = 32 * wave + 2 * (lane & 15);  // FMA phase: units ua and ua + 1
  const int unit = ua + a;                     // gate phase: this lane's unit
  const bool writer = (lane & 16) == 0;        // one of the two replicas stores

  // wreg[u][g][c] = W_hh[g*128 + ua + u][32 sl + 2c, +1] (times the gate's constant, below)
  constexpr float kSig = -1.4426950408889634f, kTanh = 2.8853900817779268f;
  f32x2 wreg[2][3][16];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4* src = reinterpret_cast<const float4*>(w.gru_w_hh + (size_t)(g * kH + ua + u) * kH + 32 * sl);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 v = src[q];
        const float gs = g == 2 ? kTanh : kSig;
        wreg[u][g][2 * q + 0] = f32x2{gs * v.x, gs * v.y};
        wreg[u][g][2 * q + 1] = f32x2{gs * v.z, gs * v.w};
      }
    }
  // every gate's weights and biases carry the constant its nonlinearity multiplies the argument with anyway (-log2 e for
  // the sigmoids, 2 log2 e for tanh): the sums feed v_exp_f32 directly (one rounding per weight, the class of the FMA
  // roundings of the row sums themselves; same yard-stick against a float64 GRU as before)
  const float wi_r0 = kSig * w.gru_w_ih[unit * 2 + 0], wi_r1 = kSig * w.gru_w_ih[unit * 2 + 1];
  const float wi_z0 = kSig * w.gru_w_ih[(kH + unit) * 2 + 0], wi_z1 = kSig * w.gru_w_ih[(kH + unit) * 2 + 1];
  const float wi_n0 = kTanh * w.gru_w_ih[(2 * kH + unit) * 2 + 0], wi_n1 = kTanh * w.gru_w_ih[(2 * kH + unit) * 2 + 1];
  const float b_r = kSig * (w.gru_b_ih[unit] + w.gru_b_hh[unit]), b_z = kSig * (w.gru_b_ih[kH + unit] + w.gru_b_hh[kH + unit]);
  const float bi_n = kTanh * w.gru_b_ih[2 * kH + unit], bh_n = kTanh * w.gru_b_hh[2 * kH + unit];

  float h_prev = h0 != nullptr ? h0[(size_t)b * kH + unit] : 0.0f;
  if (tid < kH) h_lds[0][tid] = h0 != nullptr ? h0[(size_t)b * kH + tid] : 0.0f;
  const float* x0p = control + ((size_t)b * C + 0) * T;
  const float* x1p = control + ((size_t)b * C + 1) * T;

  for (int t0 = 0; t0 < T; t0 += kXChunk) {
    const int nt = T - t0 < kXChunk ? T - t0 : kXChunk;
    __syncthreads();
    for (int i = tid; i < nt; i += 256) {
      x_lds[0][i] = x0p[t0 + i];
      x_lds[1][i] = x1p[t0 + i];
    }
    __syncthreads();
    for (int tt = 0; tt < nt; ++tt) {
      const int t = t0 + tt;
      const int cur = t & 1;
      const bool probe = DBG == 6 && t >= 200 && t < 208;
      unsigned long long tk[6] = {0, 0, 0, 0, 0, 0};
      if (probe) {
        tk[0] = gru_tick(h_prev);
        tk[1] = gru_tick(h_prev);
      }
      const float x0 = x_lds[0][tt], x1 = x_lds[1][tt];
      const float ir = fmaf(wi_r1, x1, fmaf(wi_r0, x0, b_r));
      const float iz = fmaf(wi_z1, x1, fmaf(wi_z0, x0, b_z));
      const float in = fmaf(wi_n1, x1, fmaf(wi_n0, x0, bi_n));
      const float4* hp = reinterpret_cast<const float4*>(&h_lds[cur][32 * sl]);
      f32x2 acc[2][3];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[u][g] = f32x2{0.0f, 0.0f};
      // four read buffers in rotation: the read of group q + 4 goes out as soon as group q has been consumed and stays
      // there (left alone, hipcc issued reads 5..8 in pairs right in front of their first use: two exposed LDS latencies)
      float4 hb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) hb[i] = hp[DBG == 1 ? (i & ~1) : i];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 hv = DBG == 5 ? float4{h_prev, x0, x1, h_prev} : hb[q & 3];
        const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[u][g] = __builtin_elementwise_fma(wreg[u][g][2 * q], h01, acc[u][g]);
        if (DBG == 2) {
          if (q + 4 < 8) hb[q & 3] = hp[q + 4];
          continue;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int g = 0; g < 3; ++g) acc[u][g] = __builtin_elementwise_fma(wreg[u][g][2 * q + 1], h23, acc[u][g]);
        if (q + 4 < 8) hb[q & 3] = hp[DBG == 1 ? ((q + 4) & ~1) : q + 4];
        __builtin_amdgcn_sched_barrier(0);
      }
      if (DBG == 6) {   // every accumulator is an input: no FMA can sink below the probe
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[0][1]), "v"(acc[0][2]), "v"(acc[1][0]), "v"(acc[1][1]), "v"(acc[1][2]));
        if (probe) tk[2] = gru_tick(acc[0][0].x);
      }
      float sg[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        const float pa = nws_add_scalar(acc[0][g].x, acc[0][g].y), pb = nws_add_scalar(acc[1][g].x, acc[1][g].y);
        // lanes a=0 receive both halves of the first unit's partial sums, lanes a=1 of the second's
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(pa), __float_as_uint(pb), false, false);
        sg[g] = sum_rows16(__uint_as_float(r[0]) + __uint_as_float(r[1]));
      }
      // sigmoid(x) = 1 / (1 + 2^(-x log2 e));  tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)) (saturates correctly: exp2 -> inf or 0);
      // hardware exp2 + reciprocal (each ~1 ulp): the error class of the libm forms inside torch's CPU GRU, a fraction of
      // their latency on the sequential critical path
      if (probe) tk[3] = gru_tick(sg[0] + sg[1] + sg[2]);
      auto ex2 = [](float v) { return DBG == 3 ? 0.01f * v : __builtin_amdgcn_exp2f(v); };
      auto rcp = [](float v) { return DBG == 3 ? 0.5f * v : __builtin_amdgcn_rcpf(v); };
      const float r = rcp(1.0f + ex2(ir + sg[0]));
      const float z = rcp(1.0f + ex2(iz + sg[1]));
      // h' = z h + (1 - z) n with n = 1 - 2 q:  h' = [z h + (1 - z)] - 2 (1 - z) q.  The bracket and the factor only need z,
      // so ONE FMA follows the n gate's reciprocal; n itself (a rounding at magnitude 1) is never formed: against a float64
      // GRU 1.9e-5 max-abs over 500 steps where "(h - n) z + n" gave 5.3e-5 and torch's CPU GRU 5.2e-5 (tools/gru_variants.py)
      const float omz = 1.0f - z;
      const float base = fmaf(z, h_prev, omz), fac = -2.0f * omz;
      const float hnew = fmaf(fac, rcp(1.0f + ex2(fmaf(r, sg[2] + bh_n, in))), base);
      h_prev = hnew;
      if (probe) tk[4] = gru_tick(hnew);
      if (writer) {
        h_lds[cur ^ 1][unit] = hnew;
        gru_out[((size_t)b * T + t) * kH + unit] = hnew;
      }
      if (DBG != 4) lds_barrier();
      if (probe) {
        tk[5] = gru_tick(hnew);
        if (tid == 0)
          for (int i = 0; i < 6; ++i) ticks[(t - 200) * 6 + i] = tk[i];
      }
    }
  }
  if (DBG == 6) {
    __syncthreads();
    if (b == 0 && tid < 48 && T >= 208) reinterpret_cast<unsigned long long*>(gru_out)[tid] = ticks[tid];
  }
  if (hT != nullptr && writer) hT[(size_t)b * kH + unit] = h_prev;
  if (DBG == 0 && side.gru_flag != nullptr) {
    // every gru_out row of this utterance is in memory: release the frame-MLP workgroups waiting for it in this launch
    __syncthreads();
    if (tid == 0)
      __hip_atomic_store(&side.gru_flag[b], side.counters[1] + side.counters[3] + side.K, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched form (nws_control_gru_batched; the throughput pipeline uses it): 16 utterances per workgroup, the recurrent product on the matrix cores.
//   gates (384 x 16) = W_hh (384 x 128) * H (128 x 16 utterances)   per step,
// v_mfma_f32_16x16x32_f16 with the two-term fp16 split on both operands (three MFMAs per product, fp32 accumulate).
// 8 waves: wave (w, sub) owns hidden units [32w + 16 sub, +16) for all three gates: 3 tiles x 4 K-steps, W_hh fragments
// (hi and lo, 96 VGPRs) resident in registers for the whole sequence; two waves share a SIMD, so one wave's gate math
// (VALU + transcendental unit) overlaps the other's MFMAs.  The h exchange goes through LDS as fp16 hi/lo rows per
// utterance (B-fragment order is 8 consecutive units per lane: one ds_read_b128 each).
// Every gate's weights and biases carry the constant its nonlinearity multiplies the argument with anyway (-log2 e for the
// sigmoids, 2 log2 e for tanh), so the accumulators feed v_exp_f32 directly.  (The lo parts of small weights fall into the
// fp16 subnormals, quantised at 6e-8 absolute: over a 128-term row that is ~6e-8 on a pre-activation, the size of the fp32
// rounding noise of the row sum itself.)  The two waves sharing a SIMD run at different priorities so that one's gate math
// overlaps the other's MFMAs instead of both phases colliding.
// Why it exists: the per-utterance kernel above spends ~600 wave-instructions per utterance-step (~15 % of the oscillator
// kernel's VALU work at B = 64) on B workgroups; this one ~1/10 of that per utterance on B/16 workgroups.
// Measured (MI355X, B = 64, T = 500): alone 0.553 ms against 0.280 ms for the per-utterance kernel (per step 72 MFMAs and
// the gate math of two waves serialise on each SIMD: VALU-active + MFMA-busy cycles add up to the step time); inside
// ForwardPipeline, beside the all-CU kernels of other streams, 0.74 ms against 0.38 ms, and the pipelined step comes out
// the same (0.545 vs 0.536 ms).  So the per-utterance kernel stays the default everywhere; this one is the better citizen
// when B is large enough for B workgroups to matter (B >= 256) and is kept as an explicit entry point.
constexpr int kGU = 16;        // utterances per workgroup (MFMA N)
constexpr int kGChunk = 128;   // control frames staged in LDS at a time
constexpr int kHRow = 136;     // halfs per utterance row of the h buffers: 272 B stride, conflict-free ds_read_b128
constexpr float kLog2e = 1.4426950408889634f;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Two-term fp16 split of a set-up constant.  The residual of a small weight is an fp16 SUBNORMAL (|lo| < 2^-14 for
// |w| < 2^-3); it must survive: v_cvt_f16_f32 produces it, whereas v_fma_mixlo_f16 - which the compiler picks for
// "(half)(v - (float)hi)" when the SLP vectoriser is off - flushes it to zero, and the recurrence then ran with ~12-bit
// weights (error 2e-2 instead of 1.5e-3 against a float64 GRU on the B=40, T=300 test).  Hence the explicit instruction.
__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  const float d = v - (float)hi;
  asm("v_cvt_f16_f32 %0, %1" : "=v"(lo) : "v"(d));
}

__global__ __launch_bounds__(512, 1) void control_gru_mfma_kernel(NwsWeights w, const float* __restrict__ control, int B,
                                                                  int C, int T, const float* __restrict__ h0,
                                                                  float* __restrict__ gru_out, float* __restrict__ hT,
                                                                  int upb /* utterances per workgroup, <= kGU */) {
  const int b0 = blockIdx.x * upb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the sequence is latency-bound and may share its SIMDs with throughput kernels of other streams: ask for issue priority
  // (waves w and w+4 share a SIMD: the lower one goes first)
  if (wave < 4) __builtin_amdgcn_s_setprio(1);
  const int col = lane & 15;  // A operand: row inside the tile;  B operand / D: utterance
  const int rg = lane >> 4;   // A/B operand: K quarter (8 values);  D: row group (4 rows)
  const int ubase = 16 * wave;  // first of this wave's 16 hidden units
  const int b = b0 + col;     // this lane's utterance (D layout)
  const bool live = col < upb && b < B;

  __shared__ __attribute__((aligned(16))) _Float16 hhi[2][kGU][kHRow];
  __shared__ __attribute__((aligned(16))) _Float16 hlo[2][kGU][kHRow];
  __shared__ float xs[2][kGU][kGChunk];

  const float gscale[3] = {-kLog2e, -kLog2e, 2.0f * kLog2e};
  // W_hh fragments: tile g = rows g*128 + ubase + [0,16), K-step ks = columns 32 ks + 8 rg + [0,8)
  f16x8 whi[3][4], wlo[3][4];
#pragma unroll
  for (int g = 0; g < 3; ++g)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float* src = w.gru_w_hh + (size_t)(g * kH + ubase + col) * kH + 32 * ks + 8 * rg;
      const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
      const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        _Float16 hi, lo;
        split_f16(x[e] * gscale[g], hi, lo);
        whi[g][ks][e] = hi;
        wlo[g][ks][e] = lo;
      }
    }
  // gate constants of the 4 units this lane owns in the D layout (unit ubase + 4 rg + r), scaled like the weights
  float wr0[4], wr1[4], br[4], wz0[4], wz1[4], bz[4], wn0[4], wn1[4], bin[4], bhn[4], hprev[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int u = ubase + 4 * rg + r;
    wr0[r] = gscale[0] * w.gru_w_ih[u * 2 + 0];
    wr1[r] = gscale[0] * w.gru_w_ih[u * 2 + 1];
    br[r] = gscale[0] * (w.gru_b_ih[u] + w.gru_b_hh[u]);
    wz0[r] = gscale[1] * w.gru_w_ih[(kH + u) * 2 + 0];
    wz1[r] = gscale[1] * w.gru_w_ih[(kH + u) * 2 + 1];
    bz[r] = gscale[1] * (w.gru_b_ih[kH + u] + w.gru_b_hh[kH + u]);
    wn0[r] = gscale[2] * w.gru_w_ih[(2 * kH + u) * 2 + 0];
    wn1[r] = gscale[2] * w.gru_w_ih[(2 * kH + u) * 2 + 1];
    bin[r] = gscale[2] * w.gru_b_ih[2 * kH + u];
    bhn[r] = gscale[2] * w.gru_b_hh[2 * kH + u];
    hprev[r] = (h0 != nullptr && live) ? h0[(size_t)b * kH + u] : 0.0f;
  }
  {  // initial state into the exchange buffer (fp16 hi/lo), 4 consecutive units per 8-byte write
    f16x4 hi4, lo4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      _Float16 hi, lo;
      split_f16(hprev[r], hi, lo);
      hi4[r] = hi;
      lo4[r] = lo;
    }
    *reinterpret_cast<f16x4*>(&hhi[0][col][ubase + 4 * rg]) = hi4;
    *reinterpret_cast<f16x4*>(&hlo[0][col][ubase + 4 * rg]) = lo4;
  }
  for (int t0 = 0; t0 < T; t0 += kGChunk) {
    const int nt = T - t0 < kGChunk ? T - t0 : kGChunk;
    __syncthreads();  // previous chunk consumed; first pass: initial state published
    for (int i = tid; i < 2 * kGU * kGChunk; i += 512) {
      const int ch = i / (kGU * kGChunk), rem = i - ch * (kGU * kGChunk);
      const int u = rem / kGChunk, tt = rem - u * kGChunk;
      xs[ch][u][tt] = (u < upb && b0 + u < B && tt < nt) ? control[((size_t)(b0 + u) * C + ch) * T + t0 + tt] : 0.0f;
    }
    __syncthreads();
    for (int tt = 0; tt < nt; ++tt) {
      const int t = t0 + tt;
      const int cur = t & 1;
      // B fragments of h_{t-1}: K-step ks = units 32 ks + 8 rg + [0,8) of utterance `col`
      f16x8 bhi[4], blo[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bhi[ks] = *reinterpret_cast<const f16x8*>(&hhi[cur][col][32 * ks + 8 * rg]);
        blo[ks] = *reinterpret_cast<const f16x8*>(&hlo[cur][col][32 * ks + 8 * rg]);
      }
      const float x0 = xs[0][col][tt], x1 = xs[1][col][tt];
      // accumulators start from the input-side terms (r, z: both biases; n: only b_hn, its input side stays apart)
      f32x4v ar, az, an;
      float inn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        ar[r] = fmaf(wr1[r], x1, fmaf(wr0[r], x0, br[r]));
        az[r] = fmaf(wz1[r], x1, fmaf(wz0[r], x0, bz[r]));
        an[r] = bhn[r];
        inn[r] = fmaf(wn1[r], x1, fmaf(wn0[r], x0, bin[r]));
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        ar = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[0][ks], bhi[ks], ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[1][ks], bhi[ks], az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[2][ks], bhi[ks], an, 0, 0, 0);
        ar = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[0][ks], blo[ks], ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[1][ks], blo[ks], az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[2][ks], blo[ks], an, 0, 0, 0);
        ar = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[0][ks], bhi[ks], ar, 0, 0, 0);
        az = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[1][ks], bhi[ks], az, 0, 0, 0);
        an = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlo[2][ks], bhi[ks], an, 0, 0, 0);
      }
      float hn[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // ar = -log2(e) a_r: r = 1/(1 + 2^ar);  likewise z;  an = 2 log2(e) (W_hn h + b_hn), inn = 2 log2(e) (W_in x + b_in):
        // tanh(a) = 1 - 2 / (1 + 2^(2 log2(e) a)), saturating correctly when the exponential over- or underflows
        const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ar[r]));
        const float zz = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(az[r]));
        const float nn = fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(rr, an[r], inn[r]))), 1.0f);
        hn[r] = (hprev[r] - nn) * zz + nn;
        hprev[r] = hn[r];
      }
      const float4 hv = make_float4(hn[0], hn[1], hn[2], hn[3]);
      f16x4 hi4, lo4;
#pragma unroll
      for (int p = 0; p < 2; ++p) {  // hi by v_cvt_pk_f16_f32, lo by one v_fma_mix{lo,hi}_f16 each (exact residual, rounded once)
        const f32x2 v = {hn[2 * p], hn[2 * p + 1]};
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
        const f16x2 h2 = __builtin_convertvector(v, f16x2);
        const unsigned hp = __builtin_bit_cast(unsigned, h2);
        unsigned lp;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(v.x));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(v.y));
        const f16x2 l2 = __builtin_bit_cast(f16x2, lp);
        hi4[2 * p] = h2.x;
        hi4[2 * p + 1] = h2.y;
        lo4[2 * p] = l2.x;
        lo4[2 * p + 1] = l2.y;
      }
      const int u0 = ubase + 4 * rg;
      *reinterpret_cast<f16x4*>(&hhi[cur ^ 1][col][u0]) = hi4;
      *reinterpret_cast<f16x4*>(&hlo[cur ^ 1][col][u0]) = lo4;
      if (live) *reinterpret_cast<float4*>(&gru_out[((size_t)b * T + t) * kH + u0]) = hv;
      lds_barrier();
    }
  }
  if (hT != nullptr && live) {
#pragma unroll
    for (int r = 0; r < 4; ++r) hT[(size_t)b * kH + ubase + 4 * rg + r] = hprev[r];
  }
}

}  // namespace

extern "C" int nws_control_gru_state(const NwsWeights* w, const float* control, int B, int C, int T, const float* h0,
                                     float* gru_out, float* hT, void* stream) {
  if (!w || !w->gru_w_ih || !w->gru_w_hh || !w->gru_b_ih || !w->gru_b_hh || !control || !gru_out) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
  control_gru_kernel<0><<<B, 256, 0, (hipStream_t)stream>>>(*w, control, C, T, h0, gru_out, hT, nullptr, nullptr, NwsStreamSide{});
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// nws_control_gru_state + the reverb's history parts of a streaming hop as extra workgroups of the same launch (nws_common.h)
extern "C" int nws_control_gru_stream(const NwsWeights* w, const float* control, int B, int C, int T, const float* h0,
                                      float* gru_out, float* hT, const NwsStreamSide* side, void* stream) {
  if (!w || !w->gru_w_ih || !w->gru_w_hh || !w->gru_b_ih || !w->gru_b_hh || !control || !gru_out) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
  NwsStreamSide s{};
  int extra = 0;
  if (side != nullptr) {
    s = *side;
    if (s.B != B) return NWS_ERR_BAD_ARG;
    if (s.f0_w != nullptr) {
      if (!s.f0_new || !s.prev_f0 || !s.prev_film || !s.prev_fir || !s.S || !s.film_w || !s.fir_w || !s.carry || s.K != T)
        return NWS_ERR_BAD_ARG;
      extra += B;
    }
    if (s.gru_flag != nullptr) {
      if (s.f0_w == nullptr || !s.gru_out || !s.counters || !s.counters_rw || !s.win.nzwin || !w->mlp_frags || s.K > 2 || s.gru_out != gru_out)
        return NWS_ERR_BAD_ARG;
      extra += 2 * B + 1;
    }
    if (s.ring != nullptr && s.parts > 1) {
      if (!s.ir || !s.partial || !s.counters || s.M <= 0 || s.M > 256) return NWS_ERR_BAD_ARG;
      extra += (s.parts - 1) * B;
    } else {
      s.ring = nullptr;
    }
  }
  control_gru_kernel<0><<<B + extra, 256, 0, (hipStream_t)stream>>>(*w, control, C, T, h0, gru_out, hT, nullptr, nullptr, s);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_control_gru_carry(const NwsWeights* w, const float* control, const float* f0, int B, int C, int T,
                                     float* gru_out, double* carry_out, void* stream) {
  if (!w || !w->gru_w_ih || !w->gru_w_hh || !w->gru_b_ih || !w->gru_b_hh || !control || !gru_out || !f0 || !carry_out)
    return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
  control_gru_kernel<0><<<2 * B, 256, 0, (hipStream_t)stream>>>(*w, control, C, T, nullptr, gru_out, nullptr, f0, carry_out, NwsStreamSide{});
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_debug_control_gru(int variant, const NwsWeights* w, const float* control, int B, int C, int T,
                                     float* gru_out, void* stream) {
  if (!w || !w->gru_w_hh || !control || !gru_out || B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
#define NWS_GRU_DBG(V) control_gru_kernel<V><<<B, 256, 0, (hipStream_t)stream>>>(*w, control, C, T, nullptr, gru_out, nullptr, nullptr, nullptr, NwsStreamSide{})
  switch (variant) {
    case 0: NWS_GRU_DBG(0); break;
    case 1: NWS_GRU_DBG(1); break;
    case 2: NWS_GRU_DBG(2); break;
    case 3: NWS_GRU_DBG(3); break;
    case 4: NWS_GRU_DBG(4); break;
    case 5: NWS_GRU_DBG(5); break;
    case 6: NWS_GRU_DBG(6); break;
    default: return NWS_ERR_BAD_ARG;
  }
#undef NWS_GRU_DBG
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_control_gru_batched(const NwsWeights* w, const float* control, int B, int C, int T, const float* h0,
                                       float* gru_out, float* hT, void* stream) {
  if (!w || !w->gru_w_ih || !w->gru_w_hh || !w->gru_b_ih || !w->gru_b_hh || !control || !gru_out) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
  // (fewer utterances per workgroup -- half-empty MFMA tiles, one workgroup on each XCD for B = 64 -- measured no
  // different beside the all-CU kernels of other streams: 0.544 vs 0.548 ms per pipelined step)
  const int upb = kGU;
  control_gru_mfma_kernel<<<(B + upb - 1) / upb, 512, 0, (hipStream_t)stream>>>(*w, control, B, C, T, h0, gru_out, hT, upb);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_control_gru(const NwsWeights* w, const float* control, int B, int C, int T, float* gru_out,
                               void* stream) {
  return nws_control_gru_state(w, control, B, C, T, nullptr, gru_out, nullptr, stream);
}
