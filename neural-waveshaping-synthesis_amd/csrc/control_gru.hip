// Control encoder GRU(2 -> 128), persistent over the T control frames.
//
// Replaces aten::gru as called by ControlModule.forward (models/neural_waveshaping.py:24-25) on
// control[:, 0:2] (get_embedding, :69-72): gate order [r; z; n], h0 = 0,
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr),  z likewise,
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (h - n) * z + n.
//
// Design: the recurrence is strictly sequential, so one workgroup owns one
// utterance for all T steps and W_hh (384x128 fp32 = 192 KB, more than the 160 KB LDS) lives in
// VGPRs.  Wave w owns hidden units [32w, 32w+32); lane (j = l&31, kh = l>>5) holds the three gate
// rows of unit 32w+j restricted to columns [64kh, 64kh+64): 192 weights per lane.  Per step a lane
// reads its 64 h values (16 broadcast ds_read_b128), does 96 packed FMAs (v_pk_fma_f32), meets its
// other K-half with one v_permlane32_swap per gate, evaluates the gates with v_exp/v_rcp (1 ulp) and
// writes h' into the other half of a ping-pong LDS buffer: ONE workgroup barrier per step that waits
// on LDS only (raw s_barrier + lgkmcnt(0); the 512 B h_t global store is never drained on the critical
// path), control values staged in LDS 1024 frames at a time.
// Measured alternatives (MI355X, 500 steps): 4 K-slices + LDS partial-sum exchange + libm gates + two
// __syncthreads per step 0.457 ms; this kernel 0.271 ms; an 8-wave variant (2 waves/SIMD, lane = unit x
// K-quarter) 0.290 ms -- the step is bound by VALU issue (96 v_pk_fma_f32 + gate math per lane), not latency.
#include "nws_common.h"

namespace {

constexpr int kH = NWS_HIDDEN;  // 128
constexpr int kXChunk = 1024;   // control frames staged in LDS at a time

// cross-half (lane l <-> l^32) sum on the VALU (v_permlane32_swap), no LDS round trip
__device__ __forceinline__ float sum_halves(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// workgroup barrier that orders LDS traffic only: the h_t global stores (and nothing else in this loop)
// need not be drained before the next step may start, which __syncthreads() would force (vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sigmoid / tanh from the hardware exp2 + reciprocal (each ~1 ulp): same error class as the
// libm forms inside torch's CPU GRU, a fraction of their latency on the sequential critical path
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // tanh(x) = 1 - 2 / (exp(2x) + 1); saturates correctly for |x| large (exp2 -> inf or 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

__global__ __launch_bounds__(256, 1) void control_gru_kernel(NwsWeights w, const float* __restrict__ control, int C,
                                                             int T, const float* __restrict__ h0,
                                                             float* __restrict__ gru_out, float* __restrict__ hT) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int kh = lane >> 5;
  const int unit = 32 * wave + (lane & 31);

  __shared__ __attribute__((aligned(16))) float h_lds[2][kH];
  __shared__ float x_lds[2][kXChunk];  // control[:, 0:2] of the current chunk of frames

  // wreg[g][c] = W_hh[g*128 + unit][64 kh + c]
  f32x2 wreg[3][32];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const float4* src = reinterpret_cast<const float4*>(w.gru_w_hh + (size_t)(g * kH + unit) * kH + 64 * kh);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float4 v = src[q];
      wreg[g][2 * q + 0] = f32x2{v.x, v.y};
      wreg[g][2 * q + 1] = f32x2{v.z, v.w};
    }
  }
  const float wi_r0 = w.gru_w_ih[unit * 2 + 0], wi_r1 = w.gru_w_ih[unit * 2 + 1];
  const float wi_z0 = w.gru_w_ih[(kH + unit) * 2 + 0], wi_z1 = w.gru_w_ih[(kH + unit) * 2 + 1];
  const float wi_n0 = w.gru_w_ih[(2 * kH + unit) * 2 + 0], wi_n1 = w.gru_w_ih[(2 * kH + unit) * 2 + 1];
  const float bi_r = w.gru_b_ih[unit], bi_z = w.gru_b_ih[kH + unit], bi_n = w.gru_b_ih[2 * kH + unit];
  const float bh_r = w.gru_b_hh[unit], bh_z = w.gru_b_hh[kH + unit], bh_n = w.gru_b_hh[2 * kH + unit];

  // h0 == nullptr: zero initial state (the reference's stateless forward); streaming passes the carried state
  float h_prev = h0 != nullptr ? h0[(size_t)b * kH + unit] : 0.0f;
  if (tid < kH) h_lds[0][tid] = h0 != nullptr ? h0[(size_t)b * kH + tid] : 0.0f;
  const float* x0p = control + ((size_t)b * C + 0) * T;
  const float* x1p = control + ((size_t)b * C + 1) * T;

  for (int t0 = 0; t0 < T; t0 += kXChunk) {
    const int nt = T - t0 < kXChunk ? T - t0 : kXChunk;
    __syncthreads();  // previous chunk fully consumed (and h_lds[0] initialised on the first pass)
    for (int i = tid; i < nt; i += 256) {
      x_lds[0][i] = x0p[t0 + i];
      x_lds[1][i] = x1p[t0 + i];
    }
    __syncthreads();
    for (int tt = 0; tt < nt; ++tt) {
      const int t = t0 + tt;
      const int cur = t & 1;
      const float x0 = x_lds[0][tt], x1 = x_lds[1][tt];
      // input-side gate terms: independent of h, overlap the h reads
      const float ir = fmaf(wi_r1, x1, fmaf(wi_r0, x0, bi_r)) + bh_r;
      const float iz = fmaf(wi_z1, x1, fmaf(wi_z0, x0, bi_z)) + bh_z;
      const float in = fmaf(wi_n1, x1, fmaf(wi_n0, x0, bi_n));
      const float4* hp = reinterpret_cast<const float4*>(&h_lds[cur][64 * kh]);
      f32x2 ar = {0.0f, 0.0f}, az = {0.0f, 0.0f}, an = {0.0f, 0.0f};
      f32x2 br = {0.0f, 0.0f}, bz = {0.0f, 0.0f}, bn = {0.0f, 0.0f};
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float4 hv = hp[q];
        const f32x2 h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
        ar = __builtin_elementwise_fma(wreg[0][2 * q], h01, ar);
        az = __builtin_elementwise_fma(wreg[1][2 * q], h01, az);
        an = __builtin_elementwise_fma(wreg[2][2 * q], h01, an);
        br = __builtin_elementwise_fma(wreg[0][2 * q + 1], h23, br);
        bz = __builtin_elementwise_fma(wreg[1][2 * q + 1], h23, bz);
        bn = __builtin_elementwise_fma(wreg[2][2 * q + 1], h23, bn);
      }
      const float sr = sum_halves((ar.x + ar.y) + (br.x + br.y));
      const float sz = sum_halves((az.x + az.y) + (bz.x + bz.y));
      const float sn = sum_halves((an.x + an.y) + (bn.x + bn.y));
      // both K-halves now hold the full sums; both evaluate the gates (no divergence), half 0 stores
      const float r = fast_sigmoid(ir + sr);
      const float z = fast_sigmoid(iz + sz);
      const float nn = fast_tanh(in + r * (sn + bh_n));
      const float hnew = (h_prev - nn) * z + nn;
      h_prev = hnew;
      if (kh == 0) {
        h_lds[cur ^ 1][unit] = hnew;
        gru_out[((size_t)b * T + t) * kH + unit] = hnew;
      }
      lds_barrier();
    }
  }
  if (hT != nullptr && kh == 0) hT[(size_t)b * kH + unit] = h_prev;
}

}  // namespace

extern "C" int nws_control_gru_state(const NwsWeights* w, const float* control, int B, int C, int T, const float* h0,
                                     float* gru_out, float* hT, void* stream) {
  if (!w || !w->gru_w_ih || !w->gru_w_hh || !w->gru_b_ih || !w->gru_b_hh || !control || !gru_out) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
  control_gru_kernel<<<B, 256, 0, (hipStream_t)stream>>>(*w, control, C, T, h0, gru_out, hT);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_control_gru(const NwsWeights* w, const float* control, int B, int C, int T, float* gru_out,
                               void* stream) {
  return nws_control_gru_state(w, control, B, C, T, nullptr, gru_out, nullptr, stream);
}
