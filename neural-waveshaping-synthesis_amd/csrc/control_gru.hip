// Control encoder GRU(2 -> 128), persistent over the T control frames.
//
// Replaces aten::gru as called by ControlModule.forward (models/neural_waveshaping.py:24-25) on
// control[:, 0:2] (get_embedding, :69-72): gate order [r; z; n], h0 = 0,
//   r = sigmoid(W_ir x + b_ir + W_hr h + b_hr),  z likewise,
//   n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (h - n) * z + n.
//
// Design: the recurrence is strictly sequential, so one workgroup owns one
// utterance for all T steps and W_hh (384x128 fp32 = 192 KB, more than the 160 KB LDS) lives in
// VGPRs: wave w holds the 32-column slice [32w, 32w+32) of all 384 rows, 6 rows per lane =
// 192 VGPRs.  Per step every wave reads only its 32 h values (8 broadcast ds_read_b128), does
// 192 FMAs per lane, and the four K-slices of each row are reduced through a conflict-free LDS
// exchange by the 128 gate lanes.  Two workgroup barriers per step, no global traffic on the
// critical path except the (prefetched) 2 control values and the 512 B h_t store.
#include "nws_common.h"

namespace {

constexpr int kH = NWS_HIDDEN;  // 128
constexpr int kG = 3 * kH;      // 384 gate rows

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256, 1) void control_gru_kernel(NwsWeights w, const float* __restrict__ control, int C,
                                                             int T, float* __restrict__ gru_out) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  __shared__ __attribute__((aligned(16))) float h_lds[kH];
  __shared__ __attribute__((aligned(16))) float part[4][kG];

  // W_hh slice in registers: wreg[i][c] = W_hh[lane + 64 i][32 wave + c]
  float wreg[6][32];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float4* src = reinterpret_cast<const float4*>(w.gru_w_hh + (size_t)(lane + 64 * i) * kH + 32 * wave);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = src[q];
      wreg[i][4 * q + 0] = v.x;
      wreg[i][4 * q + 1] = v.y;
      wreg[i][4 * q + 2] = v.z;
      wreg[i][4 * q + 3] = v.w;
    }
  }

  // gate lanes (tid < 128 -> hidden unit j = tid) keep their input weights / biases in registers
  float wi_r0 = 0, wi_r1 = 0, wi_z0 = 0, wi_z1 = 0, wi_n0 = 0, wi_n1 = 0;
  float bi_r = 0, bi_z = 0, bi_n = 0, bh_r = 0, bh_z = 0, bh_n = 0;
  float h_prev = 0.0f;
  if (tid < kH) {
    wi_r0 = w.gru_w_ih[(tid)*2 + 0];
    wi_r1 = w.gru_w_ih[(tid)*2 + 1];
    wi_z0 = w.gru_w_ih[(kH + tid) * 2 + 0];
    wi_z1 = w.gru_w_ih[(kH + tid) * 2 + 1];
    wi_n0 = w.gru_w_ih[(2 * kH + tid) * 2 + 0];
    wi_n1 = w.gru_w_ih[(2 * kH + tid) * 2 + 1];
    bi_r = w.gru_b_ih[tid];
    bi_z = w.gru_b_ih[kH + tid];
    bi_n = w.gru_b_ih[2 * kH + tid];
    bh_r = w.gru_b_hh[tid];
    bh_z = w.gru_b_hh[kH + tid];
    bh_n = w.gru_b_hh[2 * kH + tid];
    h_lds[tid] = 0.0f;
  }
  const float* x0p = control + ((size_t)b * C + 0) * T;
  const float* x1p = control + ((size_t)b * C + 1) * T;
  float x0 = 0.0f, x1 = 0.0f;
  if (tid < kH) {
    x0 = x0p[0];
    x1 = x1p[0];
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    // prefetch next frame's controls (independent of h)
    float nx0 = 0.0f, nx1 = 0.0f;
    if (tid < kH && t + 1 < T) {
      nx0 = x0p[t + 1];
      nx1 = x1p[t + 1];
    }
    // this wave's 32 hidden values
    float hc[32];
    const float4* hp = reinterpret_cast<const float4*>(&h_lds[32 * wave]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 v = hp[q];
      hc[4 * q + 0] = v.x;
      hc[4 * q + 1] = v.y;
      hc[4 * q + 2] = v.z;
      hc[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
      for (int c = 0; c < 32; c += 2) {
        a0 = fmaf(wreg[i][c], hc[c], a0);
        a1 = fmaf(wreg[i][c + 1], hc[c + 1], a1);
      }
      part[wave][lane + 64 * i] = a0 + a1;
    }
    __syncthreads();
    if (tid < kH) {
      const float hr = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + bh_r;
      const float hz = ((part[0][kH + tid] + part[1][kH + tid]) + (part[2][kH + tid] + part[3][kH + tid])) + bh_z;
      const float hn =
          ((part[0][2 * kH + tid] + part[1][2 * kH + tid]) + (part[2][2 * kH + tid] + part[3][2 * kH + tid])) + bh_n;
      const float ir = fmaf(wi_r1, x1, fmaf(wi_r0, x0, bi_r));
      const float iz = fmaf(wi_z1, x1, fmaf(wi_z0, x0, bi_z));
      const float in = fmaf(wi_n1, x1, fmaf(wi_n0, x0, bi_n));
      const float r = sigmoidf_acc(ir + hr);
      const float z = sigmoidf_acc(iz + hz);
      const float nn = tanhf(in + r * hn);
      const float hnew = (h_prev - nn) * z + nn;
      h_prev = hnew;
      h_lds[tid] = hnew;
      gru_out[((size_t)b * T + t) * kH + tid] = hnew;
      x0 = nx0;
      x1 = nx1;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int nws_control_gru(const NwsWeights* w, const float* control, int B, int C, int T, float* gru_out,
                               void* stream) {
  if (!w || !w->gru_w_ih || !w->gru_w_hh || !w->gru_b_ih || !w->gru_b_hh || !control || !gru_out) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || C < 2) return NWS_ERR_BAD_ARG;
  control_gru_kernel<<<B, 256, 0, (hipStream_t)stream>>>(*w, control, C, T, gru_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}
