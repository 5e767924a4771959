// Stand-alone forms of the reference's small modules (models/modules/dynamic.py, generators.py:21-28): what
// TimeDistributedMLP.forward / TimeDistributedLayerNorm.forward / FiLM.forward / FIRNoiseSynth.forward do when a caller
// invokes them OUTSIDE NeuralWaveshaping.forward (inside it the same arithmetic is fused into frame_mlps.hip /
// exciter_newt.hip).  These are API-completeness paths: plain fp32 VALU kernels, any layer sizes, exact fp32 products.
#include "nws_common.h"

namespace {

constexpr int kFT = 32;         // frames per workgroup
constexpr int kMaxDepth = 8;

struct MlpArgs {
  const float* w[kMaxDepth];
  const float* b[kMaxDepth];
  const float* ln_g[kMaxDepth];
  const float* ln_b[kMaxDepth];
};

// One workgroup (4 waves) = one utterance x 32 frames; activations in LDS as X[channel][frame] (stride 33).  Layer i:
// Conv1d(k=1) -> [LayerNorm over channels (biased variance) -> LeakyReLU] except last.  The contraction of a layer runs on the
// matrix cores in exact fp32 (v_mfma_f32_32x32x2_f32: wave w takes the 32-channel output tiles w, w + 4, ...; lane half h of a
// 32-wide K chunk contracts k0 + 16 h .. k0 + 16 h + 15, the order of a sum being free, so every lane reads 16 consecutive
// weights of its row) - round 4: the per-thread FMA loop it replaces read every weight through the vector cache once per
// frame group (0.46 ms per call at 64 x 500 frames of the default sizes; runtime sizes, guards instead of padding).
__device__ __forceinline__ int td_frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__global__ __launch_bounds__(256) void td_mlp_kernel(MlpArgs a, const float* __restrict__ x, int in_size, int hidden,
                                                     int out_size, int depth, int T, float eps, float slope,
                                                     float* __restrict__ y) {
  extern __shared__ float lds[];
  // (the last layer's outputs leave from the accumulators: the planes hold inputs and hidden activations only, so a 128 -> 256
  // output layer does not halve the workgroups per CU - 35.8 instead of 69.6 KB at the default sizes, 4 instead of 2)
  const int maxw = in_size > hidden ? in_size : hidden;
  float* bufA = lds;                       // [maxw][33]
  float* bufB = lds + (size_t)maxw * 33;   // [maxw][33]
  float* red = bufB + (size_t)maxw * 33;   // [2][8][32]
  const int b = blockIdx.y, t0 = blockIdx.x * kFT;
  const int f = threadIdx.x & 31, cg = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
  const int t = t0 + f;
  const bool live = t < T;
  for (int c = cg; c < in_size; c += 8) bufA[c * 33 + f] = live ? x[((size_t)b * in_size + c) * T + t] : 0.0f;
  __syncthreads();
  float* cur = bufA;
  float* nxt = bufB;
  for (int layer = 0; layer < depth; ++layer) {
    const int cin = layer == 0 ? in_size : hidden;
    const int cout = layer == depth - 1 ? out_size : hidden;
    const float* __restrict__ W = a.w[layer];
    const float* __restrict__ bias = a.b[layer];
    const bool last = layer == depth - 1;
    const bool vec = (cin & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    for (int mt = wave; 32 * mt < cout; mt += 4) {
      const int row = 32 * mt + col;
      const float* wr = W + (size_t)(row < cout ? row : cout - 1) * cin;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
      for (int k0 = 0; k0 < cin; k0 += 32) {
        const int kb = k0 + 16 * half;
        float av[16];
        if (vec && kb + 16 <= cin) {
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 v4 = *reinterpret_cast<const float4*>(wr + kb + 4 * q4);
            av[4 * q4] = v4.x;
            av[4 * q4 + 1] = v4.y;
            av[4 * q4 + 2] = v4.z;
            av[4 * q4 + 3] = v4.w;
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) av[s2] = kb + s2 < cin ? wr[kb + s2] : 0.0f;
        }
        if (row >= cout) {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) av[s2] = 0.0f;
        }
        // the chunk's 16 activations first, all in flight together (as `k < cin ? cur[...] : 0` every read was predicated on its
        // own and waited for right in front of its MFMA: one LDS latency per MFMA, the matrix pipe a third busy); rows beyond
        // cin are read from row cin - 1 instead and meet zero weights
        float bv[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const int k = kb + s2 < cin ? kb + s2 : cin - 1;
          bv[s2] = cur[k * 33 + col];
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
      }
      if (last) {
        // accumulator register r of lane (col, half) = channel 32 mt + row(r, half), frame t0 + col: 32 lanes = 128 contiguous bytes
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = 32 * mt + td_frag_row(r, half);
          if (c < cout && live) y[((size_t)b * out_size + c) * T + t] = acc[r] + bias[c];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = 32 * mt + td_frag_row(r, half);
          if (c < cout) nxt[c * 33 + col] = acc[r] + bias[c];
        }
      }
    }
    if (last) break;
    __syncthreads();
    // LayerNorm over the `hidden` channels of frame f: mean, then variance about the mean (two passes, like ATen)
    float s1 = 0.0f;
    for (int c = cg; c < cout; c += 8) s1 += nxt[c * 33 + f];
    red[cg * 32 + f] = s1;
    __syncthreads();
    float mean = 0.0f;
    for (int g = 0; g < 8; ++g) mean += red[g * 32 + f];
    mean /= (float)cout;
    float s2 = 0.0f;
    for (int c = cg; c < cout; c += 8) {
      const float d = nxt[c * 33 + f] - mean;
      s2 = fmaf(d, d, s2);
    }
    red[256 + cg * 32 + f] = s2;
    __syncthreads();
    float var = 0.0f;
    for (int g = 0; g < 8; ++g) var += red[256 + g * 32 + f];
    const float rstd = 1.0f / sqrtf(var / (float)cout + eps);
    const float* __restrict__ gam = a.ln_g[layer];
    const float* __restrict__ bet = a.ln_b[layer];
    for (int c = cg; c < cout; c += 8) {
      const float v = (nxt[c * 33 + f] - mean) * rstd * gam[c] + bet[c];
      nxt[c * 33 + f] = v > 0.0f ? v : slope * v;
    }
    __syncthreads();
    float* tmp = cur;
    cur = nxt;
    nxt = tmp;
  }
}

// LayerNorm over the channel axis of (B, C, T) (TimeDistributedLayerNorm, dynamic.py:11-17): one thread per (b, t)
__global__ void td_layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ be,
                                     int C, int T, float eps, float* __restrict__ y) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* xp = x + (size_t)b * C * T + t;
  float s = 0.0f;
  for (int c = 0; c < C; ++c) s += xp[(size_t)c * T];
  const float mean = s / (float)C;
  float v = 0.0f;
  for (int c = 0; c < C; ++c) {
    const float d = xp[(size_t)c * T] - mean;
    v = fmaf(d, d, v);
  }
  const float rstd = 1.0f / sqrtf(v / (float)C + eps);
  float* yp = y + (size_t)b * C * T + t;
  for (int c = 0; c < C; ++c) yp[(size_t)c * T] = (xp[(size_t)c * T] - mean) * rstd * g[c] + be[c];
}

// FiLM.forward (dynamic.py:6-8): gamma * x + beta, element-wise on equally shaped tensors (multiply, then add: two roundings)
__global__ void film_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                            int64_t n, float* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = gamma[i] * x[i] + beta[i];
}

// FIR design of FIRNoiseSynth.forward (generators.py:22-28) from H (B, 129, T): fir[b][t][d] = sum_k D[128 + d][k] H[b][k][t], d < 128
// with D = window * roll(irfft(.), 128) folded into one (256, 132) matrix (nws_fir_design_matrix).
__global__ __launch_bounds__(128) void fir_from_h_kernel(const float* __restrict__ H, const float* __restrict__ D, int T,
                                                         float* __restrict__ fir) {
  __shared__ float hs[NWS_N_BANDS][kFT + 1];
  const int b = blockIdx.y, t0 = blockIdx.x * kFT;
  for (int i = threadIdx.x; i < NWS_N_BANDS * kFT; i += 128) {
    const int k = i / kFT, f = i - k * kFT;
    hs[k][f] = t0 + f < T ? H[((size_t)b * NWS_N_BANDS + k) * T + t0 + f] : 0.0f;
  }
  __syncthreads();
  const int n = NWS_FIR_HALF + threadIdx.x;  // tap 128 .. 255 (the lower half is the mirror image, include/nws_hip.h)
  float acc[kFT];
#pragma unroll
  for (int f = 0; f < kFT; ++f) acc[f] = 0.0f;
  for (int k = 0; k < NWS_N_BANDS; ++k) {
    const float d = D[n * 132 + k];
#pragma unroll
    for (int f = 0; f < kFT; ++f) acc[f] = fmaf(d, hs[k][f], acc[f]);
  }
#pragma unroll
  for (int f = 0; f < kFT; ++f)
    if (t0 + f < T) fir[((size_t)b * T + t0 + f) * NWS_FIR_HALF + threadIdx.x] = acc[f];
}

}  // namespace

extern "C" {

int nws_td_mlp(const float* x, int B, int in_size, int hidden, int out_size, int depth, int T, const float* const* w,
               const float* const* b, const float* const* ln_g, const float* const* ln_b, float ln_eps, float leaky_slope,
               float* y, void* stream) {
  if (!x || !y || !w || !b || !ln_g || !ln_b || B <= 0 || T <= 0 || in_size <= 0 || hidden <= 0 || out_size <= 0)
    return NWS_ERR_BAD_ARG;
  if (depth < 1 || depth > kMaxDepth || B > 65535) return NWS_ERR_UNSUPPORTED;
  MlpArgs a{};
  for (int i = 0; i < depth; ++i) {
    if (!w[i] || !b[i]) return NWS_ERR_BAD_ARG;
    a.w[i] = w[i];
    a.b[i] = b[i];
    if (i < depth - 1) {
      if (!ln_g[i] || !ln_b[i]) return NWS_ERR_BAD_ARG;
      a.ln_g[i] = ln_g[i];
      a.ln_b[i] = ln_b[i];
    }
  }
  const int maxw = in_size > hidden ? in_size : hidden;     // (the output layer stores from registers: see the kernel)
  const size_t lds = ((size_t)2 * maxw * 33 + 512) * sizeof(float);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;   // layer widths up to ~600
  static unsigned long long attr_devices = 0;
  if (nws_first_use_on_device(attr_devices)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(td_mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  td_mlp_kernel<<<dim3((T + kFT - 1) / kFT, B), 256, lds, (hipStream_t)stream>>>(a, x, in_size, hidden, out_size, depth, T,
                                                                                 ln_eps, leaky_slope, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_td_layer_norm(const float* x, const float* gain, const float* bias, int B, int C, int T, float eps, float* y,
                      void* stream) {
  if (!x || !gain || !bias || !y || B <= 0 || C <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  td_layer_norm_kernel<<<dim3((T + 127) / 128, B), 128, 0, (hipStream_t)stream>>>(x, gain, bias, C, T, eps, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_film(const float* x, const float* gamma, const float* beta, int64_t n, float* y, void* stream) {
  if (!x || !gamma || !beta || !y || n < 0) return NWS_ERR_BAD_ARG;
  if (n == 0) return NWS_OK;
  const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  film_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(x, gamma, beta, n, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_fir_from_h(const float* H, const float* fir_design, int B, int T, float* fir_out, void* stream) {
  if (!H || !fir_design || !fir_out || B <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  fir_from_h_kernel<<<dim3((T + kFT - 1) / kFT, B), 128, 0, (hipStream_t)stream>>>(H, fir_design, T, fir_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
