// Frame-rate dense path on the matrix cores: one workgroup = 32 control frames, 4 waves.
//
//   emb  = proj(gru_out)            Conv1d(128,128,1)          models/neural_waveshaping.py:26
//   film = newt.mlp(emb)            TimeDistributedMLP 128->128->128->128->256   shaping.py:53-55,68
//   H    = h_generator(emb)         TimeDistributedMLP 128->128->128->128->129   neural_waveshaping.py:58,82
//   fir  = window * roll(irfft(H)) = D * H   (zero-phase FIR design, generators.py:22-27)
//   TimeDistributedMLP = [Conv1x1 -> LayerNorm(channels, eps 1e-5) -> LeakyReLU(0.01)] x3 -> Conv1x1
//                                                                  models/modules/dynamic.py:11-40
//
// Design: every layer is a [M x 128] x [128 x 32 frames] product on
// v_mfma_f32_32x32x2_f32 (exact fp32).  Activations never leave LDS (X[channel][frame], row stride
// 33 -> conflict-free for both the transposing load and the B-operand reads); each wave owns one
// 32-row M-tile per pass, preloads its 32x128 weight slice into 64 VGPRs (lane (i,h) holds
// W[row i][64h .. 64h+63]; the K order is permuted to k(s,h) = 64h+s, which is free because the
// contraction order is arbitrary) and streams 64 MFMAs.  LayerNorm statistics are per frame =
// per accumulator column: in-lane over 16 registers, one half-swap, one 4-wave LDS exchange.
// Outputs that the sample-rate kernels read frame-major (film, fir) are transposed per wave
// through a 4 KB LDS patch so every global store is a full 128 B segment.
#include "nws_common.h"

namespace {

constexpr int kFT = 32;         // frames per workgroup
constexpr int kXS = 33;         // LDS row stride (floats)
constexpr int kRows = 132;      // rows per activation buffer (129 FIR bands padded to 132)
constexpr int kDK = 132;        // padded K of the FIR design matrix (row stride of D)
constexpr float kLnEps = 1e-5f;

struct MlpLds {
  float emb[kRows * kXS];
  float p0[kRows * kXS];
  float p1[kRows * kXS];
  float stage[4][kFT * kXS];
  float red[2][4][kFT];
};

// D-fragment row of accumulator register r for lane half h
__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// acc(32 x 32 frames) = W[row0 .. row0+32)[0 .. 2*KH) * X,   K permuted as k(s,h) = KH*h + s.
// W is row-major with row stride `ldw`; rows >= n_rows contribute zeros.
template <int KH>
__device__ __forceinline__ void gemm_tile(const float* __restrict__ W, int ldw, int row0, int n_rows,
                                          const float* __restrict__ X, int lane, f32x16& acc) {
  const int half = lane >> 5, col = lane & 31;
  const int row = row0 + col;
  float a[KH];
  if (row < n_rows) {
    const float* src = W + (size_t)row * ldw + KH * half;
    if ((KH % 4) == 0 && (ldw % 4) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
#pragma unroll
      for (int q = 0; q < KH / 4; ++q) {
        const float4 v = s4[q];
        a[4 * q + 0] = v.x;
        a[4 * q + 1] = v.y;
        a[4 * q + 2] = v.z;
        a[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < KH; ++q) a[q] = src[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < KH; ++q) a[q] = 0.0f;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const float* xb = X + (KH * half) * kXS + col;
#pragma unroll
  for (int s = 0; s < KH; ++s) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], xb[s * kXS], acc, 0, 0, 0);
  }
}

// hidden layer: Xout = LeakyReLU(LayerNorm(W Xin + b)); wave w owns channels [32w, 32w+32)
__device__ __forceinline__ void hidden_layer(MlpLds& L, const float* W, const float* bias, const float* ln_g,
                                             const float* ln_b, const float* Xin, float* Xout, int wave, int lane) {
  const int half = lane >> 5, col = lane & 31;
  f32x16 acc;
  gemm_tile<64>(W, NWS_HIDDEN, 32 * wave, NWS_HIDDEN, Xin, lane, acc);
  float v[16];
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[r] = acc[r] + bias[32 * wave + frag_row(r, half)];
    s += v[r];
  }
  s += nws_swap_halves(s);
  if (half == 0) L.red[0][wave][col] = s;
  __syncthreads();
  const float mean = ((L.red[0][0][col] + L.red[0][1][col]) + (L.red[0][2][col] + L.red[0][3][col])) * (1.0f / NWS_HIDDEN);
  float q = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float d = v[r] - mean;
    q = fmaf(d, d, q);
  }
  q += nws_swap_halves(q);
  if (half == 0) L.red[1][wave][col] = q;
  __syncthreads();
  const float var = ((L.red[1][0][col] + L.red[1][1][col]) + (L.red[1][2][col] + L.red[1][3][col])) * (1.0f / NWS_HIDDEN);
  const float rstd = 1.0f / sqrtf(var + kLnEps);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int c = 32 * wave + frag_row(r, half);
    const float y = (v[r] - mean) * rstd * ln_g[c] + ln_b[c];
    Xout[c * kXS + col] = nws_leaky_relu(y);
  }
  __syncthreads();
}

// write one 32-channel x 32-frame accumulator tile to a frame-major (.., T, ld) tensor with full
// 128 B segments: through this wave's private LDS patch, 2 frames x 32 channels per store.
__device__ __forceinline__ void store_tile_frame_major(float* patch, const float v[16], int lane, float* dst /* frame t0, channel c0 */,
                                                       int ld, int frames_valid) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int r = 0; r < 16; ++r) patch[col * kXS + frag_row(r, half)] = v[r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int f = 2 * it + half;
    if (f < frames_valid) dst[(size_t)f * ld + col] = patch[f * kXS + col];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void frame_mlps_kernel(NwsWeights w, const float* __restrict__ gru_out,
                                                         const float* __restrict__ fir_design, int T,
                                                         float* __restrict__ emb_out, float* __restrict__ film_out,
                                                         float* __restrict__ H_out, float* __restrict__ fir_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpLds& L = *reinterpret_cast<MlpLds*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kFT;
  const int frames_valid = T - t0 < kFT ? T - t0 : kFT;

  // ---- load the gru_out tile transposed: p0[c][f] = gru_out[b][t0+f][c] ----
  for (int e = tid; e < kFT * NWS_HIDDEN; e += 256) {
    const int f = e >> 7, c = e & 127;
    L.p0[c * kXS + f] = f < frames_valid ? gru_out[((size_t)b * T + t0 + f) * NWS_HIDDEN + c] : 0.0f;
  }
  // rows 128..131 of p1 are the zero padding of the FIR-design contraction (H lives in p1 later)
  for (int e = tid; e < (kRows - 128) * kXS; e += 256) L.p1[128 * kXS + e] = 0.0f;
  __syncthreads();

  // ---- emb = proj(gru_out) ----
  {
    f32x16 acc;
    gemm_tile<64>(w.proj_w, NWS_HIDDEN, 32 * wave, NWS_HIDDEN, L.p0, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * wave + frag_row(r, half);
      const float v = acc[r] + w.proj_b[c];
      L.emb[c * kXS + col] = v;
      if (emb_out != nullptr && col < frames_valid) emb_out[((size_t)b * NWS_HIDDEN + c) * T + t0 + col] = v;
    }
  }
  __syncthreads();

  // ---- film = newt.mlp(emb) ----
  hidden_layer(L, w.newt_mlp_w[0], w.newt_mlp_b[0], w.newt_ln_g[0], w.newt_ln_b[0], L.emb, L.p0, wave, lane);
  hidden_layer(L, w.newt_mlp_w[1], w.newt_mlp_b[1], w.newt_ln_g[1], w.newt_ln_b[1], L.p0, L.p1, wave, lane);
  hidden_layer(L, w.newt_mlp_w[2], w.newt_mlp_b[2], w.newt_ln_g[2], w.newt_ln_b[2], L.p1, L.p0, wave, lane);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int c0 = 32 * (wave + 4 * pass);
    f32x16 acc;
    gemm_tile<64>(w.newt_mlp_w[3], NWS_HIDDEN, c0, NWS_FILM_CH, L.p0, lane, acc);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + w.newt_mlp_b[3][c0 + frag_row(r, half)];
    store_tile_frame_major(L.stage[wave], v, lane, film_out + ((size_t)b * T + t0) * NWS_FILM_CH + c0, NWS_FILM_CH,
                           frames_valid);
  }
  __syncthreads();

  // ---- H = h_generator(emb) ----
  hidden_layer(L, w.hgen_w[0], w.hgen_b[0], w.hgen_ln_g[0], w.hgen_ln_b[0], L.emb, L.p0, wave, lane);
  hidden_layer(L, w.hgen_w[1], w.hgen_b[1], w.hgen_ln_g[1], w.hgen_ln_b[1], L.p0, L.p1, wave, lane);
  hidden_layer(L, w.hgen_w[2], w.hgen_b[2], w.hgen_ln_g[2], w.hgen_ln_b[2], L.p1, L.p0, wave, lane);
  // 129 outputs = 4 full M-tiles + row 128 (tile 4, done by wave 0); result -> p1 rows 0..128
  // (p1's hidden activations are dead after the layer above; rows 129..131 stay zero)
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1 && wave != 0) break;
    const int c0 = pass == 0 ? 32 * wave : 128;
    f32x16 acc;
    gemm_tile<64>(w.hgen_w[3], NWS_HIDDEN, c0, NWS_N_BANDS, L.p0, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = c0 + frag_row(r, half);
      if (c < NWS_N_BANDS) {
        const float v = acc[r] + w.hgen_b[3][c];
        L.p1[c * kXS + col] = v;
        if (H_out != nullptr && col < frames_valid) H_out[((size_t)b * T + t0 + col) * NWS_N_BANDS + c] = v;
      }
    }
  }
  __syncthreads();

  // ---- fir = D * H  (256 taps) ----
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int c0 = 32 * (wave + 4 * pass);
    f32x16 acc;
    gemm_tile<kDK / 2>(fir_design, kDK, c0, NWS_FIR_LEN, L.p1, lane, acc);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    store_tile_frame_major(L.stage[wave], v, lane, fir_out + ((size_t)b * T + t0) * NWS_FIR_LEN + c0, NWS_FIR_LEN,
                           frames_valid);
  }
}

// D[n][k] (256 x 132): fir[n] = window[n] * h0[(n - 128) mod 256],
//   h0[m] = irfft(H)[m] = (1/256) (H_0 + (-1)^m H_128 + 2 sum_{k=1}^{127} H_k cos(2 pi k m / 256))
__global__ void fir_design_kernel(const float* __restrict__ window, float* __restrict__ D) {
  const int n = blockIdx.x;
  const int k = threadIdx.x;
  if (k >= kDK) return;
  float v = 0.0f;
  if (k < NWS_N_BANDS) {
    const int m = (n - NWS_FIR_LEN / 2) & (NWS_FIR_LEN - 1);
    const int ph = (k * m) & (NWS_FIR_LEN - 1);
    const double c = cospi(2.0 * (double)ph / (double)NWS_FIR_LEN);
    const double scale = (k == 0 || k == NWS_FIR_LEN / 2) ? 1.0 : 2.0;
    v = (float)((double)window[n] * scale * c / (double)NWS_FIR_LEN);
  }
  D[n * kDK + k] = v;
}

}  // namespace

extern "C" {

int nws_fir_design_matrix(const float* window, float* D_out, void* stream) {
  if (!window || !D_out) return NWS_ERR_BAD_ARG;
  fir_design_kernel<<<NWS_FIR_LEN, 192, 0, (hipStream_t)stream>>>(window, D_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_frame_mlps(const NwsWeights* w, const float* gru_out, const float* fir_design, int B, int T, float* emb_out,
                   float* film_out, float* H_out, float* fir_out, void* stream) {
  if (!w || !gru_out || !fir_design || !film_out || !fir_out || B <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  if (!w->proj_w || !w->proj_b) return NWS_ERR_BAD_ARG;
  for (int i = 0; i < 4; ++i)
    if (!w->newt_mlp_w[i] || !w->newt_mlp_b[i] || !w->hgen_w[i] || !w->hgen_b[i]) return NWS_ERR_BAD_ARG;
  for (int i = 0; i < 3; ++i)
    if (!w->newt_ln_g[i] || !w->newt_ln_b[i] || !w->hgen_ln_g[i] || !w->hgen_ln_b[i]) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const dim3 grid((T + kFT - 1) / kFT, B);
  frame_mlps_kernel<<<grid, 256, sizeof(MlpLds), (hipStream_t)stream>>>(*w, gru_out, fir_design, T, emb_out, film_out,
                                                                        H_out, fir_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
