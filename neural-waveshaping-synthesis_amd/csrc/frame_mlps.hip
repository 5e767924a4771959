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
#include <cstdlib>
#include <cstring>

#include "nws_common.h"
#include "mlp_few.h"

namespace {

constexpr int kFT = 32;         // frames per workgroup
constexpr int kXS = 33;         // LDS row stride (floats)
constexpr int kPS = 36;         // output patch row stride (floats): 16-byte aligned rows for the vector read-back
constexpr int kRows = 132;      // rows per activation buffer (129 FIR bands padded to 132)
constexpr int kDK = 132;        // padded K of the FIR design matrix (row stride of D)
constexpr float kLnEps = 1e-5f;

struct MlpLds {
  float emb[kRows * kXS];
  float p0[kRows * kXS];
  float p1[kRows * kXS];
  float stage[4][kFT * kPS];
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
// 32 channels x 32 frames held in the D layout -> dst[frame][channel] (row stride ld): transposed through a per-wave LDS
// patch so that every global store instruction writes eight full 128 B frame segments (one dwordx4 per lane)
__device__ __forceinline__ void store_tile_frame_major(float* patch, const float v[16], int lane, float* dst /* frame t0, channel c0 */,
                                                       int ld, int frames_valid) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g)  // 4 consecutive channels of frame `col`
    *reinterpret_cast<float4*>(&patch[col * kPS + 8 * g + 4 * half]) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int f = 8 * it + (lane >> 3), c4 = 4 * (lane & 7);
    if (f < frames_valid) *reinterpret_cast<float4*>(&dst[(size_t)f * ld + c4]) = *reinterpret_cast<const float4*>(&patch[f * kPS + c4]);
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

  // ---- fir = D[128 .. 255] * H  (the upper 128 taps: the lower half is their mirror image, include/nws_hip.h) ----
  {
    const int c0 = 32 * wave;
    f32x16 acc;
    gemm_tile<kDK / 2>(fir_design + (size_t)NWS_FIR_HALF * kDK, kDK, c0, NWS_FIR_HALF, L.p1, lane, acc);
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    store_tile_frame_major(L.stage[wave], v, lane, fir_out + ((size_t)b * T + t0) * NWS_FIR_HALF + c0, NWS_FIR_HALF,
                           frames_valid);
  }
}

// =============================================================================================
// fp16 two-term-split variant (default): every layer on v_mfma_f32_32x32x16_f16.
//   W = W_hi + W_lo  pre-split once per weight version into A-fragment order (nws_mlp_frags, 800 KB),
//   X = X_hi + X_lo  kept in LDS TRANSPOSED (XT[frame][channel], fp16, row stride 304 B -> the B fragment
//                    of lane (frame j, half h) at K-step ks is ONE conflict-free ds_read_b128),
//   W X ~= W_hi X_hi + W_hi X_lo + W_lo X_hi  (22+ bits per product, fp32 accumulate).
// 24 MFMAs of 32 cycles per 32x128 tile instead of 64 fp32 MFMAs of 64 cycles.  Inputs of every layer are
// bounded by weight norms (|gru| < 1, LayerNorm outputs <= sqrt(127)|gamma|+|beta|); the host checks those
// bounds against the fp16 range and falls back to the exact-fp32 kernel above otherwise.
// =============================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kRowB = 304;  // bytes per frame row of an XT buffer (144 halfs + 16 B pad: 19 x 16 B, odd -> no conflicts)
constexpr int kXtBytes = kFT * kRowB;

// fragment-table geometry (units of f16x8); per (M-tile, K-step): 64 hi fragments then 64 lo fragments
struct FragMap {
  int base, mt, ks;
};
__host__ __device__ constexpr FragMap frag_map(int id) {
  // 0 proj | 1-3 newt hidden | 4 newt out (256) | 5-7 hgen hidden | 8 hgen out (129 -> 160) | 9 FIR design rows 128..255 (K 132 -> 144)
  return id == 0 ? FragMap{0, 4, 8}
       : id <= 3 ? FragMap{4096 * id, 4, 8}
       : id == 4 ? FragMap{16384, 8, 8}
       : id <= 7 ? FragMap{24576 + 4096 * (id - 5), 4, 8}
       : id == 8 ? FragMap{36864, 5, 8}
                 : FragMap{41984, 4, 9};   // rows 128..255 of D only (upper half-taps)
}
constexpr int kFragTotal = 46592;  // x 16 B = 745472 B
static_assert(kFragTotal * 16 + 712 * 1024 == NWS_MLP_FRAGS_BYTES, "fragment tables: tile kernels | wave-resident kernel");

struct MlpLds16 {
  char xt[4][2][kXtBytes];     // four activation buffers E, X, Y, Z, each [hi|lo]; dead ones double as store patches
  float red[2][2][4][kFT];     // [path][mean | M2][wave][frame]
  float gb[4][NWS_HIDDEN];     // LayerNorm gain / offset of the layer pair in flight: [g_a | beta_a | g_b | beta_b]
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// (hi, lo) fp16 split of two values: v_cvt_pk_f16_f32 for hi, one v_fma_mix{lo,hi}_f16 per lo (exact residual rounded once)
__device__ __forceinline__ void split2(float a, float b, f16x2& hi, f16x2& lo) {
  hi = __builtin_convertvector(f32x2{a, b}, f16x2);
  const unsigned hp = __builtin_bit_cast(unsigned, hi);
  unsigned lp;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(b));
  lo = __builtin_bit_cast(f16x2, lp);
}

__device__ __forceinline__ void split4_store(char* xt_hi, char* xt_lo, int frame, int ch, float a, float b, float c, float d) {
  f16x2 h01, l01, h23, l23;
  split2(a, b, h01, l01);
  split2(c, d, h23, l23);
  *reinterpret_cast<f16x4*>(xt_hi + frame * kRowB + ch * 2) = f16x4{h01.x, h01.y, h23.x, h23.y};
  *reinterpret_cast<f16x4*>(xt_lo + frame * kRowB + ch * 2) = f16x4{l01.x, l01.y, l23.x, l23.y};
}

// weight fragments of one M-tile in registers (KS K-steps x (hi, lo) x 8 halfs)
template <int KS>
struct AFrag {
  f16x8 hi[KS], lo[KS];
};

template <int KS>
__device__ __forceinline__ void load_frags(AFrag<KS>& A, const f16x8* __restrict__ frags, int mt, int lane) {
  const f16x8* a = frags + (size_t)mt * KS * 128 + lane;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    A.hi[ks] = a[ks * 128];
    A.lo[ks] = a[ks * 128 + 64];
  }
}

// acc(32 rows x 32 frames) = A * X  for the XT buffer `xt` ([hi|lo]):  W_hi X_hi + (W_hi X_lo + W_lo X_hi)
template <int KS>
__device__ __forceinline__ void mma_tile(const AFrag<KS>& A, const char* xt, int lane, f32x16& acc) {
  const int half = lane >> 5, col = lane & 31;
  f32x16 cross;
  const char* bh = xt + col * kRowB + half * 16;
  const char* bl = bh + kXtBytes;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f16x8 xh = *reinterpret_cast<const f16x8*>(bh + ks * 32);
    const f16x8 xl = *reinterpret_cast<const f16x8*>(bl + ks * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[ks], xh, ks == 0 ? f32x16{} : acc, 0, 0, 0);
    cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[ks], xl, ks == 0 ? f32x16{} : cross, 0, 0, 0);
    cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.lo[ks], xh, cross, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += cross[r];
}

// write a 32-channel x 32-frame tile held in the D layout into an XT buffer (4 consecutive channels per store)
template <int XB = kXtBytes>
__device__ __forceinline__ void store_tile_xt(char* xt, int c0, const float v[16], int lane) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    split4_store(xt, xt + XB, col, c0 + 8 * g + 4 * half, v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// the 16 per-lane entries (channels 32 wave + frag_row(r, half)) of a per-channel parameter vector
__device__ __forceinline__ void load_lane_params(float (&p)[16], const float* __restrict__ vec, int wave, int lane) {
  const int half = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) p[r] = vec[32 * wave + frag_row(r, half)];
}

// per-wave LayerNorm partials over its 32 channels of frame `col`: mean_w and M2_w = sum (v - mean_w)^2, both exact to
// fp32 rounding whatever the offset of the data (no E[x^2] - mean^2 cancellation)
__device__ __forceinline__ void ln_partials(const float (&v)[16], float& mean_w, float& m2_w) {
  float s = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += v[r];
  s += nws_swap_halves(s);
  mean_w = s * (1.0f / 32.0f);
  float q = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float d = v[r] - mean_w;
    q = fmaf(d, d, q);
  }
  m2_w = q + nws_swap_halves(q);
}

// the four waves' partials -> mean and 1/std of the 128 channels (Chan's merge: M2 = sum M2_w + 32 sum (mean_w - mean)^2)
__device__ __forceinline__ void ln_merge(const float (*red)[4][kFT], int col, float& mean, float& rstd) {
  const float m0 = red[0][0][col], m1 = red[0][1][col], m2 = red[0][2][col], m3 = red[0][3][col];
  mean = ((m0 + m1) + (m2 + m3)) * 0.25f;
  const float d0 = m0 - mean, d1 = m1 - mean, d2 = m2 - mean, d3 = m3 - mean;
  const float between = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
  const float within = (red[1][0][col] + red[1][1][col]) + (red[1][2][col] + red[1][3][col]);
  const float var = fmaf(32.0f, between, within) * (1.0f / NWS_HIDDEN);
  rstd = 1.0f / sqrtf(var + kLnEps);
}

// gain / offset of the lane's 16 channels from LDS (4 consecutive channels per ds_read_b128)
__device__ __forceinline__ void ln_finish(float (&v)[16], float mean, float rstd, const float* g, const float* bt, int wave,
                                          int lane) {
  const int half = lane >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = 32 * wave + 8 * q + 4 * half;
    const float4 g4 = *reinterpret_cast<const float4*>(g + c), b4 = *reinterpret_cast<const float4*>(bt + c);
    const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float y = (v[4 * q + i] - mean) * rstd * gg[i] + bb[i];
      v[4 * q + i] = fmaxf(y, 0.01f * y);  // LeakyReLU(0.01)
    }
  }
}

// one accumulator chain for all three products (dependent 32x32x16 MFMAs issue back to back at full rate; the small cross
// terms meet the same fp32 additions either way): 16 registers less than mma_tile, no merge pass
template <int KS, int XB = kXtBytes>
__device__ __forceinline__ void mma_tile1(const AFrag<KS>& A, const char* xt, int lane, f32x16& acc) {
  const int half = lane >> 5, col = lane & 31;
  const char* bh = xt + col * kRowB + half * 16;
  const char* bl = bh + XB;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f16x8 xh = *reinterpret_cast<const f16x8*>(bh + ks * 32);
    const f16x8 xl = *reinterpret_cast<const f16x8*>(bl + ks * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[ks], xl, ks == 0 ? f32x16{} : acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.lo[ks], xh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[ks], xh, acc, 0, 0, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// newt.mlp and h_generator are independent after the embedding.  8 waves: wave = (M-tile mt, path), path 0 = newt.mlp,
// path 1 = h_generator, ONE 32 x 32 tile per wave and layer; the LayerNorm statistics of both paths travel in one exchange,
// and each wave requests the weight fragments of its next layer as soon as the current layer's MFMAs have been issued
// (their registers are free from then on), so they arrive under the epilogue.  80 KB of LDS (two workgroups per CU) and
// <= 128 registers: four waves per SIMD.  Buffers: E (embedding, later hgen L2), X, Y, Z.
//   proj: waves 0-3: X(gru) -> E        L1: E -> X | E -> Y        L2: X -> Z | Y -> E        L3: Z -> X | E -> Y
//   out : waves 0-3: X -> film tiles mt, mt+4;   waves 4-7: Y -> H tile mt -> Z (wave 4 also row 128)
//   fir : all 8 waves: Z -> fir tile `wave`
// 10 workgroup barriers per 32 frames.  Measured (B=64, T=500, one stream): 0.065 ms; the round-2 form with four waves
// and two tiles per wave (252 registers, two waves per SIMD) 0.066-0.067 ms; with the W_lo fragment loads skipped
// (wrong results, timing only) 0.055 ms: ~30 % of the time is the 819 KB of fragments every workgroup streams from L2
// (13 TB/s aggregate), the rest latency between the ten phases.
template <int FA_NEXT, int FB_NEXT>
__device__ __forceinline__ void hidden_w8(MlpLds16& L, const f16x8* __restrict__ F, AFrag<8>& A, const char* in, char* out,
                                          const float* bias, const float* g_a, const float* bt_a, const float* g_b,
                                          const float* bt_b, int mt, int path, int tid, int lane) {
  const int half = lane >> 5, col = lane & 31;
  float v[16];
  load_lane_params(v, bias, mt, lane);
  f32x16 acc;
  mma_tile1<8>(A, in, lane, acc);
  // thread t fetches gb[t >> 7][t & 127]: [g_a | beta_a | g_b | beta_b]
  const int row = tid >> 7;
  const float p0 = (row == 0 ? g_a : row == 1 ? bt_a : row == 2 ? g_b : bt_b)[tid & 127];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] += acc[r];
  __builtin_amdgcn_sched_barrier(0);   // the fragments reuse registers the MFMAs above have just released
  load_frags<8>(A, F + (path ? frag_map(FB_NEXT).base : frag_map(FA_NEXT).base), mt, lane);
  L.gb[row][tid & 127] = p0;
  float mw, m2;
  ln_partials(v, mw, m2);
  if (half == 0) {
    L.red[path][0][mt][col] = mw;
    L.red[path][1][mt][col] = m2;
  }
  __syncthreads();
  float mean, rstd;
  ln_merge(L.red[path], col, mean, rstd);
  ln_finish(v, mean, rstd, L.gb[2 * path], L.gb[2 * path + 1], mt, lane);
  store_tile_xt(out, 32 * mt, v, lane);
  __syncthreads();
}

// TAPS: also store the embedding and H (stage tests, get_embedding); the forward's instantiation carries neither the code nor
// its address registers
template <bool TAPS>
__global__ __launch_bounds__(512, 4) void frame_mlps16_kernel(NwsWeights w, const float* __restrict__ gru_out, int T,
                                                                float* __restrict__ emb_out, float* __restrict__ film_out,
                                                                float* __restrict__ H_out, float* __restrict__ fir_out,
                                                                int out_T, int out_off, NwsStreamNoiseWin win) {
  // out_T / out_off: the FiLM and FIR-tap rows of frame t go to row out_off + t of windows of out_T rows per utterance (T, 0: the
  // plain (B, T, .) outputs).  Streaming hop (nws_frame_mlps_stream): one more workgroup behind the B utterances moves the shared
  // noise window on and applies the previous hop's pending counters (nws_common.h)
  if (win.nzwin != nullptr && blockIdx.y == gridDim.y - 1) {
    nws_stream_noise_window_block<512>(win, threadIdx.x);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpLds16& L = *reinterpret_cast<MlpLds16*>(smem_raw);
  char* const E = L.xt[0][0];
  char* const X = L.xt[1][0];
  char* const Y = L.xt[2][0];
  char* const Z = L.xt[3][0];
  const f16x8* F = reinterpret_cast<const f16x8*>(w.mlp_frags);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the branches on mt / path below are uniform
  const int mt = wave & 3, path = wave >> 2;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kFT;
  const int frames_valid = T - t0 < kFT ? T - t0 : kFT;
  AFrag<8> A;
  f32x16 acc;

  __builtin_amdgcn_sched_barrier(0);   // the fragments reuse registers the MFMAs above have just released
  load_frags<8>(A, F + (path ? frag_map(5).base : frag_map(0).base), mt, lane);  // proj | hgen hidden 0
  // ---- gru_out tile -> X (4 channels per thread per pass) ----
  for (int e = tid; e < kFT * (NWS_HIDDEN / 4); e += 512) {
    const int f = e >> 5, c4 = (e & 31) * 4;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (f < frames_valid) v = *reinterpret_cast<const float4*>(&gru_out[((size_t)b * T + t0 + f) * NWS_HIDDEN + c4]);
    split4_store(X, X + kXtBytes, f, c4, v.x, v.y, v.z, v.w);
  }
  // zero the K padding (channels 128..143) of Z, which will hold H for the FIR-design contraction
  if (tid < kFT * 2 * 2) {
    const int f = tid >> 2, part = tid & 3;
    *reinterpret_cast<float4*>(Z + (part >> 1) * kXtBytes + f * kRowB + 256 + (part & 1) * 16) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  __syncthreads();

  // ---- emb = proj(gru_out) -> E (waves 0-3; the other four already hold their first hidden layer's fragments) ----
  if (path == 0) {
    float v[16];
    load_lane_params(v, w.proj_b, mt, lane);
    mma_tile1<8>(A, X, lane, acc);
    __builtin_amdgcn_sched_barrier(0);   // the fragments reuse registers the MFMAs above have just released
  load_frags<8>(A, F + frag_map(1).base, mt, lane);  // newt hidden 0
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += acc[r];
    store_tile_xt(E, 32 * mt, v, lane);
    if (TAPS && emb_out != nullptr && col < frames_valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) emb_out[((size_t)b * NWS_HIDDEN + 32 * mt + frag_row(r, half)) * T + t0 + col] = v[r];
    }
  }
  __syncthreads();

  // ---- three hidden layers per path (each requests the fragments of the one that follows) ----
  hidden_w8<2, 6>(L, F, A, E, path ? Y : X, path ? w.hgen_b[0] : w.newt_mlp_b[0], w.newt_ln_g[0], w.newt_ln_b[0],
                  w.hgen_ln_g[0], w.hgen_ln_b[0], mt, path, tid, lane);
  hidden_w8<3, 7>(L, F, A, path ? Y : X, path ? E : Z, path ? w.hgen_b[1] : w.newt_mlp_b[1], w.newt_ln_g[1], w.newt_ln_b[1],
                  w.hgen_ln_g[1], w.hgen_ln_b[1], mt, path, tid, lane);
  hidden_w8<4, 8>(L, F, A, path ? E : Z, path ? Y : X, path ? w.hgen_b[2] : w.newt_mlp_b[2], w.newt_ln_g[2], w.newt_ln_b[2],
                  w.hgen_ln_g[2], w.hgen_ln_b[2], mt, path, tid, lane);

  // ---- output layers.  E is dead: transposition patches of waves 0-3 ----
  float* patch = reinterpret_cast<float*>(E) + mt * (kFT * kPS);
  if (path == 0) {
    // film (256 channels): M-tiles mt and mt + 4 from X
    float v[16];
    load_lane_params(v, w.newt_mlp_b[3], mt, lane);
    mma_tile1<8>(A, X, lane, acc);
    __builtin_amdgcn_sched_barrier(0);   // the fragments reuse registers the MFMAs above have just released
  load_frags<8>(A, F + frag_map(4).base, mt + 4, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += acc[r];
    store_tile_frame_major(patch, v, lane, film_out + ((size_t)b * out_T + out_off + t0) * NWS_FILM_CH + 32 * mt, NWS_FILM_CH, frames_valid);
    load_lane_params(v, w.newt_mlp_b[3], mt + 4, lane);
    mma_tile1<8>(A, X, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += acc[r];
    store_tile_frame_major(patch, v, lane, film_out + ((size_t)b * out_T + out_off + t0) * NWS_FILM_CH + 32 * (mt + 4), NWS_FILM_CH,
                           frames_valid);
  } else {
    // H (129 bands): M-tiles 0..3 from Y -> Z channels 0..127; wave 4 also M-tile 4 = row 128 (129..143 stay zero)
    float vh[16];
    load_lane_params(vh, w.hgen_b[3], mt, lane);
    const float b128 = w.hgen_b[3][128];
    mma_tile1<8>(A, Y, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) vh[r] += acc[r];
    store_tile_xt(Z, 32 * mt, vh, lane);
    if (TAPS && H_out != nullptr && col < frames_valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) H_out[((size_t)b * T + t0 + col) * NWS_N_BANDS + 32 * mt + frag_row(r, half)] = vh[r];
    }
    if (mt == 0) {
      // (its own fragment registers, live inside this block only: a conditional reload of A would keep the old contents alive)
      AFrag<8> A4;
      __builtin_amdgcn_sched_barrier(0);   // not above the MFMAs that still read A
      load_frags<8>(A4, F + frag_map(8).base, 4, lane);
      mma_tile1<8>(A4, Y, lane, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 128 + frag_row(r, half);
        vh[r] = c < NWS_N_BANDS ? acc[r] + b128 : 0.0f;   // only row 128 is real
        if (TAPS && H_out != nullptr && c < NWS_N_BANDS && col < frames_valid) H_out[((size_t)b * T + t0 + col) * NWS_N_BANDS + c] = vh[r];
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)  // rows 128..143 only: the XT row holds 144 channels
        split4_store(Z, Z + kXtBytes, col, 128 + 8 * g + 4 * half, vh[4 * g], vh[4 * g + 1], vh[4 * g + 2], vh[4 * g + 3]);
    }
  }
  // ---- fir = D[128 .. 255] * H  (upper half-taps = 4 M-tiles, K = 144 padded): waves 0-3; their fragments are requested
  // before the barrier (nothing else to wait for).  Waves 4-7 have produced H and are done. ----
  AFrag<9> A9;
  if (path == 0) {
    __builtin_amdgcn_sched_barrier(0);   // the fragments reuse registers the MFMAs above have just released
    load_frags<9>(A9, F + frag_map(9).base, mt, lane);
  }
  __syncthreads();
  if (path == 0) {
    float v[16];
    mma_tile1<9>(A9, Z, lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    store_tile_frame_major(patch, v, lane, fir_out + ((size_t)b * out_T + out_off + t0) * NWS_FIR_HALF + 32 * mt, NWS_FIR_HALF, frames_valid);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 64 frames per workgroup (round 3): the same wave layout (wave = (M-tile, path)), but every wave runs its
// layer on TWO 32-frame N-tiles with ONE set of weight fragments: half the fragment bytes per frame (each workgroup streams
// 745 KB from L2 at the CU's 64 B/clk whatever its tile: 35 % of the 32-frame kernel's time), two independent MFMA chains per
// wave, half the barrier phases per frame.  155 KB of LDS: one workgroup (8 waves) per CU, <= 256 registers.
constexpr int kFT2 = 64;
constexpr int kXtBytes2 = kFT2 * kRowB;
struct MlpLds64 {
  char xt[4][2][kXtBytes2];    // E, X, Y, Z, each [hi|lo]
  float red[2][2][4][kFT2];    // [path][mean | M2][wave][frame]
  float gb[4][NWS_HIDDEN];
};
static_assert(sizeof(MlpLds64) <= 160 * 1024, "LDS");
static_assert(8 * kFT * kPS * 4 <= 2 * kXtBytes2, "eight per-wave store patches inside the dead E buffer");

template <bool TAPS>
__global__ __launch_bounds__(512, 2) void frame_mlps64_kernel(NwsWeights w, const float* __restrict__ gru_out, int T,
                                                                float* __restrict__ emb_out, float* __restrict__ film_out,
                                                                float* __restrict__ H_out, float* __restrict__ fir_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpLds64& L = *reinterpret_cast<MlpLds64*>(smem_raw);
  constexpr int XB = kXtBytes2;
  constexpr int kTileB = kFT * kRowB;          // byte offset of the second N-tile (frames 32..63) inside an XT buffer
  char* const E = L.xt[0][0];
  char* const X = L.xt[1][0];
  char* const Y = L.xt[2][0];
  char* const Z = L.xt[3][0];
  const f16x8* F = reinterpret_cast<const f16x8*>(w.mlp_frags);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = wave & 3, path = wave >> 2;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kFT2;
  const int frames_valid = T - t0 < kFT2 ? T - t0 : kFT2;
  AFrag<8> A;
  f32x16 acc0, acc1;

  __builtin_amdgcn_sched_barrier(0);
  load_frags<8>(A, F + frag_map(0).base, mt, lane);  // proj
  for (int e = tid; e < kFT2 * (NWS_HIDDEN / 4); e += 512) {
    const int f = e >> 5, c4 = (e & 31) * 4;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (f < frames_valid) v = *reinterpret_cast<const float4*>(&gru_out[((size_t)b * T + t0 + f) * NWS_HIDDEN + c4]);
    split4_store(X, X + XB, f, c4, v.x, v.y, v.z, v.w);
  }
  if (tid < kFT2 * 2 * 2) {     // K padding (channels 128..143) of Z
    const int f = tid >> 2, part = tid & 3;
    *reinterpret_cast<float4*>(Z + (part >> 1) * XB + f * kRowB + 256 + (part & 1) * 16) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  __syncthreads();

  // ---- emb = proj(gru_out) -> E: the layer has 4 M-tiles x 2 N-tiles = one tile per wave (wave = (mt, N-tile `path`)) ----
  {
    float vb[16];
    load_lane_params(vb, w.proj_b, mt, lane);
    mma_tile1<8, XB>(A, X + path * kTileB, lane, acc0);
    __builtin_amdgcn_sched_barrier(0);
    load_frags<8>(A, F + (path ? frag_map(5).base : frag_map(1).base), mt, lane);  // hgen hidden 0 | newt hidden 0
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = vb[r] + acc0[r];
    store_tile_xt<XB>(E + path * kTileB, 32 * mt, v, lane);
    if (TAPS && emb_out != nullptr && 32 * path + col < frames_valid) {
#pragma unroll
      for (int r = 0; r < 16; ++r) emb_out[((size_t)b * NWS_HIDDEN + 32 * mt + frag_row(r, half)) * T + t0 + 32 * path + col] = v[r];
    }
  }
  __syncthreads();

  // ---- hidden layers: Xout = LeakyReLU(LayerNorm(W Xin + b)) on both N-tiles; the next layer's fragments are requested as
  // soon as both tiles' MFMAs have been issued ----
  auto hidden = [&](const int fa_next, const int fb_next, const char* in, char* out, const float* bias, const float* g_a,
                    const float* bt_a, const float* g_b, const float* bt_b) {
    float v0[16], v1[16];
    load_lane_params(v0, bias, mt, lane);
    mma_tile1<8, XB>(A, in, lane, acc0);
    mma_tile1<8, XB>(A, in + kTileB, lane, acc1);
    const int row = tid >> 7;
    const float p0 = (row == 0 ? g_a : row == 1 ? bt_a : row == 2 ? g_b : bt_b)[tid & 127];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      v1[r] = v0[r] + acc1[r];
      v0[r] += acc0[r];
    }
    __builtin_amdgcn_sched_barrier(0);
    load_frags<8>(A, F + frag_map(path ? fb_next : fa_next).base, mt, lane);
    L.gb[row][tid & 127] = p0;
    float mw, m2;
    ln_partials(v0, mw, m2);
    if (half == 0) {
      L.red[path][0][mt][col] = mw;
      L.red[path][1][mt][col] = m2;
    }
    ln_partials(v1, mw, m2);
    if (half == 0) {
      L.red[path][0][mt][32 + col] = mw;
      L.red[path][1][mt][32 + col] = m2;
    }
    __syncthreads();
    auto merge = [&](int f, float& mean, float& rstd) {
      const float (*red)[4][kFT2] = L.red[path];
      const float m0 = red[0][0][f], m1 = red[0][1][f], m2_ = red[0][2][f], m3 = red[0][3][f];
      mean = ((m0 + m1) + (m2_ + m3)) * 0.25f;
      const float d0 = m0 - mean, d1 = m1 - mean, d2 = m2_ - mean, d3 = m3 - mean;
      const float between = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
      const float within = (red[1][0][f] + red[1][1][f]) + (red[1][2][f] + red[1][3][f]);
      rstd = 1.0f / sqrtf(fmaf(32.0f, between, within) * (1.0f / NWS_HIDDEN) + kLnEps);
    };
    float mean, rstd;
    merge(col, mean, rstd);
    ln_finish(v0, mean, rstd, L.gb[2 * path], L.gb[2 * path + 1], mt, lane);
    store_tile_xt<XB>(out, 32 * mt, v0, lane);
    merge(32 + col, mean, rstd);
    ln_finish(v1, mean, rstd, L.gb[2 * path], L.gb[2 * path + 1], mt, lane);
    store_tile_xt<XB>(out + kTileB, 32 * mt, v1, lane);
    __syncthreads();
  };
  hidden(2, 6, E, path ? Y : X, path ? w.hgen_b[0] : w.newt_mlp_b[0], w.newt_ln_g[0], w.newt_ln_b[0], w.hgen_ln_g[0], w.hgen_ln_b[0]);
  hidden(3, 7, path ? Y : X, path ? E : Z, path ? w.hgen_b[1] : w.newt_mlp_b[1], w.newt_ln_g[1], w.newt_ln_b[1], w.hgen_ln_g[1],
         w.hgen_ln_b[1]);
  hidden(4, 8, path ? E : Z, path ? Y : X, path ? w.hgen_b[2] : w.newt_mlp_b[2], w.newt_ln_g[2], w.newt_ln_b[2], w.hgen_ln_g[2],
         w.hgen_ln_b[2]);

  // ---- output layers.  E is dead: transposition patches of waves 0-3 ----
  float* patch = reinterpret_cast<float*>(E) + mt * (kFT * kPS);
  auto fv = [&](int nt) { return frames_valid - 32 * nt < 0 ? 0 : (frames_valid - 32 * nt > 32 ? 32 : frames_valid - 32 * nt); };
  if (path == 0) {
    float vb[16], v[16];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {     // film M-tiles mt and mt + 4
      const int tile = mt + 4 * pass;
      load_lane_params(vb, w.newt_mlp_b[3], tile, lane);
      mma_tile1<8, XB>(A, X, lane, acc0);
      mma_tile1<8, XB>(A, X + kTileB, lane, acc1);
      if (pass == 0) {
        __builtin_amdgcn_sched_barrier(0);
        load_frags<8>(A, F + frag_map(4).base, mt + 4, lane);
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = vb[r] + (nt ? acc1[r] : acc0[r]);
        store_tile_frame_major(patch, v, lane, film_out + ((size_t)b * T + t0 + 32 * nt) * NWS_FILM_CH + 32 * tile, NWS_FILM_CH, fv(nt));
      }
    }
  } else {
    // H (129 bands): M-tiles 0..3 from Y -> Z channels 0..127; wave 4 also M-tile 4 = row 128 (129..143 stay zero)
    float vb[16], vh[16];
    load_lane_params(vb, w.hgen_b[3], mt, lane);
    const float b128 = w.hgen_b[3][128];
    mma_tile1<8, XB>(A, Y, lane, acc0);
    mma_tile1<8, XB>(A, Y + kTileB, lane, acc1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) vh[r] = vb[r] + (nt ? acc1[r] : acc0[r]);
      store_tile_xt<XB>(Z + nt * kTileB, 32 * mt, vh, lane);
      if (TAPS && H_out != nullptr && 32 * nt + col < frames_valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) H_out[((size_t)b * T + t0 + 32 * nt + col) * NWS_N_BANDS + 32 * mt + frag_row(r, half)] = vh[r];
      }
    }
    if (mt == 0) {
      AFrag<8> A4;
      __builtin_amdgcn_sched_barrier(0);
      load_frags<8>(A4, F + frag_map(8).base, 4, lane);
      mma_tile1<8, XB>(A4, Y, lane, acc0);
      mma_tile1<8, XB>(A4, Y + kTileB, lane, acc1);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = 128 + frag_row(r, half);
          vh[r] = c < NWS_N_BANDS ? (nt ? acc1[r] : acc0[r]) + b128 : 0.0f;   // only row 128 is real
          if (TAPS && H_out != nullptr && c < NWS_N_BANDS && 32 * nt + col < frames_valid)
            H_out[((size_t)b * T + t0 + 32 * nt + col) * NWS_N_BANDS + c] = vh[r];
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
          split4_store(Z + nt * kTileB, Z + nt * kTileB + XB, col, 128 + 8 * g + 4 * half, vh[4 * g], vh[4 * g + 1], vh[4 * g + 2], vh[4 * g + 3]);
      }
    }
  }
  // ---- fir = D[128 .. 255] * H  (upper half-taps = 4 M-tiles, K = 144 padded) x 2 N-tiles: one tile per wave ----
  AFrag<9> A9;
  __builtin_amdgcn_sched_barrier(0);
  load_frags<9>(A9, F + frag_map(9).base, mt, lane);
  __syncthreads();
  {
    float v[16];
    mma_tile1<9, XB>(A9, Z + path * kTileB, lane, acc0);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc0[r];
    float* patch8 = reinterpret_cast<float*>(E) + wave * (kFT * kPS);     // E is dead for every wave by now
    store_tile_frame_major(patch8, v, lane, fir_out + ((size_t)b * T + t0 + 32 * path) * NWS_FIR_HALF + 32 * mt, NWS_FIR_HALF, fv(path));
  }
}

// =====================================================================================================================
// Wave-resident frames (round 4).  The kernels above give every wave one M-tile of a layer and pass the
// activations from wave to wave through LDS: ten workgroup-wide phases per tile, each an MFMA burst, a cross-wave LayerNorm
// exchange, a split and a barrier (MFMA busy 23 %, 77 % of the wave cycles waiting).  Here a wave OWNS 32 frames and runs the
// whole chain on them in registers:
//   * the D layout of a 32x32x16 MFMA gives lane (frame j, half h) the channels 32 mt + 8 q + 4 h + i of its frame, and the B
//     operand of the next layer wants lane (j, h) to hold eight k indices of the same frame - the contraction order is free,
//     so the weight fragments are stored in the order the accumulators come out (kperm_d below) and a layer's output becomes
//     the next layer's operand with no LDS round trip and no cross-lane traffic;
//   * LayerNorm statistics are in-lane sums over 64 values plus ONE half swap - no cross-wave exchange, no barrier;
//   * the weights are what is shared: every layer's pre-split fragments (64-72 KB) are copied into LDS by the DMA path
//     (global_load_lds_dwordx4) one layer ahead, double buffered, and read by all eight waves as A operands (ds_read_b128,
//     lane-linear, conflict-free) - 8 x 32 = 256 frames per 712 KB of fragment traffic instead of 64;
//   * newt.mlp and h_generator are separate workgroups (blockIdx.y; both compute proj: 9 % more MFMAs) so that one layer of
//     one path fits a 72 KB slot; B x T = 32000 frames -> 125 + 125 workgroups, one round on 256 CUs, two waves per SIMD;
//   * the output layers (film, fir) run TRANSPOSED (activations as the A operand, weights as B: the same register contents),
//     so the accumulators come out as lane = channel, register = frame and every store instruction writes two full 128 B
//     segments of frame-major rows straight from registers - no transposition patches;
//   * band 128 of H (the 129th row of h_generator's last layer) is a 128-term fp32 dot product on the vector pipe.
// Frames are the flattened (b, t) index: every tensor here is frame-major and contiguous, so tiles need not respect
// utterance boundaries.  6 workgroup barriers per 256 frames (they only guard the reuse of a weight slot).
// Measured on MI355X at B x T = 64 x 500 (rocprofv3, one stream, inside whole forwards): 41.5 us against 57.7 us for
// frame_mlps64_kernel; tools/mlp_variants.py, tools/mlp_timeline.py.  What the time is: 1164 MFMAs per SIMD (37 K cycles of the
// matrix pipe) and ~2 x 2700 vector instructions that do not overlap them (no MFMAs: 15 us; no LayerNorm arithmetic -3.6, no split
// -5.8); s_setprio around the MFMA phases -5 %.  Measured and dropped: waves 4..7 half a layer out of phase with waves 0..3 (a SIMD
// would always have one wave in its MFMA phase and one in its vector phase): same time; the three products of a K-step issued
// across the M-tiles instead of per accumulator: same time.
// =====================================================================================================================
constexpr int kWrFrames = 256;
constexpr int kWrSlot = 72 * 1024;
constexpr int kWrPar = 1600;
struct WrLds {
  char slot[2][kWrSlot];
  float par[kWrPar];   // proj_b | 3 x (bias, ln gain, ln offset) | out-layer bias (256) resp. bias (128), w128 (128), b128
};
static_assert(sizeof(WrLds) <= 160 * 1024, "LDS");

// second fragment table (behind the first): chunk offsets in KB; [M-tile][K-step][hi | lo][lane] x 16 B per chunk
constexpr int kT2Base = kFragTotal * 16;
constexpr int kT2Kb[2][6] = {{0, 64, 128, 192, 256, 320}, {0, 384, 448, 512, 576, 640}};
constexpr int kT2Bytes = 712 * 1024;
static_assert(kT2Base + kT2Bytes == NWS_MLP_FRAGS_BYTES, "fragment table size");

// contraction order of a layer whose input is (a) the GRU output read from memory: lane half h holds channels 64 h .. 64 h + 63
// of its frame; (b) a previous layer's accumulators: K-step (mt, g) holds registers 8 g .. 8 g + 7 of M-tile mt
__host__ __device__ constexpr int kperm_in(int ks, int h, int i) { return 64 * h + 8 * ks + i; }
__host__ __device__ constexpr int kperm_d(int ks, int h, int i) { return 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h + i + (i >= 4 ? 4 : 0); }

struct WrAct {
  f16x8 hi[9], lo[9];   // K-steps 0..7: 128 channels of this lane's frame; 8: band 128 of H (FIR design only)
};

// (hi, lo) split in plain arithmetic: the results feed MFMAs straight from registers, and hipcc does not see through inline
// asm when it places the wait states a VALU-write -> MFMA-read pair needs (split2's v_fma_mix as inline asm gave wrong first
// products of a layer here; the tile kernels pass their splits through LDS)
// hi = the top 11 significant bits (a mask: exactly representable in fp16 wherever fp16 is normal), lo = the exact remainder
// rounded to fp16: 22 bits per value like round-then-subtract, with two v_and instead of two conversions back to fp32.
// (Below 2^-14 the fp16 conversion of hi rounds by <= 2^-25 absolute that lo does not see: nothing against O(1) activations.)
__device__ __forceinline__ void wr_split2(float a, float b, f16x2& hi, f16x2& lo) {
  const float ha = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFFE000u);
  const float hb = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xFFFFE000u);
  hi = __builtin_convertvector(f32x2{ha, hb}, f16x2);
  lo = __builtin_convertvector(f32x2{a - ha, b - hb}, f16x2);
}
__device__ __forceinline__ void wr_pack8(const float (&y)[8], f16x8& hi, f16x8& lo) {
  f16x2 h0, l0, h1, l1, h2, l2, h3, l3;
  wr_split2(y[0], y[1], h0, l0);
  wr_split2(y[2], y[3], h1, l1);
  wr_split2(y[4], y[5], h2, l2);
  wr_split2(y[6], y[7], h3, l3);
  hi = f16x8{h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x, h3.y};
  lo = f16x8{l0.x, l0.y, l1.x, l1.y, l2.x, l2.y, l3.x, l3.y};
}

// this wave's share (1/8) of a chunk: global -> LDS by the DMA path, 1 KB per instruction
__device__ __forceinline__ void wr_dma(const char* __restrict__ src, char* dst, int kb, int wave, int lane) {
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int per = kb * 128;   // bytes per wave: 8192 or 9216
  const char* s = src + wave * per + lane * 16;
  char* d = dst + wave * per;
#pragma unroll
  for (int j = 0; j < 8; ++j) __builtin_amdgcn_global_load_lds((gptr_t)(s + j * 1024), (lptr_t)(d + j * 1024), 16, 0, 0);
  if (kb == 72) __builtin_amdgcn_global_load_lds((gptr_t)(s + 8192), (lptr_t)(d + 8192), 16, 0, 0);
}

// NMT M-tiles of one layer against the lane's frame: acc[mt] += W[mt] X (standard orientation: weights as A, activations as B;
// accumulators = channels x frames) or, TRANSPOSED, acc[mt] += X^T W[mt]^T (activations as A, weights as B: frames x channels).
// The weight fragments come from LDS (lane-linear 1 KB pieces, [mt][ks][hi | lo]).  Steps run K-major over the M-tiles (NMT
// independent accumulator chains) and the fragments of step s + 2 are requested before the three MFMAs of step s are issued:
// left to itself hipcc requests each pair right in front of its first use and the matrix pipe idles for an LDS latency per
// three MFMAs (measured: 34 us of a 49 us kernel in the MFMA phases against 16 us of MFMA time).
#ifndef NWS_WR_PRIO
#define NWS_WR_PRIO 1
#endif

template <int KS, int NMT, bool TRANSPOSED, bool ZERO = false, bool NOLDS = false, bool PRIO = NWS_WR_PRIO>
__device__ __forceinline__ void wr_layer_mma(const char* slot, int mt0, const WrAct& x, f32x16* acc, int lane) {
  const char* a = slot + (size_t)mt0 * KS * 2048 + lane * 16;
  constexpr int S = KS * NMT;
  f16x8 wh[3], wl[3];
  auto ld = [&](int s2, int b) {
    if (NOLDS && s2 > 2) return;    // timing ablation: the first three steps' fragments serve every step
    const int mt = s2 % NMT, ks = s2 / NMT;
    wh[b] = *reinterpret_cast<const f16x8*>(a + (mt * KS + ks) * 2048);
    wl[b] = *reinterpret_cast<const f16x8*>(a + (mt * KS + ks) * 2048 + 1024);
  };
  ld(0, 0);
  if (S > 1) ld(1, 1);
  __builtin_amdgcn_sched_barrier(0x6);
  if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
  for (int s2 = 0; s2 < S; ++s2) {
    const int mt = s2 % NMT, ks = s2 / NMT, b = s2 % 3;
    if (s2 + 2 < S) ld(s2 + 2, (s2 + 2) % 3);
    __builtin_amdgcn_sched_barrier(0x6);
    if (!TRANSPOSED) {
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b], x.lo[ks], (ZERO && ks == 0) ? f32x16{} : acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[b], x.hi[ks], acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[b], x.hi[ks], acc[mt], 0, 0, 0);
    } else {
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.lo[ks], wh[b], (ZERO && ks == 0) ? f32x16{} : acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.hi[ks], wl[b], acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x.hi[ks], wh[b], acc[mt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0x6);   // vector / scalar ALU work may move across (the previous layer's split is interleaved
                                           // here by the compiler), LDS reads and MFMAs stay in this order
  }
  if (PRIO) __builtin_amdgcn_s_setprio(0);
}

// the lane's 64 entries of a per-channel vector (channels 32 mt + 8 q + 4 h + i) as four accumulator-shaped tiles
__device__ __forceinline__ void wr_lane_vec(const float* par, int half, f32x16 (&v)[4]) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 t = *reinterpret_cast<const float4*>(par + 32 * mt + 8 * q + 4 * half);
      v[mt][4 * q + 0] = t.x;
      v[mt][4 * q + 1] = t.y;
      v[mt][4 * q + 2] = t.z;
      v[mt][4 * q + 3] = t.w;
    }
}

// four accumulator tiles (128 channels of the lane's frame) -> operand of the next layer
template <bool CHEAP = false>
__device__ __forceinline__ void wr_pack(const f32x16 (&v)[4], WrAct& out) {
  if (CHEAP) {   // timing ablation: no split arithmetic
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      out.hi[ks] = __builtin_bit_cast(f16x8, f32x4{v[ks >> 1][8 * (ks & 1)], v[ks >> 1][8 * (ks & 1) + 1], v[ks >> 1][8 * (ks & 1) + 2], v[ks >> 1][8 * (ks & 1) + 3]});
      out.lo[ks] = __builtin_bit_cast(f16x8, f32x4{v[ks >> 1][8 * (ks & 1) + 4], v[ks >> 1][8 * (ks & 1) + 5], v[ks >> 1][8 * (ks & 1) + 6], v[ks >> 1][8 * (ks & 1) + 7]});
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float y[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = v[mt][8 * g + i];
      wr_pack8(y, out.hi[2 * mt + g], out.lo[2 * mt + g]);
    }
}

// LeakyReLU(LayerNorm(v)) over the 128 channels of the lane's frame (v includes the bias), in place
__device__ __forceinline__ void wr_layer_norm(f32x16 (&v)[4], const float* g, const float* bt, int half) {
  float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s0 += v[0][r];
    s1 += v[1][r];
    s2 += v[2][r];
    s3 += v[3][r];
  }
  float s = (s0 + s1) + (s2 + s3);
  s += nws_swap_halves(s);
  const float mean = s * (1.0f / NWS_HIDDEN);
  float q0 = 0.0f, q1 = 0.0f, q2 = 0.0f, q3 = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    v[0][r] -= mean;
    v[1][r] -= mean;
    v[2][r] -= mean;
    v[3][r] -= mean;
    q0 = fmaf(v[0][r], v[0][r], q0);
    q1 = fmaf(v[1][r], v[1][r], q1);
    q2 = fmaf(v[2][r], v[2][r], q2);
    q3 = fmaf(v[3][r], v[3][r], q3);
  }
  float q = (q0 + q1) + (q2 + q3);
  q += nws_swap_halves(q);
  const float rstd = 1.0f / sqrtf(q * (1.0f / NWS_HIDDEN) + kLnEps);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const int c = 32 * mt + 8 * qd + 4 * half;
      const float4 g4 = *reinterpret_cast<const float4*>(g + c), b4 = *reinterpret_cast<const float4*>(bt + c);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y = fmaf(v[mt][4 * qd + i] * rstd, gg[i], bb[i]);
        v[mt][4 * qd + i] = fmaxf(y, 0.01f * y);
      }
    }
}

// a transposed accumulator tile (lane = channel, register r = frame f0 + frag_row(r, half)) -> frame-major rows of LD floats: two
// full 128 B segments per store instruction.  Buffer stores: ONE per-lane byte offset for every tile of the wave (the tile's
// channel offset and the row r & 3 ride in the instruction's immediate, the row group r >> 2 in a scalar offset) and the
// descriptor's size drops the rows of frames >= F - no address arithmetic and no guards on the vector pipe.
template <int LD, bool BIAS>
__device__ __forceinline__ void wr_store_rows(__amdgpu_buffer_rsrc_t out, const f32x16& acc, int lane_off, int tile, float bias) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, BIAS ? acc[r] + bias : acc[r]), out,
                                          lane_off + ((r & 3) * LD + 32 * tile) * 4, 8 * (r >> 2) * LD * 4, 0);
}

// ABL: timing ablations (results meaningless): 1 no LayerNorm / LeakyReLU arithmetic, 2 no MFMAs (and no weight reads), 3 MFMAs without their weight reads from LDS, 5 no
// (hi, lo) split arithmetic; 6: product arithmetic + cycle timeline (s_memtime probes, tools/mlp_timeline.py)
template <bool TAPS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void frame_mlps_wr_kernel(NwsWeights w, const float* __restrict__ gru_out, int F, int T,
                                                                 float* __restrict__ emb_out, float* __restrict__ film_out,
                                                                 float* __restrict__ H_out, float* __restrict__ fir_out,
                                                                 const int xcd_blocks = 0, const int only_path = -1) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  WrLds& L = *reinterpret_cast<WrLds*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Both paths of a frame block read the same 256 GRU rows.  As grid (blocks, 2) the two workgroups of a block sat on different
  // XCDs (linear id % 8, MI355X_MICROARCH) and the rows came from HBM twice (88 MB per launch against 66 MB algorithmic).
  // xcd_blocks > 0: a 1-D grid of 16-workgroup groups - ids 16 g + x and 16 g + 8 + x are block 8 g + x on path 0 / 1, i.e. the
  // same XCD, eight launch slots apart; ids past the last block leave at once.  Placement only: same results.
  const int fb = xcd_blocks > 0 ? (int)(blockIdx.x >> 4) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;
  const int path = xcd_blocks > 0 ? (int)((blockIdx.x >> 3) & 1) : (int)blockIdx.y;   // 0: proj + newt.mlp -> film; 1: proj + h_generator -> H -> fir
  if (xcd_blocks > 0 && fb >= xcd_blocks) return;
  if (only_path >= 0 && path != only_path) return;   // measurements (tools/mlp_paths_ab.py): one path's workgroups alone
  const int half = lane >> 5, col = lane & 31;
  const int f0 = fb * kWrFrames + 32 * wave;         // first frame of this wave
  const int frame = f0 + col;                        // this lane's frame (standard orientation)
  const char* T2 = static_cast<const char*>(w.mlp_frags) + kT2Base;
  char* const S0 = L.slot[0];
  char* const S1 = L.slot[1];

  // ---- prologue.  Weight chunk 0 is requested first (the long pole: 64 KB per workgroup), then the parameter vectors (<= 4
  // values per thread) and this lane's GRU row, all unconditional loads in flight together; hipcc waits for every outstanding
  // request in front of the first use of a loaded register while an LDS-bound load is in flight, so chunk 1 is requested only
  // after that point, LAST: the first barrier waits for everything but the eight most recent requests (memory reads return in
  // order) and the first layer runs while chunk 1 is still arriving ----
  wr_dma(T2 + (size_t)kT2Kb[0][0] * 1024, S0, 64, wave, lane);
  const int grp = wave >> 1, pc = tid & 127;
  float pv[4];
  {
    // vector n of the parameter block goes to par[128 n ..]; quarter `grp` of the workgroup fetches vectors grp, grp + 4, ...
    // (compile-time struct indices selected by scalar compares: a run-time index into the by-value struct would be a private copy)
    auto sel4 = [&](const float* a, const float* b, const float* c, const float* d) { return grp == 0 ? a : grp == 1 ? b : grp == 2 ? c : d; };
    const float* const hb0 = path ? w.hgen_b[0] : w.newt_mlp_b[0];
    const float* const hg0 = path ? w.hgen_ln_g[0] : w.newt_ln_g[0];
    const float* const ht0 = path ? w.hgen_ln_b[0] : w.newt_ln_b[0];
    const float* const hb1 = path ? w.hgen_b[1] : w.newt_mlp_b[1];
    const float* const hg1 = path ? w.hgen_ln_g[1] : w.newt_ln_g[1];
    const float* const ht1 = path ? w.hgen_ln_b[1] : w.newt_ln_b[1];
    const float* const hb2 = path ? w.hgen_b[2] : w.newt_mlp_b[2];
    const float* const hg2 = path ? w.hgen_ln_g[2] : w.newt_ln_g[2];
    const float* const ht2 = path ? w.hgen_ln_b[2] : w.newt_ln_b[2];
    const float* const ob = path ? w.hgen_b[3] : w.newt_mlp_b[3];
    const float* const ox = path ? w.hgen_w[3] + (size_t)128 * NWS_HIDDEN : w.newt_mlp_b[3] + 128;
    pv[0] = sel4(w.proj_b, hb0, hg0, ht0)[pc];
    pv[1] = sel4(hb1, hg1, ht1, hb2)[pc];
    pv[2] = sel4(hg2, ht2, ob, ox)[pc];
    pv[3] = w.proj_b[pc];   // (vector 12 does not exist: quarter 0 rewrites vector 0 with the same values)
  }
  const float b128 = path ? w.hgen_b[3][128] : 0.0f;
  float4 in[16];
  {
    // (frames >= F read the last row: their results are never stored)
    const float4* src = reinterpret_cast<const float4*>(gru_out + (size_t)(frame < F ? frame : F - 1) * NWS_HIDDEN + 64 * half);
#pragma unroll
    for (int k = 0; k < 16; ++k) in[k] = src[k];
  }
  L.par[128 * grp + pc] = pv[0];
  L.par[128 * (4 + grp) + pc] = pv[1];
  L.par[128 * (8 + grp) + pc] = pv[2];
  if (grp == 0) L.par[pc] = pv[3];
  if (tid == 0) L.par[1536] = b128;
  // (the conversion of the GRU row stays in front of the request for chunk 1: a first use of a loaded register behind it
  // would make hipcc wait for chunk 1 as well)
  WrAct X;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const float y[8] = {in[2 * ks].x, in[2 * ks].y, in[2 * ks].z, in[2 * ks].w, in[2 * ks + 1].x, in[2 * ks + 1].y, in[2 * ks + 1].z, in[2 * ks + 1].w};
    wr_pack8(y, X.hi[ks], X.lo[ks]);
  }
  __builtin_amdgcn_sched_barrier(0);
  wr_dma(T2 + (size_t)(path ? kT2Kb[1][1] : kT2Kb[0][1]) * 1024, S1, 64, wave, lane);   // the wave's 8 most recent requests
  __builtin_amdgcn_sched_barrier(0);
  // barrier k closes interval k: every wave has finished reading slot (k - 1) & 1, and its share of chunk k (requested one
  // interval earlier) has landed; then chunk k + 1 goes into the slot just released
  // ABL == 6: cycle timeline - s_memtime at numbered points, every wave of workgroup (0, path), into emb_out as long long [path][wave][32]
  int probe_n = 0;
  auto probe = [&]() {
    if (ABL == 6 && fb == 0 && lane == 0 && probe_n < 32)
      reinterpret_cast<long long*>(emb_out)[(path * 8 + wave) * 32 + probe_n] = (long long)__builtin_readcyclecounter();
    ++probe_n;
  };
  probe();   // 0: prologue done (DMA requested, inputs converted)
  auto sync = [&](int k) {
    probe();
    if (k == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // everything but chunk 1
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (not __syncthreads(): hipcc puts a vmcnt(0) in front of it while LDS-bound loads are in flight, which would wait for
    // chunk 1 as well; the LDS writes of the parameters are drained by the lgkmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    probe();
    if (k >= 1 && k <= 4) wr_dma(T2 + (size_t)(path ? kT2Kb[1][k + 1] : kT2Kb[0][k + 1]) * 1024, ((k + 1) & 1) ? S1 : S0, (path == 1 && k == 4) ? 72 : 64, wave, lane);
  };
  sync(0);

  f32x16 v[4];
  // ---- layer 0: emb = proj(gru_out) ----
  wr_lane_vec(L.par, half, v);
  if (ABL != 2) wr_layer_mma<8, 4, false, false, ABL == 3>(S0, 0, X, v, lane);
  probe();
  if (TAPS && path == 0 && emb_out != nullptr && frame < F) {
    const int b = frame / T, t = frame - b * T;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) emb_out[((size_t)b * NWS_HIDDEN + 32 * mt + frag_row(r, half)) * T + t] = v[mt][r];
  }
  wr_pack<ABL == 5>(v, X);
  probe();
  sync(1);

  // ---- three hidden layers: X = LeakyReLU(LayerNorm(W X + b)) ----
  float h128 = 0.0f;
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    const float* par = L.par + 128 + 384 * l;
    const char* slot = (l & 1) ? S0 : S1;          // chunks 1, 2, 3 -> slots 1, 0, 1
    wr_lane_vec(par, half, v);
    if (ABL != 2) wr_layer_mma<8, 4, false, false, ABL == 3>(slot, 0, X, v, lane);
    probe();
    if (ABL != 1) wr_layer_norm(v, par + 128, par + 256, half);
    if (l == 2 && path == 1) {
      // band 128 of H: fp32 dot product of the frame's 128 activations with row 128 of the last layer
      f32x16 w128[4];
      wr_lane_vec(L.par + 1408, half, w128);
      float d0 = 0.0f, d1 = 0.0f, d2 = 0.0f, d3 = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        d0 = fmaf(v[0][r], w128[0][r], d0);
        d1 = fmaf(v[1][r], w128[1][r], d1);
        d2 = fmaf(v[2][r], w128[2][r], d2);
        d3 = fmaf(v[3][r], w128[3][r], d3);
      }
      float d = (d0 + d1) + (d2 + d3);
      d += nws_swap_halves(d);
      h128 = d + L.par[1536];
    }
    wr_pack<ABL == 5>(v, X);
    probe();
    sync(l + 2);
  }

  if (path == 0) {
    // ---- film = last layer of newt.mlp, transposed: lane = channel, register = frame; chunk 4 (slot 0): tiles 0..3, chunk 5: 4..7
    const __amdgpu_buffer_rsrc_t film_rs = __builtin_amdgcn_make_buffer_rsrc(film_out, 0, F * NWS_FILM_CH * 4, 0x00020000);
    const int lane_off = ((f0 + 4 * half) * NWS_FILM_CH + col) * 4;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const char* slot = c ? S1 : S0;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {           // two tiles at a time: two accumulator chains
        f32x16 acc[2];
        if (ABL != 2) wr_layer_mma<8, 2, true, true, ABL == 3>(slot, 2 * pr, X, acc, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wr_store_rows<NWS_FILM_CH, true>(film_rs, acc[j], lane_off, 4 * c + 2 * pr + j, L.par[1280 + 32 * (4 * c + 2 * pr + j) + col]);
      }
      probe();
      if (c == 0) sync(5);
    }
  } else {
    // ---- H = last layer of h_generator (bands 0..127 on the matrix pipe, band 128 from above) -> operand of the FIR design ----
    wr_lane_vec(L.par + 1280, half, v);
    if (ABL != 2) wr_layer_mma<8, 4, false, false, ABL == 3>(S0, 0, X, v, lane);
    if (TAPS && H_out != nullptr && frame < F) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) H_out[(size_t)frame * NWS_N_BANDS + 32 * mt + frag_row(r, half)] = v[mt][r];
      if (half == 0) H_out[(size_t)frame * NWS_N_BANDS + 128] = h128;
    }
    wr_pack(v, X);
    {
      const float y[8] = {half == 0 ? h128 : 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};   // kperm_d(8, 0, 0) = 128
      wr_pack8(y, X.hi[8], X.lo[8]);
    }
    sync(5);
    // ---- fir = D[128..255] H, transposed; K = 9 steps (129 bands) ----
    const __amdgpu_buffer_rsrc_t fir_rs = __builtin_amdgcn_make_buffer_rsrc(fir_out, 0, F * NWS_FIR_HALF * 4, 0x00020000);
    const int lane_off = ((f0 + 4 * half) * NWS_FIR_HALF + col) * 4;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      f32x16 acc[2];
      if (ABL != 2) wr_layer_mma<9, 2, true, true, ABL == 3>(S1, 2 * pr, X, acc, lane);
#pragma unroll
      for (int j = 0; j < 2; ++j) wr_store_rows<NWS_FIR_HALF, false>(fir_rs, acc[j], lane_off, 2 * pr + j, 0.0f);
    }
    probe();
  }
}

// second fragment table: the layers in the contraction orders of frame_mlps_wr_kernel
__global__ void mlp_frags2_kernel(NwsWeights w, const float* __restrict__ fir_design, f16x8* __restrict__ out) {
  // chunks: 0 proj | 1-3 newt hidden | 4 newt out (8 M-tiles) | 5-7 hgen hidden | 8 hgen out rows 0..127 | 9 FIR design (9 K-steps)
  constexpr int kPairs[10] = {2048, 2048, 2048, 2048, 4096, 2048, 2048, 2048, 2048, 2304};   // (hi, lo) pairs per chunk
  constexpr int kKb[10] = {0, 64, 128, 192, 256, 384, 448, 512, 576, 640};
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  int id = 0;
  while (id < 10 && e >= kPairs[id]) e -= kPairs[id++];
  if (id >= 10) return;
  const int KS = id == 9 ? 9 : 8;
  const int li = e & 63, ks = (e >> 6) % KS, mt = (e >> 6) / KS;
  const int row = 32 * mt + (li & 31), h = li >> 5;
  const float* W;
  int rows, ld, kmax;
  if (id == 0) { W = w.proj_w; rows = 128; ld = 128; kmax = 128; }
  else if (id <= 4) { W = w.newt_mlp_w[id - 1]; rows = id == 4 ? 256 : 128; ld = 128; kmax = 128; }
  else if (id <= 8) { W = w.hgen_w[id - 5]; rows = 128; ld = 128; kmax = 128; }
  else { W = fir_design + (size_t)NWS_FIR_HALF * kDK; rows = NWS_FIR_HALF; ld = kDK; kmax = NWS_N_BANDS; }
  f16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = id == 0 ? kperm_in(ks, h, i) : (ks < 8 ? kperm_d(ks, h, i) : 128 + 4 * h + i + (i >= 4 ? 4 : 0));
    const float v = (row < rows && k < kmax) ? W[(size_t)row * ld + k] : 0.0f;
    hi[i] = (_Float16)v;
    lo[i] = (_Float16)(v - (float)hi[i]);
  }
  f16x8* dst = out + (size_t)kKb[id] * 64 + ((size_t)(mt * KS + ks) * 2) * 64 + li;
  dst[0] = hi;
  dst[64] = lo;
}

// one thread per fragment pair: 8 consecutive-k weights of one row, split into hi / lo
__global__ void mlp_frags_kernel(NwsWeights w, const float* __restrict__ fir_design, f16x8* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= kFragTotal / 2) return;
  // locate the map: entries are counted in (hi,lo) PAIRS here, so bases are halved
  int id = 9;
#pragma unroll
  for (int i = 9; i >= 0; --i)
    if (e < (i == 9 ? kFragTotal / 2 : frag_map(i + 1).base / 2)) id = i;
  const FragMap m = frag_map(id);
  const int local = e - m.base / 2;          // [mt][ks][h*32 + i]
  const int li = local & 63, ks = (local >> 6) % m.ks, mt = (local >> 6) / m.ks;
  const int row = 32 * mt + (li & 31), k0 = 16 * ks + 8 * (li >> 5);
  const float* W;
  int rows, ld, kmax;
  if (id == 0) { W = w.proj_w; rows = 128; ld = 128; kmax = 128; }
  else if (id <= 4) { W = w.newt_mlp_w[id - 1]; rows = id == 4 ? 256 : 128; ld = 128; kmax = 128; }
  else if (id <= 8) { W = w.hgen_w[id - 5]; rows = id == 8 ? NWS_N_BANDS : 128; ld = 128; kmax = 128; }
  else { W = fir_design + (size_t)NWS_FIR_HALF * kDK; rows = NWS_FIR_HALF; ld = kDK; kmax = NWS_N_BANDS; }   // D rows 128..255
  f16x8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = k0 + i;
    const float v = (row < rows && k < kmax) ? W[(size_t)row * ld + k] : 0.0f;
    hi[i] = (_Float16)v;
    lo[i] = (_Float16)(v - (float)hi[i]);
  }
  f16x8* dst = out + m.base + ((size_t)mt * m.ks + ks) * 128 + li;
  dst[0] = hi;
  dst[64] = lo;
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

// A handful of frames per utterance (streaming hop): matrix-vector form, one workgroup per (path, utterance) - mlp_few.h.
// Streaming: one more pair of workgroups behind the B utterances; the first of them runs the hop's shared head (nws_common.h).
template <int NF>
__global__ __launch_bounds__(256) void frame_mlps_few_kernel(NwsWeights w, const float* __restrict__ gru_out, int T,
                                                             float* __restrict__ film_out, float* __restrict__ fir_out, int out_T,
                                                             int out_off, NwsStreamNoiseWin win, long long* probe) {
  if (win.nzwin != nullptr && blockIdx.y == gridDim.y - 1) {
    if (blockIdx.x == 0) nws_stream_noise_window_block<256>(win, threadIdx.x);
    return;
  }
  __shared__ __attribute__((aligned(16))) NwsFewLds L;
  nws_mlp_few_path<NF>(L, w, gru_out, T, blockIdx.y, blockIdx.x, film_out, fir_out, out_T, out_off, threadIdx.x, [] { return true; }, probe);
}

}  // namespace

extern "C" {

int nws_fir_design_matrix(const float* window, float* D_out, void* stream) {
  if (!window || !D_out) return NWS_ERR_BAD_ARG;
  fir_design_kernel<<<NWS_FIR_LEN, 192, 0, (hipStream_t)stream>>>(window, D_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_mlp_frags(const NwsWeights* w, const float* fir_design, void* frags_out, void* stream) {
  if (!w || !fir_design || !frags_out || !w->proj_w) return NWS_ERR_BAD_ARG;
  for (int i = 0; i < 4; ++i)
    if (!w->newt_mlp_w[i] || !w->hgen_w[i]) return NWS_ERR_BAD_ARG;
  mlp_frags_kernel<<<(kFragTotal / 2 + 255) / 256, 256, 0, (hipStream_t)stream>>>(*w, fir_design,
                                                                                   static_cast<f16x8*>(frags_out));
  NWS_CHECK_LAUNCH();
  mlp_frags2_kernel<<<(kT2Bytes / 32 + 255) / 256, 256, 0, (hipStream_t)stream>>>(
      *w, fir_design, reinterpret_cast<f16x8*>(static_cast<char*>(frags_out) + kT2Base));
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// measurements / tests: 0 automatic (by frame count, NWS_MLP_KERNEL), 1 tile kernels (frame_mlps16 / 64), 2 wave-resident frames
static int g_mlp_kernel_mode = 0, g_mlp_dbg = 0;
static void* g_mlp_probe = nullptr;
int nws_debug_frame_mlps_probe(void* buf) {
  g_mlp_probe = buf;   // 2 x 8 x 32 long long: s_memtime timeline of workgroups (0, path), ablation 6
  return NWS_OK;
}
int nws_debug_frame_mlps_kernel(int mode) {
  if (mode < 0 || (mode & 15) > 2) return NWS_ERR_BAD_ARG;
  g_mlp_kernel_mode = mode & 15;
  g_mlp_dbg = mode >> 4;     // bits 8..10: timing ablation
  return NWS_OK;
}

// NWS_MLP_FEW=0: utterances of one or two frames take the 32-frame tile kernel like longer ones (measurements, bit-identity tests)
static bool few_frames_enabled() {
  static const bool on = [] {
    const char* e = getenv("NWS_MLP_FEW");
    return e == nullptr || e[0] != '0';
  }();
  return on;
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
  static unsigned long long attr_devices = 0;
  if (nws_first_use_on_device(attr_devices)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds));
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps16_kernel<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds16));
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps16_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds16));
    if (e != hipSuccess) return (int)e;
  }
  // enough frames to give most CUs a 256-frame workgroup: wave-resident frames (NWS_MLP_KERNEL=tiles keeps the kernels below)
  static const int env_mode = [] {
    const char* e = getenv("NWS_MLP_KERNEL");
    return e == nullptr ? 0 : strcmp(e, "tiles") == 0 ? 1 : strcmp(e, "frames") == 0 ? 2 : 0;
  }();
  const int mode = g_mlp_kernel_mode ? g_mlp_kernel_mode : env_mode;
  const long long F = (long long)B * T;
  // (from the point where the 64-frame tile kernel needs a second round of workgroups: 256 tiles)
  const bool many = (long long)B * ((T + kFT2 - 1) / kFT2) > 256;
  // (the wave-resident kernel addresses its outputs through buffer resources with 32-bit BYTE offsets - num_records and the
  // per-lane offsets of the film rows, the widest output: F * 256 * 4 bytes must stay below 2^31, i.e. F < 2^21 frames = 64 x 8.7 min;
  // beyond that the tile kernels below, which index with size_t, take over)
  if (w->mlp_frags != nullptr && mode != 1 && (many || mode == 2) && F * NWS_FILM_CH * (long long)sizeof(float) < (1ll << 31)) {
    static unsigned long long attr_wr = 0;
    if (nws_first_use_on_device(attr_wr)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps_wr_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(WrLds));
      if (e != hipSuccess) return (int)e;
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps_wr_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sizeof(WrLds));
      if (e != hipSuccess) return (int)e;
    }
    // both paths of a frame block on one XCD (see the kernel); NWS_MLP_XCD=0 restores grid (blocks, 2) (measurements)
    static const bool xcd_map = [] { const char* e = getenv("NWS_MLP_XCD"); return !(e && e[0] == '0'); }();
    const unsigned nblk = (unsigned)((F + kWrFrames - 1) / kWrFrames);
    const int xcd_blocks = xcd_map ? (int)nblk : 0;
    const dim3 gridw = xcd_map ? dim3(16 * ((nblk + 7) / 8), 1) : dim3(nblk, 2);
    const int abl = g_mlp_dbg >> 4;
    if (abl == 6 && g_mlp_probe) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps_wr_kernel<false, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(WrLds));
      frame_mlps_wr_kernel<false, 6><<<gridw, 512, sizeof(WrLds), (hipStream_t)stream>>>(*w, gru_out, (int)F, T, static_cast<float*>(g_mlp_probe), film_out, nullptr, fir_out, xcd_blocks);
    } else if (abl == 1 || abl == 2 || abl == 3 || abl == 5) {
      auto fn = abl == 1 ? frame_mlps_wr_kernel<false, 1> : abl == 2 ? frame_mlps_wr_kernel<false, 2> : abl == 3 ? frame_mlps_wr_kernel<false, 3> : frame_mlps_wr_kernel<false, 5>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(WrLds));
      fn<<<gridw, 512, sizeof(WrLds), (hipStream_t)stream>>>(*w, gru_out, (int)F, T, nullptr, film_out, nullptr, fir_out, xcd_blocks, -1);
    } else if (abl == 7 || abl == 8) {   // one path's workgroups alone (the other path's outputs are not written)
      frame_mlps_wr_kernel<false><<<gridw, 512, sizeof(WrLds), (hipStream_t)stream>>>(*w, gru_out, (int)F, T, nullptr, film_out, nullptr, fir_out, xcd_blocks, abl - 7);
    } else if (!emb_out && !H_out)
      frame_mlps_wr_kernel<false><<<gridw, 512, sizeof(WrLds), (hipStream_t)stream>>>(*w, gru_out, (int)F, T, nullptr, film_out, nullptr, fir_out, xcd_blocks);
    else
      frame_mlps_wr_kernel<true><<<gridw, 512, sizeof(WrLds), (hipStream_t)stream>>>(*w, gru_out, (int)F, T, emb_out, film_out, H_out, fir_out, xcd_blocks);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  // more than one 32-frame tile per utterance: 64-frame tiles.  NWS_MLP_TILE=32 keeps the 32-frame kernel for A/B timing
  // (same box, back to back at B=64, T=500: 66.4 us per call with 32-frame tiles, 62.5 with 64-frame tiles)
  static const bool tile64 = [] {
    const char* e = getenv("NWS_MLP_TILE");
    return e == nullptr || atoi(e) != 32;
  }();
  // (only when the 64-frame grid still gives every CU a workgroup: at B = 1 the 16 workgroups of the 32-frame kernel finish
  // a 4 s clip sooner than 8 twice as long ones - batch-1 latency 0.305 against 0.32 ms)
  if (w->mlp_frags != nullptr && T > kFT && tile64 && (long long)B * ((T + kFT2 - 1) / kFT2) >= 256) {
    static unsigned long long attr64 = 0;
    if (nws_first_use_on_device(attr64)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps64_kernel<false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds64));
      if (e != hipSuccess) return (int)e;
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)sizeof(MlpLds64));
      if (e != hipSuccess) return (int)e;
    }
    const dim3 grid64((T + kFT2 - 1) / kFT2, B);
    if (!emb_out && !H_out)
      frame_mlps64_kernel<false><<<grid64, 512, sizeof(MlpLds64), (hipStream_t)stream>>>(*w, gru_out, T, nullptr, film_out, nullptr, fir_out);
    else
      frame_mlps64_kernel<true><<<grid64, 512, sizeof(MlpLds64), (hipStream_t)stream>>>(*w, gru_out, T, emb_out, film_out, H_out, fir_out);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  // one or two frames per utterance (256-sample streaming buffers, scripts/time_buffer_sizes.py): the matrix-vector form of mlp_few.h
  // (four frames through the NF = 4 instantiation measured no faster than the tile kernel: 65.5 against 64.8 us per 512-sample buffer)
  if (w->mlp_frags != nullptr && !emb_out && !H_out && T <= 2 && mode == 0 && few_frames_enabled()) {
    frame_mlps_few_kernel<2><<<dim3(2, B), 256, 0, (hipStream_t)stream>>>(*w, gru_out, T, film_out, fir_out, T, 0, NwsStreamNoiseWin{},
                                                                          static_cast<long long*>(g_mlp_probe));
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  const dim3 grid((T + kFT - 1) / kFT, B);
  if (w->mlp_frags != nullptr && !emb_out && !H_out)
    frame_mlps16_kernel<false><<<grid, 512, sizeof(MlpLds16), (hipStream_t)stream>>>(*w, gru_out, T, nullptr, film_out,
                                                                                       nullptr, fir_out, T, 0, NwsStreamNoiseWin{});
  else if (w->mlp_frags != nullptr)
    frame_mlps16_kernel<true><<<grid, 512, sizeof(MlpLds16), (hipStream_t)stream>>>(*w, gru_out, T, emb_out, film_out,
                                                                                      H_out, fir_out, T, 0, NwsStreamNoiseWin{});
  else
    frame_mlps_kernel<<<grid, 256, sizeof(MlpLds), (hipStream_t)stream>>>(*w, gru_out, fir_design, T, emb_out, film_out,
                                                                          H_out, fir_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// Streaming hop (stream.hip): the 32-frame tile kernel on the T <= 32 new frames of every utterance; rows land in the windows the
// oscillator / noise kernels read, one more workgroup runs the hop's shared head (nws_common.h)
int nws_frame_mlps_stream(const NwsWeights* w, const float* gru_out, int B, int T, float* film_w, float* fir_w, int out_T,
                          int out_off, const NwsStreamNoiseWin* win, void* stream) {
  if (!w || !gru_out || !film_w || !fir_w || !win || !win->nzwin || !win->counters || B <= 0 || T <= 0 || out_off < 0 ||
      out_off + T > out_T)
    return NWS_ERR_BAD_ARG;
  if (w->mlp_frags == nullptr || T > kFT || B > 65534) return NWS_ERR_UNSUPPORTED;
  static unsigned long long attr_devices = 0;
  if (nws_first_use_on_device(attr_devices)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps16_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds16));
    if (e != hipSuccess) return (int)e;
  }
  // one or two frames: the matrix-vector form (NWS_MLP_FEW=0 keeps the tile kernel: measurements, bit-identity tests)
  if (few_frames_enabled() && T <= 2) {
    frame_mlps_few_kernel<2><<<dim3(2, B + 1), 256, 0, (hipStream_t)stream>>>(*w, gru_out, T, film_w, fir_w, out_T, out_off, *win, nullptr);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  frame_mlps16_kernel<false><<<dim3(1, B + 1), 512, sizeof(MlpLds16), (hipStream_t)stream>>>(*w, gru_out, T, nullptr, film_w, nullptr,
                                                                                              fir_w, out_T, out_off, *win);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
