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
  // 0 proj | 1-3 newt hidden | 4 newt out (256) | 5-7 hgen hidden | 8 hgen out (129 -> 160) | 9 FIR design (K 132 -> 144)
  return id == 0 ? FragMap{0, 4, 8}
       : id <= 3 ? FragMap{4096 * id, 4, 8}
       : id == 4 ? FragMap{16384, 8, 8}
       : id <= 7 ? FragMap{24576 + 4096 * (id - 5), 4, 8}
       : id == 8 ? FragMap{36864, 5, 8}
                 : FragMap{41984, 8, 9};
}
constexpr int kFragTotal = 51200;  // x 16 B = 819200 B

struct MlpLds16 {
  char emb[2][kXtBytes];  // [hi|lo]
  char p0[2][kXtBytes];
  char p1[2][kXtBytes];
  float stage[4][kFT * kXS];
  float red[2][4][kFT];
};

__device__ __forceinline__ void split4_store(char* xt_hi, char* xt_lo, int frame, int ch, float a, float b, float c, float d) {
  f16x4 h, l;
  const float v[4] = {a, b, c, d};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (_Float16)v[i];
    l[i] = (_Float16)(v[i] - (float)h[i]);
  }
  *reinterpret_cast<f16x4*>(xt_hi + frame * kRowB + ch * 2) = h;
  *reinterpret_cast<f16x4*>(xt_lo + frame * kRowB + ch * 2) = l;
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

// acc(32 rows x 32 frames) = A * X  for the XT buffer (hi, lo):  W_hi X_hi + (W_hi X_lo + W_lo X_hi)
template <int KS>
__device__ __forceinline__ void mma_tile(const AFrag<KS>& A, const char* xt_hi, const char* xt_lo, int lane, f32x16& acc) {
  const int half = lane >> 5, col = lane & 31;
  f32x16 cross;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc[r] = 0.0f;
    cross[r] = 0.0f;
  }
  const char* bh = xt_hi + col * kRowB + half * 16;
  const char* bl = xt_lo + col * kRowB + half * 16;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const f16x8 xh = *reinterpret_cast<const f16x8*>(bh + ks * 32);
    const f16x8 xl = *reinterpret_cast<const f16x8*>(bl + ks * 32);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[ks], xh, acc, 0, 0, 0);
    cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.hi[ks], xl, cross, 0, 0, 0);
    cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.lo[ks], xh, cross, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += cross[r];
}

// write a 32-channel x 32-frame tile held in the D layout into an XT buffer (4 consecutive channels per store)
__device__ __forceinline__ void store_tile_xt(char* xt_hi, char* xt_lo, int c0, const float v[16], int lane) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    split4_store(xt_hi, xt_lo, col, c0 + 8 * g + 4 * half, v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// bias + LayerNorm(channels) + LeakyReLU on the accumulator tile of wave `wave`, result -> XT buffer `out`
__device__ __forceinline__ void ln_epilogue(MlpLds16& L, const f32x16& acc, const float* bias, const float* ln_g,
                                            const float* ln_b, char* out_hi, char* out_lo, int wave, int lane) {
  const int half = lane >> 5, col = lane & 31;
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
    v[r] = nws_leaky_relu((v[r] - mean) * rstd * ln_g[c] + ln_b[c]);
  }
  store_tile_xt(out_hi, out_lo, 32 * wave, v, lane);
  __syncthreads();
}

// The layer sequence is static, so the weight fragments are software-pipelined by hand: while layer l's epilogue
// (LayerNorm: two workgroup barriers and an LDS exchange) runs, layer l+1's 16 KB of fragments are already in flight
// from L2 into the other register set (A0 / A1 alternate; a barrier is a memory fence to hipcc, so loads written
// after it would only be issued once every wave has arrived).
__global__ __launch_bounds__(256, 2) void frame_mlps16_kernel(NwsWeights w, const float* __restrict__ gru_out, int T,
                                                              float* __restrict__ emb_out, float* __restrict__ film_out,
                                                              float* __restrict__ H_out, float* __restrict__ fir_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  MlpLds16& L = *reinterpret_cast<MlpLds16*>(smem_raw);
  const f16x8* F = reinterpret_cast<const f16x8*>(w.mlp_frags);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kFT;
  const int frames_valid = T - t0 < kFT ? T - t0 : kFT;
  AFrag<8> A0, A1;
  AFrag<9> A9;
  f32x16 acc;

  load_frags<8>(A0, F + frag_map(0).base, wave, lane);  // proj
  // ---- gru_out tile -> XT p0 (4 channels per thread per pass) ----
  for (int e = tid; e < kFT * (NWS_HIDDEN / 4); e += 256) {
    const int f = e >> 5, c4 = (e & 31) * 4;
    float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (f < frames_valid) v = *reinterpret_cast<const float4*>(&gru_out[((size_t)b * T + t0 + f) * NWS_HIDDEN + c4]);
    split4_store(L.p0[0], L.p0[1], f, c4, v.x, v.y, v.z, v.w);
  }
  // zero the K padding (channels 128..143) of p1, which will hold H for the FIR-design contraction
  for (int e = tid; e < kFT * 2 * 2; e += 256) {
    const int f = e >> 2, part = e & 3;
    *reinterpret_cast<float4*>(L.p1[part >> 1] + f * kRowB + 256 + (part & 1) * 16) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  __syncthreads();

  // ---- emb = proj(gru_out) ----
  mma_tile<8>(A0, L.p0[0], L.p0[1], lane, acc);
  load_frags<8>(A1, F + frag_map(1).base, wave, lane);  // newt hidden 0
  {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * wave + frag_row(r, half);
      v[r] = acc[r] + w.proj_b[c];
      if (emb_out != nullptr && col < frames_valid) emb_out[((size_t)b * NWS_HIDDEN + c) * T + t0 + col] = v[r];
    }
    store_tile_xt(L.emb[0], L.emb[1], 32 * wave, v, lane);
  }
  __syncthreads();

  // ---- film = newt.mlp(emb) ----
  mma_tile<8>(A1, L.emb[0], L.emb[1], lane, acc);
  load_frags<8>(A0, F + frag_map(2).base, wave, lane);
  ln_epilogue(L, acc, w.newt_mlp_b[0], w.newt_ln_g[0], w.newt_ln_b[0], L.p0[0], L.p0[1], wave, lane);
  mma_tile<8>(A0, L.p0[0], L.p0[1], lane, acc);
  load_frags<8>(A1, F + frag_map(3).base, wave, lane);
  ln_epilogue(L, acc, w.newt_mlp_b[1], w.newt_ln_g[1], w.newt_ln_b[1], L.p1[0], L.p1[1], wave, lane);
  mma_tile<8>(A1, L.p1[0], L.p1[1], lane, acc);
  load_frags<8>(A0, F + frag_map(4).base, wave, lane);      // newt out, M-tile wave
  ln_epilogue(L, acc, w.newt_mlp_b[2], w.newt_ln_g[2], w.newt_ln_b[2], L.p0[0], L.p0[1], wave, lane);
  {
    float v[16];
    mma_tile<8>(A0, L.p0[0], L.p0[1], lane, acc);
    load_frags<8>(A1, F + frag_map(4).base, wave + 4, lane);  // newt out, M-tile wave+4
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + w.newt_mlp_b[3][32 * wave + frag_row(r, half)];
    store_tile_frame_major(L.stage[wave], v, lane, film_out + ((size_t)b * T + t0) * NWS_FILM_CH + 32 * wave, NWS_FILM_CH,
                           frames_valid);
    mma_tile<8>(A1, L.p0[0], L.p0[1], lane, acc);
    load_frags<8>(A0, F + frag_map(5).base, wave, lane);      // hgen hidden 0
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r] + w.newt_mlp_b[3][32 * (wave + 4) + frag_row(r, half)];
    store_tile_frame_major(L.stage[wave], v, lane, film_out + ((size_t)b * T + t0) * NWS_FILM_CH + 32 * (wave + 4),
                           NWS_FILM_CH, frames_valid);
  }
  __syncthreads();

  // ---- H = h_generator(emb) ----
  mma_tile<8>(A0, L.emb[0], L.emb[1], lane, acc);
  load_frags<8>(A1, F + frag_map(6).base, wave, lane);
  ln_epilogue(L, acc, w.hgen_b[0], w.hgen_ln_g[0], w.hgen_ln_b[0], L.p0[0], L.p0[1], wave, lane);
  mma_tile<8>(A1, L.p0[0], L.p0[1], lane, acc);
  load_frags<8>(A0, F + frag_map(7).base, wave, lane);
  ln_epilogue(L, acc, w.hgen_b[1], w.hgen_ln_g[1], w.hgen_ln_b[1], L.p1[0], L.p1[1], wave, lane);
  mma_tile<8>(A0, L.p1[0], L.p1[1], lane, acc);
  load_frags<8>(A1, F + frag_map(8).base, wave, lane);      // hgen out, M-tile wave
  ln_epilogue(L, acc, w.hgen_b[2], w.hgen_ln_g[2], w.hgen_ln_b[2], L.p0[0], L.p0[1], wave, lane);
  // 129 outputs: M-tiles 0..3 by the four waves, M-tile 4 (row 128 only) by wave 0; H -> p1 channels 0..128,
  // channels 129..143 stay zero (zero weights, zero bias)
  {
    float v[16];
    mma_tile<8>(A1, L.p0[0], L.p0[1], lane, acc);
    load_frags<9>(A9, F + frag_map(9).base, wave, lane);     // FIR design, M-tile wave
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int c = 32 * wave + frag_row(r, half);
      v[r] = acc[r] + w.hgen_b[3][c];
      if (H_out != nullptr && col < frames_valid) H_out[((size_t)b * T + t0 + col) * NWS_N_BANDS + c] = v[r];
    }
    store_tile_xt(L.p1[0], L.p1[1], 32 * wave, v, lane);
    if (wave == 0) {
      load_frags<8>(A0, F + frag_map(8).base, 4, lane);      // M-tile 4 = row 128 (not prefetched: register budget)
      mma_tile<8>(A0, L.p0[0], L.p0[1], lane, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 128 + frag_row(r, half);
        v[r] = c < NWS_N_BANDS ? acc[r] + w.hgen_b[3][c] : 0.0f;
        if (H_out != nullptr && c < NWS_N_BANDS && col < frames_valid) H_out[((size_t)b * T + t0 + col) * NWS_N_BANDS + c] = v[r];
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)  // rows 128..143 only: the XT row holds 144 channels
        split4_store(L.p1[0], L.p1[1], col, 128 + 8 * g + 4 * half, v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
    }
  }
  __syncthreads();

  // ---- fir = D * H  (256 taps, K = 144 padded) ----
  {
    float v[16];
    mma_tile<9>(A9, L.p1[0], L.p1[1], lane, acc);
    load_frags<9>(A9, F + frag_map(9).base, wave + 4, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    store_tile_frame_major(L.stage[wave], v, lane, fir_out + ((size_t)b * T + t0) * NWS_FIR_LEN + 32 * wave, NWS_FIR_LEN,
                           frames_valid);
    mma_tile<9>(A9, L.p1[0], L.p1[1], lane, acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    store_tile_frame_major(L.stage[wave], v, lane, fir_out + ((size_t)b * T + t0) * NWS_FIR_LEN + 32 * (wave + 4),
                           NWS_FIR_LEN, frames_valid);
  }
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
  else { W = fir_design; rows = NWS_FIR_LEN; ld = kDK; kmax = NWS_N_BANDS; }
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
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(frame_mlps16_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(MlpLds16));
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const dim3 grid((T + kFT - 1) / kFT, B);
  if (w->mlp_frags != nullptr)
    frame_mlps16_kernel<<<grid, 256, sizeof(MlpLds16), (hipStream_t)stream>>>(*w, gru_out, T, emb_out, film_out, H_out,
                                                                              fir_out);
  else
    frame_mlps_kernel<<<grid, 256, sizeof(MlpLds), (hipStream_t)stream>>>(*w, gru_out, fir_design, T, emb_out, film_out,
                                                                          H_out, fir_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
