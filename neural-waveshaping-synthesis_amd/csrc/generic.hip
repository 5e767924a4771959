// Runtime-size path: NeuralWaveshaping.forward for ANY gin configuration of the reference (models/neural_waveshaping.py:31-62,
// models/modules/shaping.py:15-65, generators.py:11-48, dynamic.py:20-40): n_harmonics, n_waveshapers, shaping_fn_size,
// TrainableNonlinearity.depth, GRU / embedding sizes, h_generator depth / width, ir_length, hop_length / control_hop,
// out_channels, Reverb.sr * length_in_seconds, FastNEWT table size / range.
//
// The fused kernels of exciter_newt.hip / frame_mlps.hip / control_gru.hip / fir_noise.hip are compiled for the one
// architecture the reference ships (gin/models/newt.gin).  Everything else runs here: plain fp32 VALU kernels with runtime
// sizes, one stage per launch, the reference's own rounding chains where the result depends on them (F0 upsample, fp64
// prefix sum, fl(fl(tau c)/sr), fl(fl(k phase) + shift), the LUT index chain).  Correct first: stage boundaries are
// materialised in the caller's workspace (the oscillator bank (B, K, N) included), nothing is tuned.  No kernel here indexes
// a private array with a runtime value (that would become scratch memory, which the build refuses): per-thread vectors of
// runtime length live in LDS.
#include <stdlib.h>
#include <string.h>

#include "nws_common.h"

typedef float gfloat16 __attribute__((ext_vector_type(16)));   // one 32x32 MFMA accumulator tile

namespace {

constexpr float kTauF = 6.283185307179586f;  // fl32(math.tau)
constexpr float kPiF = 3.141592653589793f;   // fl32(math.pi)

// F.upsample(x, T*hop, mode="linear") for any hop (align_corners=False), the arithmetic of nws_lerp_coeff with a runtime
// scale: src = (n + 0.5) * (T / N) - 0.5 with the scale computed like ATen's area_pixel_compute_scale (float(T) / float(N)).
struct GLerp {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ GLerp g_lerp_coeff(int n, int T, float scale) {
  // torch's CPU upsample_linear1d, bit for bit at every hop: the source index is ONE fused multiply-add (its AVX2 / AVX-512
  // builds contract scale * (dst + 0.5) - 0.5), the output fma(w0, x0, fl(w1 * x1)); for power-of-two hops the product is exact
  // and nothing depends on it, at hop 10 / 25 / 40 the two-rounding form moved 1-3 % of the samples by an ulp
  float src = fmaf((float)n + 0.5f, scale, -0.5f);
  src = src < 0.0f ? 0.0f : src;
  const int i0 = (int)src;
  GLerp c;
  c.i0 = i0 < T - 1 ? i0 : T - 1;
  c.i1 = i0 + 1 < T ? i0 + 1 : T - 1;
  c.w1 = src - (float)i0;
  c.w0 = 1.0f - c.w1;
  return c;
}

// ---- GRU (models/neural_waveshaping.py:21-25, torch.nn.GRU gate order [r; z; n]) ---------------------------------------
// W_hh transposed once per call (k-major) so that thread j's reads for a fixed k are coalesced over j.
__global__ void g_transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int r = i / cols, c = i - r * cols;
  wt[(size_t)c * rows + r] = w[i];
}

// one workgroup per utterance, thread j = hidden unit j (+ blockDim strides); h ping-pongs in LDS
__global__ __launch_bounds__(256) void g_gru_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh_t,
                                                    const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                    const float* __restrict__ control, int C_total, int C_in, int H, int T,
                                                    const float* __restrict__ h0, float* __restrict__ out,
                                                    float* __restrict__ hT) {
  extern __shared__ float lds[];   // h[2][H] | x[C_in]
  float* hbuf = lds;
  float* xs = lds + 2 * (size_t)H;
  const int b = blockIdx.x;
  for (int j = threadIdx.x; j < H; j += blockDim.x) hbuf[j] = h0 ? h0[(size_t)b * H + j] : 0.0f;
  __syncthreads();
  int cur = 0;
  const int H3 = 3 * H;
  for (int t = 0; t < T; ++t) {
    for (int c = threadIdx.x; c < C_in; c += blockDim.x) xs[c] = control[((size_t)b * C_total + c) * T + t];
    __syncthreads();
    const float* hp = hbuf + (size_t)cur * H;
    float* hn = hbuf + (size_t)(cur ^ 1) * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
      float ir = b_ih[j], iz = b_ih[H + j], in = b_ih[2 * H + j];
      for (int c = 0; c < C_in; ++c) {
        const float xv = xs[c];
        ir = fmaf(w_ih[(size_t)j * C_in + c], xv, ir);
        iz = fmaf(w_ih[(size_t)(H + j) * C_in + c], xv, iz);
        in = fmaf(w_ih[(size_t)(2 * H + j) * C_in + c], xv, in);
      }
      float hr = b_hh[j], hz = b_hh[H + j], hnn = b_hh[2 * H + j];
      // sixteen rows of W_hh^T requested before the first of their FMAs (the same FMAs in the same order: as a plain loop hipcc
      // waited for every row's loads in turn, one L2 latency per k: 15 us per step at H = 128)
      int k = 0;
      for (; k + 16 <= H; k += 16) {
        float wa[16], wb[16], wc[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float* wr = w_hh_t + (size_t)(k + u) * H3;
          wa[u] = wr[j];
          wb[u] = wr[H + j];
          wc[u] = wr[2 * H + j];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float hv = hp[k + u];
          hr = fmaf(wa[u], hv, hr);
          hz = fmaf(wb[u], hv, hz);
          hnn = fmaf(wc[u], hv, hnn);
        }
      }
      for (; k < H; ++k) {
        const float hv = hp[k];
        const float* wr = w_hh_t + (size_t)k * H3;
        hr = fmaf(wr[j], hv, hr);
        hz = fmaf(wr[H + j], hv, hz);
        hnn = fmaf(wr[2 * H + j], hv, hnn);
      }
      const float r = 1.0f / (1.0f + expf(-(ir + hr)));
      const float z = 1.0f / (1.0f + expf(-(iz + hz)));
      const float nn = tanhf(in + r * hnn);
      const float hv = (1.0f - z) * nn + z * hp[j];
      hn[j] = hv;
      out[((size_t)b * T + t) * H + j] = hv;
    }
    __syncthreads();
    cur ^= 1;
  }
  if (hT)
    for (int j = threadIdx.x; j < H; j += blockDim.x) hT[(size_t)b * H + j] = hbuf[(size_t)cur * H + j];
}

// ---- the same recurrence with W_hh in registers (round 4; hidden sizes up to 128: 3 H^2 floats fit the registers of 4 H lanes) -----------------------------------------
// g_gru_kernel re-reads the 3 H^2 weights from L2 every step (196 KB at H = 128: 2.8 ms per 500 steps).  Here four adjacent
// lanes share a hidden unit: lane (j, q) keeps rows j, H + j, 2H + j of W_hh for k in [q KQ, (q + 1) KQ) in 3 KQ registers
// (KQ = compile-time bucket 8 / 16 / 32 >= H / 4, zero padded), reads its quarter of h from LDS (broadcast across the
// units), and the quarters meet in two DPP quad steps.  The input projection of gate g is computed by lane q = g (W_ih in LDS).
// One barrier per step; h ping-pongs in LDS.
template <int CTRL>
__device__ __forceinline__ float g_quad_perm(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float g_quad_sum(float v) {
  v += g_quad_perm<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += g_quad_perm<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  return v;
}

template <int KQ>
__global__ __launch_bounds__(16 * KQ) void g_gru_q_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh,
                                                       const float* __restrict__ b_ih, const float* __restrict__ b_hh,
                                                       const float* __restrict__ control, int C_total, int C_in, int H, int T,
                                                       const float* __restrict__ h0, float* __restrict__ out,
                                                       float* __restrict__ hT) {
  extern __shared__ float lds[];   // h[2][4 KQ] | x[2][C_in] | W_ih (3 H x C_in)
  float* hbuf = lds;
  float* xs = lds + 2 * 4 * KQ;
  float* wih = xs + 2 * C_in;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int j = tid >> 2, q = tid & 3;
  const bool unit = j < H;
  float wr[KQ], wz[KQ], wn[KQ];
#pragma unroll
  for (int i = 0; i < KQ; ++i) {
    const int k = q * KQ + i;
    const bool in = unit && k < H;
    wr[i] = in ? w_hh[(size_t)j * H + k] : 0.0f;
    wz[i] = in ? w_hh[(size_t)(H + j) * H + k] : 0.0f;
    wn[i] = in ? w_hh[(size_t)(2 * H + j) * H + k] : 0.0f;
  }
  for (int i = tid; i < 2 * 4 * KQ; i += blockDim.x) hbuf[i] = (i < H && h0) ? h0[(size_t)b * H + i] : 0.0f;
  for (int i = tid; i < 3 * H * C_in; i += blockDim.x) wih[i] = w_ih[i];
  if (tid < C_in) xs[tid] = control[((size_t)b * C_total + tid) * T];
  // frame t + 2 travels from memory while step t runs and reaches LDS during step t + 1: no load latency inside a step
  float xnext = (tid < C_in && T > 1) ? control[((size_t)b * C_total + tid) * T + 1] : 0.0f;
  // this lane's input-projection gate (q = 0, 1, 2: r, z, n) and the recurrent biases
  const int grow = (q < 3 ? q : 0) * H + (unit ? j : 0);
  const float bi = b_ih[grow];
  const float bhr = unit ? b_hh[j] : 0.0f, bhz = unit ? b_hh[H + j] : 0.0f, bhn = unit ? b_hh[2 * H + j] : 0.0f;
  __syncthreads();
  int cur = 0;
  for (int t = 0; t < T; ++t) {
    const float* hp = hbuf + cur * 4 * KQ;
    const float* xc = xs + (t & 1) * C_in;
    if (tid < C_in && t + 1 < T) {
      xs[((t + 1) & 1) * C_in + tid] = xnext;
      if (t + 2 < T) xnext = control[((size_t)b * C_total + tid) * T + t + 2];
    }
    float pr = 0.0f, pz = 0.0f, pn = 0.0f;
#pragma unroll
    for (int i4 = 0; i4 < KQ; i4 += 4) {
      const float4 hv = *reinterpret_cast<const float4*>(hp + q * KQ + i4);
      pr = fmaf(wr[i4], hv.x, pr); pz = fmaf(wz[i4], hv.x, pz); pn = fmaf(wn[i4], hv.x, pn);
      pr = fmaf(wr[i4 + 1], hv.y, pr); pz = fmaf(wz[i4 + 1], hv.y, pz); pn = fmaf(wn[i4 + 1], hv.y, pn);
      pr = fmaf(wr[i4 + 2], hv.z, pr); pz = fmaf(wz[i4 + 2], hv.z, pz); pn = fmaf(wn[i4 + 2], hv.z, pn);
      pr = fmaf(wr[i4 + 3], hv.w, pr); pz = fmaf(wz[i4 + 3], hv.w, pz); pn = fmaf(wn[i4 + 3], hv.w, pn);
    }
    float ig = bi;                       // gate q of W_ih x + b_ih
    for (int c = 0; c < C_in; ++c) ig = fmaf(wih[(size_t)grow * C_in + c], xc[c], ig);
    pr += q == 0 ? ig : 0.0f;            // r and z: input and recurrent parts simply add
    pz += q == 1 ? ig : 0.0f;
    const float in_n = g_quad_perm<0xAA>(ig);     // quad_perm [2, 2, 2, 2]: the n gate's input part, kept apart
    const float sr = g_quad_sum(pr) + bhr, sz = g_quad_sum(pz) + bhz, sn = g_quad_sum(pn) + bhn;
    // sigmoid(x) = 1 / (1 + 2^(-x log2 e)); tanh(x) = 1 - 2 / (1 + 2^(2 x log2 e)) on v_exp_f32 / v_rcp_f32 (~1 ulp each, the
    // error class of the libm forms inside torch's CPU GRU); h' = [z h + (1 - z)] - 2 (1 - z) / (1 + 2^..): n is never rounded
    // at magnitude 1 (control_gru.hip: 1.9e-5 against a float64 GRU over 500 steps where "(1 - z) n + z h" gave 5.3e-5)
    const float kL2E = 1.4426950408889634f;
    const float r = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kL2E * sr));
    const float z = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-kL2E * sz));
    const float omz = 1.0f - z;
    const float base = fmaf(z, hp[unit ? j : 0], omz);
    const float hv = fmaf(-2.0f * omz, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.0f * kL2E * (in_n + r * sn))), base);
    if (unit && q == 0) {
      hbuf[(cur ^ 1) * 4 * KQ + j] = hv;
      out[((size_t)b * T + t) * H + j] = hv;
    }
    // LDS-only barrier: the store of h and the load of frame t + 2 stay in flight across it
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    cur ^= 1;
  }
  if (hT && unit && q == 0) hT[(size_t)b * H + j] = hbuf[cur * 4 * KQ + j];
}

// (B, T, H) -> (B, H, T) (the GRU writes frame-major, the Conv1d stacks read channel-major)
__global__ void g_bth_to_bht_kernel(const float* __restrict__ x, int T, int H, float* __restrict__ y) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * H) return;
  const int h = i / T, t = i - h * T;
  y[(size_t)b * H * T + i] = x[((size_t)b * T + t) * H + h];
}

// ---- phase: upsample + double-accumulated prefix sum + the reference's fp32 chain (generators.py:59) -------------------
// one workgroup (1024 threads) per utterance; thread = contiguous chunk; exact fp64 sums, so the chunking is immaterial
__global__ __launch_bounds__(1024) void g_phase_kernel(const float* __restrict__ f0, const float* __restrict__ f0_up_in, int T,
                                                       int N, float scale, float sample_rate, float* __restrict__ f0_up_out,
                                                       float* __restrict__ phase_out) {
  __shared__ double tot[1024];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (N + 1023) / 1024;
  const int n0 = tid * per, n1 = n0 + per < N ? n0 + per : N;
  const float* x = f0 ? f0 + (size_t)b * T : nullptr;
  auto value = [&](int n) {
    if (f0_up_in) return f0_up_in[(size_t)b * N + n];
    const GLerp L = g_lerp_coeff(n, T, scale);
    return fmaf(L.w0, x[L.i0], L.w1 * x[L.i1]);
  };
  double s = 0.0;
  for (int n = n0; n < n1; ++n) s += (double)value(n);
  tot[tid] = s;
  __syncthreads();
  // exclusive prefix over the 1024 chunk sums (Hillis-Steele on doubles in LDS)
  for (int off = 1; off < 1024; off <<= 1) {
    const double v = tid >= off ? tot[tid - off] : 0.0;
    __syncthreads();
    tot[tid] += v;
    __syncthreads();
  }
  double run = tid > 0 ? tot[tid - 1] : 0.0;
  for (int n = n0; n < n1; ++n) {
    const float v = value(n);
    run += (double)v;
    const float c = (float)run;                         // torch's CPU cumsum: accumulate in double, round per element
    const float tc = kTauF * c;                         // math.tau * cumsum      (one rounding)
    phase_out[(size_t)b * N + n] = __fdiv_rn(tc, sample_rate);   // ... / sample_rate (true division)
    if (f0_up_out) f0_up_out[(size_t)b * N + n] = v;
  }
}

// The same for long rows in two parallel passes (the kernel above is one serial chain of N / 1024 double additions per
// thread, twice, on ONE compute unit per utterance: 88 us at 4 s whatever the batch): workgroups of 4096 samples - their sums,
// then per workgroup the sum of the partials before it + a block scan of 256 x 16 consecutive samples.  Same double sums.
constexpr int kPhChunk = 4096, kPhPer = 16;
__device__ __forceinline__ float g_f0_value(const float* __restrict__ x, const float* __restrict__ up, int n, int T, float scale) {
  if (up) return up[n];
  const GLerp L = g_lerp_coeff(n, T, scale);
  return fmaf(L.w0, x[L.i0], L.w1 * x[L.i1]);
}
__device__ __forceinline__ double g_block_sum(double v, double* red) {      // 256 threads; the total in every thread
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return t;
}
__global__ __launch_bounds__(256) void g_phase_sum_kernel(const float* __restrict__ f0, const float* __restrict__ f0_up_in, int T,
                                                          int N, float scale, double* __restrict__ partial) {
  __shared__ double red[4];
  const int b = blockIdx.y, n0 = blockIdx.x * kPhChunk;
  const float* x = f0 ? f0 + (size_t)b * T : nullptr;
  const float* up = f0_up_in ? f0_up_in + (size_t)b * N : nullptr;
  double s = 0.0;
#pragma unroll 4
  for (int i = 0; i < kPhPer; ++i) {
    const int n = n0 + i * 256 + threadIdx.x;
    if (n < N) s += (double)g_f0_value(x, up, n, T, scale);
  }
  s = g_block_sum(s, red);
  if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void g_phase_scan_kernel(const float* __restrict__ f0, const float* __restrict__ f0_up_in, int T,
                                                           int N, float scale, float sample_rate, const double* __restrict__ partial,
                                                           float* __restrict__ f0_up_out, float* __restrict__ phase_out) {
  __shared__ double red[4];
  __shared__ double wtot[4];
  const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x;
  const float* x = f0 ? f0 + (size_t)b * T : nullptr;
  const float* up = f0_up_in ? f0_up_in + (size_t)b * N : nullptr;
  double base = 0.0;
  for (int j = tid; j < g; j += 256) base += partial[(size_t)b * gridDim.x + j];
  base = g_block_sum(base, red);
  const int n0 = g * kPhChunk + tid * kPhPer;
  float v[kPhPer];
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < kPhPer; ++i) {
    v[i] = n0 + i < N ? g_f0_value(x, up, n0 + i, T, scale) : 0.0f;
    s += (double)v[i];
  }
  // exclusive scan of the 256 thread sums: inclusive scan inside each wave, wave totals through LDS
  double inc = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(inc, off);
    if ((tid & 63) >= off) inc += o;
  }
  if ((tid & 63) == 63) wtot[tid >> 6] = inc;
  __syncthreads();
  double run = base + (inc - s);
  for (int w = 0; w < (tid >> 6); ++w) run += wtot[w];
#pragma unroll
  for (int i = 0; i < kPhPer; ++i) {
    if (n0 + i < N) {
      run += (double)v[i];
      const float c = (float)run;                         // torch's CPU cumsum: accumulate in double, round per element
      const float tc = kTauF * c;                         // math.tau * cumsum      (one rounding)
      phase_out[(size_t)b * N + n0 + i] = __fdiv_rn(tc, sample_rate);   // ... / sample_rate (true division)
      if (f0_up_out) f0_up_out[(size_t)b * N + n0 + i] = v[i];
    }
  }
}

// ---- F.upsample(x, T*hop, mode="linear") on (rows, T) -> (rows, T*hop) (neural_waveshaping.py:75, shaping.py:69) -----------
__global__ __launch_bounds__(256) void g_upsample_kernel(const float* __restrict__ x, int T, int N, float scale, float* __restrict__ y) {
  const size_t row = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const GLerp L = g_lerp_coeff(n, T, scale);
  const float* xr = x + row * T;
  y[row * N + n] = fmaf(L.w0, xr[L.i0], L.w1 * xr[L.i1]);
}

// ---- oscillator bank (generators.py:58-66): out[b][k-1][n] = sin(fl(fl(k phase) + shift_k)) * [fl(f0 k) < sr/2] --------
__global__ __launch_bounds__(256) void g_oscillator_kernel(const float* __restrict__ f0_up, const float* __restrict__ phase,
                                                           const float* __restrict__ phase_u, const float* __restrict__ rand_phase,
                                                           int K, int N, float sample_rate, float* __restrict__ out) {
  const int b = blockIdx.z;
  const int k = blockIdx.y + 1;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const float shift = phase_u[k - 1] * rand_phase[k - 1] - kPiF;     // generators.py:54-56, two roundings
  const float kf = (float)k;
  const float f0n = f0_up[(size_t)b * N + n];
  const float arg = kf * phase[(size_t)b * N + n] + shift;            // two roundings (-ffp-contract=off)
  const float v = nws_sinf_nocall(arg);
  out[((size_t)b * K + (k - 1)) * N + n] = (f0n * kf) < sample_rate * 0.5f ? v : 0.0f;
}

// ---- Conv1d(Cin, Cout, 1) on (B, Cin, N): harmonic_mixer (neural_waveshaping.py:54), newt.mixer (shaping.py:63-65) ------
// thread = sample, blockIdx.y = output-channel tile of 8 (weights are wave-uniform: scalar loads), blockIdx.z = utterance
__global__ __launch_bounds__(256) void g_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, int Cin, int Cout, int N,
                                                        float* __restrict__ y) {
  const int b = blockIdx.z;
  const int o0 = blockIdx.y * 8;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
  const float* xp = x + (size_t)b * Cin * N + n;
  auto wrow = [&](int o) { return w + (size_t)(o < Cout ? o : Cout - 1) * Cin; };
  const float *w0 = wrow(o0), *w1 = wrow(o0 + 1), *w2 = wrow(o0 + 2), *w3 = wrow(o0 + 3), *w4 = wrow(o0 + 4),
              *w5 = wrow(o0 + 5), *w6 = wrow(o0 + 6), *w7 = wrow(o0 + 7);
  for (int c = 0; c < Cin; ++c) {
    const float xv = xp[(size_t)c * N];
    a0 = fmaf(w0[c], xv, a0);
    a1 = fmaf(w1[c], xv, a1);
    a2 = fmaf(w2[c], xv, a2);
    a3 = fmaf(w3[c], xv, a3);
    a4 = fmaf(w4[c], xv, a4);
    a5 = fmaf(w5[c], xv, a5);
    a6 = fmaf(w6[c], xv, a6);
    a7 = fmaf(w7[c], xv, a7);
  }
  float* yp = y + (size_t)b * Cout * N + n;
  auto put = [&](int o, float v) {
    if (o < Cout) yp[(size_t)o * N] = v + (bias ? bias[o] : 0.0f);
  };
  put(o0, a0);
  put(o0 + 1, a1);
  put(o0 + 2, a2);
  put(o0 + 3, a3);
  put(o0 + 4, a4);
  put(o0 + 5, a5);
  put(o0 + 6, a6);
  put(o0 + 7, a7);
}

// ---- shapers ------------------------------------------------------------------------------------------------------------
struct GShaper {
  int S, width, depth;          // TrainableNonlinearity(channels = S, width, depth)
  const float* in_scale;        // (S)
  const float* w[8];            // layer i: depth 1: (S); else i = 0: (S*width), 0 < i < depth-1: (S*width, width), last: (S, width)
  const float* b[8];
  const float* lut;             // (S, lut_size) or NULL
  int lut_size;
  float lut_min, lut_max;
};

// exact sin-MLP of one (sample, shaper): hidden activations in LDS columns hb[2][width][blockDim] (runtime width).
// Rounding chain of the reference (shaping.py:36-37): grouped Conv1d = bias + sum (sequential FMAs are within an ulp of
// ATen's order), torch.sin -> nws_sinf.
__device__ __forceinline__ float g_exact_shaper(const GShaper& P, int s, float x, float* hb, int tid, int nthreads) {
  float a = P.in_scale[s] * x;
  if (P.depth == 1) return nws_sinf_nocall(fmaf(P.w[0][s], a, P.b[0][s]));
  const int W = P.width;
  float* cur = hb;
  float* nxt = hb + (size_t)W * nthreads;
  for (int j = 0; j < W; ++j) cur[(size_t)j * nthreads + tid] = nws_sinf_nocall(fmaf(P.w[0][s * W + j], a, P.b[0][s * W + j]));
  for (int layer = 1; layer < P.depth - 1; ++layer) {
    const float* wl = P.w[layer] + (size_t)s * W * W;
    const float* bl = P.b[layer] + (size_t)s * W;
    for (int i = 0; i < W; ++i) {
      float acc = bl[i];
      for (int j = 0; j < W; ++j) acc = fmaf(wl[i * W + j], cur[(size_t)j * nthreads + tid], acc);
      nxt[(size_t)i * nthreads + tid] = nws_sinf_nocall(acc);
    }
    float* t = cur;
    cur = nxt;
    nxt = t;
  }
  const float* wl = P.w[P.depth - 1] + (size_t)s * W;
  float acc = P.b[P.depth - 1][s];
  for (int j = 0; j < W; ++j) acc = fmaf(wl[j], cur[(size_t)j * nthreads + tid], acc);
  return nws_sinf_nocall(acc);
}

// FastNEWT.shaping_fn (shaping.py:136-151), the reference's chain rounding for rounding (index scale `size`, clamped floor,
// fraction against the CLAMPED floor: linear extrapolation below the table, flat above)
__device__ __forceinline__ float g_lut_shaper(const GShaper& P, int s, float x) {
  const float t = (float)P.lut_size * (x - P.lut_min);
  const float idx = __fdiv_rn(t, P.lut_max - P.lut_min);
  float fl = floorf(idx);
  fl = fmaxf(fl, 0.0f);
  fl = fminf(fl, (float)(P.lut_size - 1));
  const int lo = (int)fl;
  const int up = lo + 1 < P.lut_size ? lo + 1 : P.lut_size - 1;
  const float fract = idx - fl;
  const float lv = P.lut[(size_t)s * P.lut_size + lo], uv = P.lut[(size_t)s * P.lut_size + up];
  return (uv - lv) * fract + lv;
}

// TrainableNonlinearity.forward / FastNEWT.shaping_fn on (B, S, N): y = shaper_s(x)
__global__ __launch_bounds__(128) void g_shaper_apply_kernel(GShaper P, const float* __restrict__ x, int64_t N,
                                                             float* __restrict__ y) {
  extern __shared__ float hb[];
  const int s = blockIdx.y % P.S;
  const size_t row = blockIdx.y;
  for (int64_t n = (int64_t)blockIdx.x * 128 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 128) {
    const float v = x[row * N + n];
    y[row * N + n] = P.lut ? g_lut_shaper(P, s, v) : g_exact_shaper(P, s, v, hb, threadIdx.x, 128);
  }
}

// FastNEWT._init_lookup_table (shaping.py:107-119): table[s][i] = shaper_s(linspace(min, max, size)[i]); torch.linspace's
// fp32 kernel: start + step * i for the first half, end - step * (size - 1 - i) for the second
__global__ __launch_bounds__(128) void g_shaper_table_kernel(GShaper P, int size, float tmin, float tmax,
                                                             float* __restrict__ table) {
  extern __shared__ float hb[];
  const int s = blockIdx.y;
  const int i = blockIdx.x * 128 + threadIdx.x;
  const float step = (tmax - tmin) / (float)(size - 1);
  const int half = size / 2;
  const float xv = i < half ? tmin + step * (float)i : tmax - step * (float)(size - 1 - i);
  const float v = g_exact_shaper(P, s, xv, hb, threadIdx.x, 128);     // every thread takes part (LDS columns), stores guarded
  if (i < size) table[(size_t)s * size + i] = v;
}

// the sin-MLP of one (sample, shaper) with the hidden activations in registers: compile-time width W (4 / 8 / 16), run-time
// depth; the shaper index is workgroup-uniform, so every weight is a scalar operand.  Sines as v_sin_f32(fract(x / 2 pi)) -
// the form of the fused sin-MLP kernel (exciter_newt.hip bank_sin; there the 1 / 2 pi sits in the weights): the one rounding of
// x / 2 pi is 6e-8 |x| / 2 pi turns, 4e-7 rad at |x| = 6 (sum |W| + |b| of the shipped checkpoints stays below 4), against
// 1.5e-7 for the polynomial nws_sinf that the table construction and the stand-alone shaper keep.
__device__ __forceinline__ float g_sin_mlp(float x) {
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(x * 0.15915493667125702f));
}
template <int W>
__device__ __forceinline__ float g_exact_shaper_reg(const GShaper& P, int s, float x) {
  const float a = P.in_scale[s] * x;
  float h[W], g[W];
  {
    const float* w0 = P.w[0] + (size_t)s * W;
    const float* b0 = P.b[0] + (size_t)s * W;
#pragma unroll
    for (int j = 0; j < W; ++j) h[j] = g_sin_mlp(fmaf(w0[j], a, b0[j]));
  }
  for (int layer = 1; layer < P.depth - 1; ++layer) {
    const float* wl = P.w[layer] + (size_t)s * W * W;
    const float* bl = P.b[layer] + (size_t)s * W;
#pragma unroll
    for (int i = 0; i < W; ++i) {
      float acc = bl[i];
#pragma unroll
      for (int j = 0; j < W; ++j) acc = fmaf(wl[i * W + j], h[j], acc);
      g[i] = g_sin_mlp(acc);
    }
#pragma unroll
    for (int i = 0; i < W; ++i) h[i] = g[i];
  }
  const float* wl = P.w[P.depth - 1] + (size_t)s * W;
  float acc = P.b[P.depth - 1][s];
#pragma unroll
  for (int j = 0; j < W; ++j) acc = fmaf(wl[j], h[j], acc);
  return g_sin_mlp(acc);
}

// NEWT.forward up to the mixer (shaping.py:68-76): film (B, 4S, T) channel-major, upsampled xhop on the fly;
// v[b][s][n] = g_norm * shaper_s(g_idx * e + b_idx) + b_norm   (FiLM: multiply then add, two roundings, dynamic.py:8)
// W = the sin-MLP's width when it is 4 / 8 / 16 and depth >= 2 (registers), 0 = any width (LDS columns) or table shapers
template <int W>
__global__ __launch_bounds__(128) void g_film_shaper_kernel(GShaper P, const float* __restrict__ exciter,
                                                            const float* __restrict__ film, int T, int N, float scale,
                                                            float* __restrict__ out) {
  extern __shared__ float hb[];
  const int b = blockIdx.z, s = blockIdx.y;
  const int S = P.S;
  const float* fb = film + (size_t)b * 4 * S * T;
  for (int n = blockIdx.x * 128 + threadIdx.x; n < N; n += gridDim.x * 128) {
    const GLerp lc = g_lerp_coeff(n, T, scale);
    auto lerp = [&](int ch) { return fmaf(lc.w0, fb[(size_t)ch * T + lc.i0], lc.w1 * fb[(size_t)ch * T + lc.i1]); };
    const float g_i = lerp(s), b_i = lerp(S + s), g_n = lerp(2 * S + s), b_n = lerp(3 * S + s);
    const float x = g_i * exciter[((size_t)b * S + s) * N + n] + b_i;
    float sh;
    if (W > 0) sh = g_exact_shaper_reg<(W > 0 ? W : 4)>(P, s, x);
    else sh = P.lut ? g_lut_shaper(P, s, x) : g_exact_shaper(P, s, x, hb, threadIdx.x, 128);
    out[((size_t)b * S + s) * N + n] = g_n * sh + b_n;
  }
}

// ---- oscillator bank + harmonic mixer + FiLM + shapers + NEWT mixer in ONE kernel (round 4) -----------------------------
// The stage kernels above materialise the oscillator bank (B, K, N) (26 MB per 4 s utterance at 101 harmonics), the exciter
// (B, S, N) and the shaped signal (B, S, N).  Inside a forward none of them is needed: thread = sample keeps the S mixer
// accumulators in registers (S rounded up to the compile-time bucket SB: padded weight columns are zero), walks the harmonics
// once (one sine per harmonic, the K x SB transposed mixer weights are wave-uniform: scalar loads), then runs the FiLM'ed
// shapers channel by channel and folds them into the <= 4 NEWT output channels.  The FiLM rows of the workgroup's frames sit
// in LDS.  Same summation orders and rounding chains as g_oscillator / g_conv1x1 / g_film_shaper (bit-identical results).
__global__ void g_mixer_t_kernel(const float* __restrict__ w, int S, int K, int SB, float* __restrict__ wt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // wt[k][s], s < SB
  if (i >= K * SB) return;
  const int k = i / SB, sidx = i - k * SB;
  wt[i] = sidx < S ? w[(size_t)sidx * K + k] : 0.0f;
}

// EXC_ONLY: stop at the exciter (B, S, N) = oscillator bank x mixer + bias; the sin-MLP shapers then run in g_film_shaper_kernel,
// whose 128-thread workgroups keep many more waves per CU than this kernel's LDS allows (measured at B = 64, default sizes,
// exact shapers: 43 ms with everything in here against 17.7 ms for the stage kernels)
template <int SB, bool EXC_ONLY>
__global__ __launch_bounds__(256) void g_exciter_newt_kernel(GShaper P, const float* __restrict__ f0_up, const float* __restrict__ phase,
                                                             const float* __restrict__ phase_u, const float* __restrict__ rand_phase,
                                                             const float* __restrict__ wt, const float* __restrict__ mixer_b,
                                                             const float* __restrict__ film, const float* __restrict__ out_w,
                                                             const float* __restrict__ out_b, int K, int T, int N, int hop, float scale,
                                                             float sample_rate, int OC, int nf, float* __restrict__ out) {
  extern __shared__ float lds[];     // shift[K] | film rows [4 S][nf]
  float* shift = lds;
  float* fl = lds + K;
  const int S = P.S;
  const int b = blockIdx.z, tid = threadIdx.x;
  const int n0 = blockIdx.x * 256;
  const int n = n0 + tid;
  const bool live = n < N;
  // frames the workgroup's samples interpolate between: [fa, fa + nf)
  const GLerp first = g_lerp_coeff(n0, T, scale);
  const int fa = first.i0;
  for (int k = tid; k < K; k += 256) shift[k] = phase_u[k] * rand_phase[k] - kPiF;     // generators.py:54-56, two roundings
  if (!EXC_ONLY) {
    const float* fb = film + (size_t)b * 4 * S * T;
    for (int i = tid; i < 4 * S * nf; i += 256) {
      const int ch = i / nf, f = i - ch * nf;
      fl[i] = fb[(size_t)ch * T + (fa + f < T ? fa + f : T - 1)];
    }
  }
  __syncthreads();
  const int nn = live ? n : N - 1;
  const float f0n = f0_up[(size_t)b * N + nn], ph = phase[(size_t)b * N + nn];
  float acc[SB];
#pragma unroll
  for (int c = 0; c < SB; ++c) acc[c] = 0.0f;
  const float nyq = sample_rate * 0.5f;
  for (int k = 1; k <= K; ++k) {
    const float kf = (float)k;
    const float arg = kf * ph + shift[k - 1];                       // two roundings (-ffp-contract=off)
    const float v = (f0n * kf) < nyq ? nws_sinf_nocall(arg) : 0.0f;
    const float* wr = wt + (size_t)(k - 1) * SB;
#pragma unroll
    for (int c = 0; c < SB; ++c) acc[c] = fmaf(wr[c], v, acc[c]);
  }
  if (EXC_ONLY) {
    if (live) {
#pragma unroll
      for (int c = 0; c < SB; ++c)
        if (c < S) out[((size_t)b * S + c) * N + n] = acc[c] + mixer_b[c];
    }
    return;
  }
  const GLerp lc = g_lerp_coeff(nn, T, scale);
  const int l0 = lc.i0 - fa, l1 = lc.i1 - fa;
  // (table shapers only on this route - the launcher sends the sin-MLP shapers through EXC_ONLY: their body is far too large to
  // unroll SB times, and a run-time channel index into the accumulators would be scratch memory)
  float o0 = 0.0f, o1 = 0.0f, o2 = 0.0f, o3 = 0.0f;
#pragma unroll
  for (int c = 0; c < SB; ++c) {
    if (c < S) {
      auto lerp = [&](int ch) { return fmaf(lc.w0, fl[ch * nf + l0], lc.w1 * fl[ch * nf + l1]); };
      const float g_i = lerp(c), b_i = lerp(S + c), g_n = lerp(2 * S + c), b_n = lerp(3 * S + c);
      const float e = acc[c] + mixer_b[c];
      const float x = g_i * e + b_i;
      const float v = g_n * g_lut_shaper(P, c, x) + b_n;
      o0 = fmaf(out_w[c], v, o0);
      if (OC > 1) o1 = fmaf(out_w[S + c], v, o1);
      if (OC > 2) o2 = fmaf(out_w[2 * S + c], v, o2);
      if (OC > 3) o3 = fmaf(out_w[3 * S + c], v, o3);
    }
  }
  if (live) {
    float* op = out + (size_t)b * OC * N + n;
    op[0] = o0 + out_b[0];
    if (OC > 1) op[(size_t)N] = o1 + out_b[1];
    if (OC > 2) op[(size_t)2 * N] = o2 + out_b[2];
    if (OC > 3) op[(size_t)3 * N] = o3 + out_b[3];
  }
}

// The same with the harmonic mixer on the matrix pipe (v_mfma_f32_32x32x2_f32: fp32 products and sums, any K / S <= 64): rows =
// 32 shapers (A = mixer weights, K-major in LDS, zero padded), columns = 32 samples (B = the oscillator bank: every lane
// computes ONE sine per MFMA - harmonic 2 s + lane / 32 + 1 of sample lane % 32 - so the bank is evaluated exactly once).  The
// accumulators come out lane = sample, register = shaper: FiLM, table shapers and the NEWT mixer run on them in place, the two
// lane halves (16 MT shapers each) meet in one cross-half add per output channel.  A wave walks `tpw` tiles of 32 samples; the
// 6464 FMAs per sample of the thread-per-sample kernel above are what this removes (1.17 -> see LABBOOK at B = 64, defaults).

// table (S, size) -> pairs (SBM, size): (v[i], v[min(i + 1, size - 1)] - v[i]) - the two gathers and the difference of
// FastNEWT.shaping_fn's interpolation (shaping.py:147-151) as one 8-byte load; rows beyond S are zero
__global__ void g_lut_pairs_kernel(const float* __restrict__ lut, int S, int size, int SBM, float2* __restrict__ pairs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= SBM * size) return;
  const int c = i / size, j = i - c * size;
  float2 v{0.0f, 0.0f};
  if (c < S) {
    const float lo = lut[(size_t)c * size + j], up = lut[(size_t)c * size + (j + 1 < size ? j + 1 : size - 1)];
    v = float2{lo, up - lo};
  }
  pairs[i] = v;
}

typedef _Float16 gf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 gf16x2 __attribute__((ext_vector_type(2)));

// Mixer weights (S, K) -> A fragments of v_mfma_f32_32x32x16_f16 as two fp16 terms (hi + lo = 22 bits), scaled by a power of
// two that brings max |w| into [1, 2) (any weight magnitude stays inside fp16's range; the kernel multiplies the sums by the
// exact inverse).  Layout [step][mt][hi | lo][lane] x 16 B: lane l holds row 32 mt + l % 32, harmonics 16 step + 8 (l / 32)
// + 0..7; zero beyond S / K.  scl[0] = scale, scl[1] = 1 / scale.  One workgroup (the table is 28 KB at the default sizes).
__global__ __launch_bounds__(256) void g_mixer_frag_kernel(const float* __restrict__ w, int S, int K, int MT, gf16x8* __restrict__ frag,
                                                           float* __restrict__ scl) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  float mx = 0.0f;
  for (int i = tid; i < S * K; i += 256) {
    const float a = fabsf(w[i]);
    mx = (a <= 3.0e38f && a > mx) ? a : mx;          // finite maximum (NaN / inf weights give NaN / inf sums whatever the scale)
  }
  red[tid] = mx;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] = fmaxf(red[tid], red[tid + off]);
    __syncthreads();
  }
  mx = red[0];
  int e = 1;
  if (mx > 0.0f) (void)frexpf(mx, &e);               // mx = f 2^e, f in [0.5, 1)
  const float scale = ldexpf(1.0f, 1 - e), inv = ldexpf(1.0f, e - 1);
  if (tid == 0) {
    scl[0] = scale;
    scl[1] = inv;
  }
  const int K16 = (K + 15) / 16;
  for (int idx = tid; idx < K16 * MT * 64; idx += 256) {
    const int lane = idx & 63, mt = (idx >> 6) % MT, st = idx / (64 * MT);
    const int row = 32 * mt + (lane & 31), k0 = 16 * st + 8 * (lane >> 5);
    gf16x8 hi, lo;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = (row < S && k0 + i < K) ? w[(size_t)row * K + k0 + i] * scale : 0.0f;
      const _Float16 h = (_Float16)v;
      hi[i] = h;
      lo[i] = (_Float16)(v - (float)h);
    }
    frag[((st * MT + mt) * 2 + 0) * 64 + lane] = hi;
    frag[((st * MT + mt) * 2 + 1) * 64 + lane] = lo;
  }
}

// fp32 pair -> fp16 (hi, lo) pairs: hi = the value cut to 11 significant bits (exact in fp16), lo = the rest rounded to fp16
__device__ __forceinline__ void g_split2(float a, float b, gf16x2& hi, gf16x2& lo) {
  const float ha = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, a) & 0xFFFFE000u);
  const float hb = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, b) & 0xFFFFE000u);
  hi = __builtin_convertvector(f32x2{ha, hb}, gf16x2);
  lo = __builtin_convertvector(f32x2{a - ha, b - hb}, gf16x2);
}

// one tile's mixer: acc (rows = shapers, columns = samples) += W x bank on v_mfma_f32_32x32x16_f16, both operands as two fp16
// terms (hi hi + hi lo + lo hi: 22 bits, fp32 accumulation).  Every lane evaluates the 8 harmonics 16 step + 8 (lane / 32) +
// 1..8 of its sample per step - the bank is computed exactly once.  WIDE = some lane's phase is beyond the v_sin_f32 reduction's
// range (minutes of audio).  (Harmonics beyond K meet zero weights.)
template <int MT, bool WIDE>
__device__ __forceinline__ void g_mix_tile(gfloat16 (&acc)[MT], const gf16x8* fr, const float* sp, int K16, int half, float f0n,
                                           float ph, float nyq) {
  for (int st = 0; st < K16; ++st) {
    const float4 sa = *reinterpret_cast<const float4*>(sp + 16 * st), sb = *reinterpret_cast<const float4*>(sp + 16 * st + 4);
    const float sh[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
    const float kf0 = (float)(16 * st + 8 * half + 1);
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float kf = kf0 + (float)i;
      const float arg = kf * ph + sh[i];                              // two roundings (-ffp-contract=off)
      const float sv = WIDE ? nws_sin_wide(arg) : nws_sin_turns(arg);
      v[i] = (f0n * kf) < nyq ? sv : 0.0f;
    }
    gf16x2 h0, l0, h1, l1, h2, l2, h3, l3;
    g_split2(v[0], v[1], h0, l0);
    g_split2(v[2], v[3], h1, l1);
    g_split2(v[4], v[5], h2, l2);
    g_split2(v[6], v[7], h3, l3);
    const gf16x8 bhi = {h0.x, h0.y, h1.x, h1.y, h2.x, h2.y, h3.x, h3.y};
    const gf16x8 blo = {l0.x, l0.y, l1.x, l1.y, l2.x, l2.y, l3.x, l3.y};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const gf16x8 ahi = fr[((st * MT + mt) * 2 + 0) * 64], alo = fr[((st * MT + mt) * 2 + 1) * 64];
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[mt], 0, 0, 0);
      acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[mt], 0, 0, 0);
    }
  }
}

// OCT = NEWT output channels rounded up to 1 / 2 / 4 (0: stop at the exciter (B, S, N), the sin-MLP shapers follow in
// g_film_shaper_kernel).  rdiv = fl(1 / (lut_max - lut_min)): the table index t / (max - min) as q0 = t * rdiv, one residual
// and one correction FMA (Markstein's sequence: the correctly rounded quotient whenever rdiv is the correctly rounded
// reciprocal - three instructions instead of the v_div_scale / v_div_fmas / v_div_fixup chain).  FiLM rows sit in LDS as
// (x[f], x[f + 1] - x[f]) so that an interpolated parameter is one FMA (within an ulp of upsample_linear1d's w0 x0 + w1 x1).
template <int MT, int OCT>
__global__ __launch_bounds__(256, 2) void g_exciter_newt_mfma_kernel(GShaper P, const float* __restrict__ f0_up,
                                                                     const float* __restrict__ phase, const float* __restrict__ phase_u,
                                                                     const float* __restrict__ rand_phase, const gf16x8* __restrict__ frag,
                                                                     const float* __restrict__ scl,
                                                                     const float* __restrict__ mixer_b, const float* __restrict__ film,
                                                                     const float* __restrict__ out_w, const float* __restrict__ out_b,
                                                                     const float2* __restrict__ pairs, int K, int T, int N, float scale,
                                                                     float sample_rate, int OC, int nf, int tpw, float rdiv,
                                                                     float* __restrict__ out) {
  constexpr int SBM = 32 * MT;
  constexpr bool EXC_ONLY = OCT == 0;
  extern __shared__ float lds[];
  const int K16 = (K + 15) / 16;
  gf16x8* wl = reinterpret_cast<gf16x8*>(lds);            // [K16][MT][hi | lo][64] mixer fragments
  float* shift = lds + K16 * MT * 2 * 64 * 4;             // [16 K16]
  float* mb = shift + 16 * K16;     // [SBM] mixer bias x scale
  float* ow = mb + SBM;             // [4][SBM] NEWT mixer (zero rows / columns beyond OC / S)
  float2* fl = reinterpret_cast<float2*>(ow + 4 * SBM);   // [nf][4][SBM] FiLM rows of the workgroup's frames (zero beyond S)
  const int S = P.S;
  const int b = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * (128 * tpw);
  const int fa = g_lerp_coeff(n0, T, scale).i0;
  for (int i = tid; i < K16 * MT * 2 * 64; i += 256) wl[i] = frag[i];
  for (int k = tid; k < 16 * K16; k += 256) shift[k] = k < K ? phase_u[k] * rand_phase[k] - kPiF : 0.0f;     // generators.py:54-56
  const float wscale = scl[0], winv = scl[1];
  for (int c = tid; c < SBM; c += 256) mb[c] = c < S ? mixer_b[c] * wscale : 0.0f;
  if (!EXC_ONLY) {
    for (int i = tid; i < 4 * SBM; i += 256) {
      const int j = i / SBM, c = i - j * SBM;
      ow[i] = (j < OC && c < S) ? out_w[j * S + c] : 0.0f;
    }
    const float* fb = film + (size_t)b * 4 * S * T;
    for (int i = tid; i < nf * 4 * SBM; i += 256) {
      const int f = i / (4 * SBM), rc = i - f * (4 * SBM), row = rc / SBM, c = rc - row * SBM;
      float2 v{0.0f, 0.0f};
      if (c < S) {
        const float* src = fb + (size_t)(row * S + c) * T;
        const int t0 = fa + f < T ? fa + f : T - 1, t1 = t0 + 1 < T ? t0 + 1 : T - 1;
        const float x0 = src[t0];
        v = float2{x0, src[t1] - x0};
      }
      fl[i] = v;
    }
  }
  __syncthreads();
  const float nyq = sample_rate * 0.5f;
  const float lsize = (float)P.lut_size, ltop = (float)(P.lut_size - 1), ldiv = P.lut_max - P.lut_min;
  for (int tile = 0; tile < tpw; ++tile) {
    const int nt = n0 + (wave * tpw + tile) * 32;
    if (nt >= N) break;
    const int n = nt + col;
    const bool live = n < N;
    const int nn = live ? n : N - 1;
    const float f0n = f0_up[(size_t)b * N + nn], ph = phase[(size_t)b * N + nn];
    // the lane's shapers: 32 mt + (r & 3) + 8 (r >> 2) + 4 half.  hsel is opaque so that the per-shaper addresses are NOT
    // hoisted out of the tile loop as loop invariants (4 x 32 of them: registers); they are base + immediate instead
    int hsel = 4 * half;
    asm volatile("" : "+v"(hsel));
    const float* mbp = mb + hsel;
    gfloat16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = mbp[32 * mt + (r & 3) + 8 * (r >> 2)];      // the mixer bias starts the sum
    const bool wide = __any(fabsf(ph) * (float)(16 * K16 + 1) + 8.0f > 6.0e6f);
    if (__builtin_expect(wide, 0))
      g_mix_tile<MT, true>(acc, wl + lane, shift + 8 * half, K16, half, f0n, ph, nyq);
    else
      g_mix_tile<MT, false>(acc, wl + lane, shift + 8 * half, K16, half, f0n, ph, nyq);
    if (EXC_ONLY) {
      if (live) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * mt + (r & 3) + 8 * (r >> 2) + hsel;
            if (c < S) out[((size_t)b * S + c) * N + n] = acc[mt][r] * winv;
          }
      }
      continue;
    }
    const GLerp lc = g_lerp_coeff(nn, T, scale);
    // (i1 == i0 only at the clamped last frame, where the staged difference is zero as well)
    const float2* fp = fl + (lc.i0 - fa) * (4 * SBM) + hsel;
    const float* owp = ow + hsel;
    const unsigned hrow = (unsigned)hsel * (unsigned)P.lut_size;
    float o[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // (no branch on shaper < S: padded shapers have zero FiLM rows, zero table rows and zero NEWT-mixer weights, so a
        // tile's table gathers stay in one basic block, in flight together)
        const int cb = 32 * mt + (r & 3) + 8 * (r >> 2);
        auto lerp = [&](int row) {
          const float2 q = fp[row * SBM + cb];
          return fmaf(lc.w1, q.y, q.x);
        };
        const float g_i = lerp(0), b_i = lerp(1), g_n = lerp(2), b_n = lerp(3);
        const float x = g_i * (acc[mt][r] * winv) + b_i;
        // FastNEWT.shaping_fn (shaping.py:136-151), g_lut_shaper's chain with the quotient as above
        const float t = lsize * (x - P.lut_min);
        const float q0 = t * rdiv;
        const float idx = fmaf(fmaf(-ldiv, q0, t), rdiv, q0);
        const float fi = __builtin_amdgcn_fmed3f(floorf(idx), 0.0f, ltop);
        const float2 pr = (pairs + (size_t)cb * P.lut_size)[hrow + (unsigned)(int)fi];
        const float sh = pr.y * (idx - fi) + pr.x;
        const float v = g_n * sh + b_n;
#pragma unroll
        for (int j = 0; j < OCT; ++j) o[j] = fmaf(owp[j * SBM + cb], v, o[j]);
        if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);     // eight shapers' gathers in flight at a time (registers)
      }
#pragma unroll
    for (int j = 0; j < OCT; ++j) o[j] += __shfl_xor(o[j], 32);
    if (live && half == 0) {
      float* op = out + (size_t)b * OC * N + n;
#pragma unroll
      for (int j = 0; j < OCT; ++j)
        if (j < OC) op[(size_t)j * N] = o[j] + out_b[j];
    }
  }
}

// ---- FIR noise (generators.py:21-35) for any ir_length L / hop --------------------------------------------------------
// zero-phase FIR design: fir[b][t][n] = window[n] * roll(irfft(H[b, :, t]), L/2)[n]; irfft of a real half-spectrum as a
// cosine sum (cos table of L entries in LDS, built in double)
__global__ __launch_bounds__(256) void g_fir_design_kernel(const float* __restrict__ H, const float* __restrict__ window, int L,
                                                           int T, float* __restrict__ fir) {
  extern __shared__ float lds[];   // cos[L] | Hs[L/2 + 1]
  float* ct = lds;
  float* hs = lds + L;
  const int b = blockIdx.y, t = blockIdx.x;
  const int nb = L / 2 + 1;
  for (int i = threadIdx.x; i < L; i += 256) ct[i] = (float)cospi(2.0 * (double)i / (double)L);
  for (int k = threadIdx.x; k < nb; k += 256) hs[k] = H[((size_t)b * nb + k) * T + t];
  __syncthreads();
  const float inv = 1.0f / (float)L;
  for (int n = threadIdx.x; n < L; n += 256) {
    const int m = (n + L - L / 2) % L;                       // roll(h, L/2): h_rolled[n] = h[(n - L/2) mod L]
    float acc = 0.0f;
    int idx = m % L;                                         // (k m) mod L, advanced incrementally
    for (int k = 1; k < L / 2; ++k) {
      acc = fmaf(hs[k], ct[idx], acc);
      idx += m;
      if (idx >= L) idx -= L;
    }
    float v = hs[0] + 2.0f * acc;
    v += (m & 1) ? -hs[L / 2] : hs[L / 2];
    fir[((size_t)b * T + t) * L + n] = window[n] * (v * inv);
  }
}

// The same as ONE GEMM on the matrix pipe: fir (B T x L) = H^T (B T x L/2+1) x D, D[k][n] = window[n] c_k cos(2 pi k m(n) / L) / L
// (c_0 = c_{L/2} = 1, else 2; m(n) = (n - L/2) mod L), built per call in double and rounded once.  v_mfma_f32_32x32x2_f32:
// rows = 32 frames of one utterance (A = the H tile, staged channel-major in LDS), columns = 32 taps (B = D rows from L2), each
// wave walks the K loop once for TWO column tiles; the accumulators leave as full 128 B segments of the frame-major tap rows.
__global__ void g_fir_dmat_kernel(const float* __restrict__ window, int L, float* __restrict__ D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int nb = L / 2 + 1;
  if (i >= nb * L) return;
  const int k = i / L, n = i - k * L;
  const int m = (n + L - L / 2) % L;
  const long long idx = ((long long)k * m) % L;
  const double c = (k == 0 || k == L / 2) ? 1.0 : 2.0;
  D[i] = (float)((double)window[n] * c * cospi(2.0 * (double)idx / (double)L) / (double)L);
}

__global__ __launch_bounds__(256) void g_fir_design_mfma_kernel(const float* __restrict__ H, const float* __restrict__ D, int L,
                                                                int T, float* __restrict__ fir) {
  extern __shared__ float hs[];          // [nb + 1][33]: hs[k][frame], one zero row of padding for odd nb
  const int nb = L / 2 + 1, nbp = (nb + 1) & ~1;
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const int f = threadIdx.x & 31, cg = threadIdx.x >> 5;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
  for (int k = cg; k < nbp; k += 8) hs[k * 33 + f] = (k < nb && t0 + f < T) ? H[((size_t)b * nb + k) * T + t0 + f] : 0.0f;
  __syncthreads();
  const int ntile = (L + 31) / 32;
  for (int nt = 2 * wave; nt < ntile; nt += 8) {
    const int n0 = nt * 32 + col, n1 = n0 + 32;
    const bool two = nt + 1 < ntile;
    const float* d0 = D + (n0 < L ? n0 : L - 1);
    const float* d1 = D + (n1 < L ? n1 : L - 1);
    gfloat16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] = a1[r] = 0.0f;
#pragma unroll 4
    for (int k2 = 0; k2 < nbp; k2 += 2) {
      const int k = k2 + half;
      const int kc = k < nb ? k : nb - 1;                 // (the padded row of hs is zero)
      const float hv = hs[k * 33 + col];
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(hv, d0[(size_t)kc * L], a0, 0, 0, 0);
      if (two) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(hv, d1[(size_t)kc * L], a1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t = t0 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (t < T) {
        float* row = fir + ((size_t)b * T + t) * L;
        if (n0 < L) row[n0] = a0[r];
        if (two && n1 < L) row[n1] = a1[r];
      }
    }
  }
}

// rectangular-window STFT of the shared noise (reflect-padded by L/2), per-frame L-point CIRCULAR convolution with the
// frame's taps, overlap-add divided by the number of covering frames (torch.istft with window = ones), first hop*T samples;
// out = noise branch + sum over channels of add_in (B, O, N) (the cat + sum(1) of neural_waveshaping.py:85-86)
__global__ __launch_bounds__(256) void g_fir_noise_kernel(const float* __restrict__ fir, const float* __restrict__ noise, int M,
                                                          int L, int hop, int T, const float* __restrict__ add_in, int O,
                                                          int N, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  int t_hi = n / hop;
  if (t_hi > T - 1) t_hi = T - 1;
  float acc = 0.0f;
  int count = 0;
  for (int t = t_hi; t >= 0 && hop * t + L > n; --t) {
    const int j = n - hop * t;                    // position inside frame t
    const float* h = fir + ((size_t)b * T + t) * L;
    const int base = hop * t - L / 2;             // frame_t[i] = noise[reflect(base + i)]
    float y = 0.0f;
    for (int k = 0; k < L; ++k) {
      int i = j - k;
      if (i < 0) i += L;
      int p = base + i;
      if (p < 0) p = -p;
      if (p >= M) p = 2 * (M - 1) - p;
      y = fmaf(h[k], noise[p], y);
    }
    acc += y;
    ++count;
  }
  float v = acc / (float)count;
  if (add_in)
    for (int o = 0; o < O; ++o) v += add_in[((size_t)b * O + o) * N + n];
  out[(size_t)b * N + n] = v;
}

// The same on the matrix pipe.  Every utterance is filtered against the SAME noise frames (generators.py:30), so for an output
// hop block h the sum over its covering frames t and taps k is one GEMM: rows = 32 utterances (A = taps[b][t][k]), columns =
// 32 samples of the block (B = circulant of frame t: frame_t[(j - k) mod L], read from a doubled, reversed copy of the frame in
// LDS so that consecutive k are consecutive addresses), accumulated over (t, k) in fp32 (v_mfma_f32_32x32x2_f32: products and
// sums at fp32, any L / hop).  One workgroup per (hop block, 32 utterances); its four waves take the block's 32-sample column
// tiles.  Columns of a frame past its L samples (L not a multiple of hop) are masked.
constexpr int kFirTilesPerWave = 4;     // hop <= 4 waves x 4 tiles x 32 columns
__global__ __launch_bounds__(256) void g_fir_noise_mfma_kernel(const float* __restrict__ fir, const float* __restrict__ noise,
                                                               int M, int L, int hop, int T, int B,
                                                               const float* __restrict__ add_in, int O, int N,
                                                               float* __restrict__ out) {
  extern __shared__ float lds[];        // taps[32][L + 1] | rv[2 L]: rv[i] = frame[(-i) mod L]
  float* taps = lds;
  float* rv = lds + 32 * (L + 1);
  const int h = blockIdx.x, b0 = blockIdx.y * 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int ntile = (hop + 31) >> 5;
  gfloat16 acc[kFirTilesPerWave];
#pragma unroll
  for (int i = 0; i < kFirTilesPerWave; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
  for (int t = h; t >= 0 && hop * t + L > hop * h; --t) {
    __syncthreads();                    // the previous frame's operands are no longer read
    for (int r = wave; r < 32; r += 4) {
      const int b = b0 + r;
      const float* src = fir + ((size_t)(b < B ? b : B - 1) * T + t) * L;
      for (int k = lane; k < L; k += 64) taps[r * (L + 1) + k] = b < B ? src[k] : 0.0f;
    }
    const int base = hop * t - L / 2;   // frame_t[i] = noise[reflect(base + i)]
    for (int i = threadIdx.x; i < L; i += 256) {
      int p = base + i;
      if (p < 0) p = -p;
      if (p >= M) p = 2 * (M - 1) - p;
      const float v = noise[p];
      const int q = i == 0 ? 0 : L - i;
      rv[q] = v;
      rv[q + L] = v;
    }
    __syncthreads();
    const int joff = hop * (h - t);     // position of the block's first sample inside frame t
    const float* ap = taps + col * (L + 1) + half;
#pragma unroll
    for (int i = 0; i < kFirTilesPerWave; ++i) {
      const int tile = wave + 4 * i;
      if (tile < ntile) {
        const int j = joff + tile * 32 + col;
        if (joff + tile * 32 < L) {     // wave-uniform: the tile has at least one live column
          const bool live = j < L;
          const float* bp = rv + (L - (live ? j : L - 1)) + half;       // rv[k - j + L], k = 2 s + half
          gfloat16 a = acc[i];
#pragma unroll 8
          for (int s = 0; s < L / 2; ++s) {
            const float av = ap[2 * s];
            float bv = bp[2 * s];
            bv = live ? bv : 0.0f;
            a = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, a, 0, 0, 0);
          }
          acc[i] = a;
        }
      }
    }
  }
  // epilogue: divide by the number of covering frames, add the other branch's channels, store (lane = sample, register = row)
#pragma unroll
  for (int i = 0; i < kFirTilesPerWave; ++i) {
    const int tile = wave + 4 * i;
    const int jj = tile * 32 + col;
    if (tile < ntile && jj < hop) {
      const int n = hop * h + jj;
      int count = 0;
      for (int t = h; t >= 0 && hop * t + L > n; --t) ++count;
      const float inv = (float)count;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int b = b0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (b < B) {
          float v = acc[i][r] / inv;
          if (add_in)
            for (int o = 0; o < O; ++o) v += add_in[((size_t)b * O + o) * N + n];
          out[(size_t)b * N + n] = v;
        }
      }
    }
  }
}

// ---- reverb, time-domain form for lengths the four-step FFT plan does not factor (shaping.py:161-173) -------------------
// y[n] = x[n] + sum_{m=1..ir_len} ir[m-1] * xz[(n - m) mod Lc], xz = x zero-padded to Lc = max(N, ir_len + 1)
__global__ __launch_bounds__(256) void g_reverb_direct_kernel(const float* __restrict__ x, const float* __restrict__ ir,
                                                              int ir_len, int N, int Lc, float* __restrict__ y) {
  extern __shared__ float tile[];     // 256 + 1024 x-samples | 1024 taps
  float* xs = tile;
  float* hs = tile + 1280;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * 256;
  const int n = n0 + threadIdx.x;
  const float* xb = x + (size_t)b * N;
  float acc = 0.0f;
  for (int m0 = 1; m0 <= ir_len; m0 += 1024) {
    // taps m0 .. m0+1023; x indices (n0 - m0 - 1023) .. (n0 + 255 - m0), circular over Lc
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) hs[i] = m0 + i <= ir_len ? ir[m0 + i - 1] : 0.0f;
    const int lo = n0 - m0 - 1023;
    for (int i = threadIdx.x; i < 1280; i += 256) {
      int p = (lo + i) % Lc;
      if (p < 0) p += Lc;
      xs[i] = p < N ? xb[p] : 0.0f;
    }
    __syncthreads();
    // tap m0 + i pairs with x index n - m0 - i = lo + (threadIdx.x + 1023 - i)
    for (int i = 0; i < 1024; ++i) acc = fmaf(hs[i], xs[threadIdx.x + 1023 - i], acc);
  }
  if (n < N) y[(size_t)b * N + n] = xb[n] + acc;
}

// ---- reverb at an ODD circular length: what the reference's rfft / irfft pair really returns there ----------------------
// Reverb.forward (shaping.py:161-173) multiplies torch.fft.rfft of the two length-Lo signals ((Lo + 1) / 2 bins) and calls
// torch.fft.irfft WITHOUT a length: the inverse then assumes an even signal of M = 2 (bins - 1) = Lo - 1 samples.  For even
// Lo that is the circular convolution; for odd Lo (a reverb of an odd number of samples, e.g. sr = 11025, or an odd N beyond
// it) it is
//     y[n] = x[n] + (1/M) [ Re Y_0 + 2 sum_{0<k<M/2} Re(Y_k e^{2 pi i n k / M}) + Re(Y_{M/2}) (-1)^n ],   Y_k = X_k H_k (DFTs of length Lo)
// - not a convolution, but what a drop-in has to return (found by the random-configuration sweep of tests/test_gpu_generic.py).
// Evaluated as written, in float64 (the reference's transforms are fp32: this is closer to their exact value than they are
// to each other across FFT libraries): O(Lo^2) per utterance, the slow-but-correct class of this file.
__global__ void g_odd_tables_kernel(int Lo, double2* __restrict__ twLo, double2* __restrict__ twM) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < Lo) {
    double s, c;
    sincospi(2.0 * (double)j / (double)Lo, &s, &c);
    twLo[j] = make_double2(c, s);
  }
  if (j < Lo - 1) {
    double s, c;
    sincospi(2.0 * (double)j / (double)(Lo - 1), &s, &c);
    twM[j] = make_double2(c, s);
  }
}
// X[row][k] = sum_n v[n] e^{-2 pi i n k / Lo}, k < K; v = x row (len samples), or [0, ir] when shift_one (ir_[0] = 0)
__global__ void g_odd_dft_kernel(const float* __restrict__ v, int len, int row_stride, int shift_one, int Lo, int K,
                                 const double2* __restrict__ twLo, double2* __restrict__ X) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;
  if (k >= K) return;
  const float* vr = v + (size_t)row * row_stride;
  double re = 0.0, im = 0.0;
  int idx = shift_one ? k % Lo : 0;          // n k mod Lo, advanced by k per sample
  for (int n = 0; n < len; ++n) {
    const double2 w = twLo[idx];
    const double a = (double)vr[n];
    re += a * w.x;
    im -= a * w.y;
    idx += k;
    if (idx >= Lo) idx -= Lo;
  }
  X[(size_t)row * K + k] = make_double2(re, im);
}
__global__ void g_odd_inverse_kernel(const float* __restrict__ x, const double2* __restrict__ X, const double2* __restrict__ H,
                                     int N, int M, int K, const double2* __restrict__ twM, float* __restrict__ y) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (n >= N) return;
  const double2* Xr = X + (size_t)b * K;
  double acc = 0.0;
  int idx = n % M;                            // n k mod M for k = 1
  const int step = idx;
  for (int k = 1; k < K - 1; ++k) {
    const double2 a = Xr[k], h = H[k], w = twM[idx];
    const double yr = a.x * h.x - a.y * h.y, yi = a.x * h.y + a.y * h.x;
    acc += yr * w.x - yi * w.y;
    idx += step;
    if (idx >= M) idx -= M;
  }
  const double y0 = Xr[0].x * H[0].x - Xr[0].y * H[0].y;
  const double yq = Xr[K - 1].x * H[K - 1].x - Xr[K - 1].y * H[K - 1].y;
  const double wet = (y0 + 2.0 * acc + ((n & 1) ? -yq : yq)) / (double)M;
  y[(size_t)b * N + n] = x[(size_t)b * N + n] + (float)wet;
}

bool shaper_ok(const NwsShaperDesc* d) {
  if (!d || d->n_shapers <= 0) return false;
  if (d->lut) return d->lut_size >= 2 && d->lut_max > d->lut_min;
  if (d->depth < 1 || d->depth > 8 || d->width < 1 || !d->in_scale) return false;
  for (int i = 0; i < d->depth; ++i)
    if (!d->w[i] || !d->b[i]) return false;
  return true;
}

GShaper to_dev(const NwsShaperDesc* d) {
  GShaper g{};
  g.S = d->n_shapers;
  g.width = d->width;
  g.depth = d->depth;
  g.in_scale = d->in_scale;
  for (int i = 0; i < 8; ++i) {
    g.w[i] = d->w[i];
    g.b[i] = d->b[i];
  }
  g.lut = d->lut;
  g.lut_size = d->lut_size;
  g.lut_min = d->lut_min;
  g.lut_max = d->lut_max;
  return g;
}

// LDS of the exact shapers: hb[2][width][128 threads]
size_t shaper_lds(const NwsShaperDesc* d) { return d->lut ? 16 : (size_t)2 * d->width * 128 * sizeof(float); }

int ensure_lds(const void* fn, unsigned long long& mask) {
  if (nws_first_use_on_device(mask)) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  return NWS_OK;
}

size_t al(size_t b) { return (b + 255) & ~size_t(255); }

}  // namespace

extern "C" {

size_t nws_g_gru_workspace_bytes(int hidden) { return hidden > 0 ? al((size_t)3 * hidden * hidden * sizeof(float)) : 0; }

int nws_g_gru(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* control, int B,
              int C_total, int C_in, int hidden, int T, const float* h0, float* out, float* hT, void* workspace,
              size_t workspace_bytes, void* stream) {
  if (!w_ih || !w_hh || !b_ih || !b_hh || !control || !out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T <= 0 || hidden <= 0 || C_in <= 0 || C_total < C_in) return NWS_ERR_BAD_ARG;
  if (workspace_bytes < nws_g_gru_workspace_bytes(hidden)) return NWS_ERR_WORKSPACE;
  const size_t lds = ((size_t)2 * hidden + C_in) * sizeof(float);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  // (measurement switches, read once: NWS_G_GRU_RUNTIME keeps the runtime-size recurrence for the default shape, NWS_G_GRU_L2 the
  // form that reads W_hh from L2 every step)
  static const bool env_runtime = getenv("NWS_G_GRU_RUNTIME") != nullptr, env_l2 = getenv("NWS_G_GRU_L2") != nullptr;
  if (hidden == 128 && C_in == 2 && !env_runtime) {
    // the reference's default recurrence (GRU(2 -> 128)) inside an otherwise non-default configuration: the fused path's
    // kernel (control_gru.hip: 0.44 us per step against 0.8 for the runtime-size recurrence below), same layouts
    NwsWeights w{};
    w.gru_w_ih = w_ih;
    w.gru_w_hh = w_hh;
    w.gru_b_ih = b_ih;
    w.gru_b_hh = b_hh;
    return nws_control_gru_state(&w, control, B, C_total, T, h0, out, hT, stream);
  }
  if (hidden <= 128 && !env_l2) {
    // W_hh in registers: four lanes per hidden unit (whole waves: units rounded up to 16)
    const int threads = 4 * ((hidden + 15) & ~15);
    const int kq = hidden <= 32 ? 8 : hidden <= 64 ? 16 : 32;
    const size_t qlds = ((size_t)2 * 4 * kq + 2 * C_in + (size_t)3 * hidden * C_in) * sizeof(float);
    if (qlds <= 64 * 1024) {
      switch (kq) {
        case 8: g_gru_q_kernel<8><<<B, threads, qlds, st>>>(w_ih, w_hh, b_ih, b_hh, control, C_total, C_in, hidden, T, h0, out, hT); break;
        case 16: g_gru_q_kernel<16><<<B, threads, qlds, st>>>(w_ih, w_hh, b_ih, b_hh, control, C_total, C_in, hidden, T, h0, out, hT); break;
        default: g_gru_q_kernel<32><<<B, threads, qlds, st>>>(w_ih, w_hh, b_ih, b_hh, control, C_total, C_in, hidden, T, h0, out, hT); break;
      }
      NWS_CHECK_LAUNCH();
      return NWS_OK;
    }
  }
  float* wt = static_cast<float*>(workspace);
  const int cells = 3 * hidden * hidden;
  g_transpose_kernel<<<(cells + 255) / 256, 256, 0, st>>>(w_hh, 3 * hidden, hidden, wt);
  NWS_CHECK_LAUNCH();
  static unsigned long long attr = 0;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(g_gru_kernel), attr)) return rc;
  g_gru_kernel<<<B, 256, lds, st>>>(w_ih, wt, b_ih, b_hh, control, C_total, C_in, hidden, T, h0, out, hT);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_bth_to_bht(const float* x, int B, int T, int H, float* y, void* stream) {
  if (!x || !y || B <= 0 || T <= 0 || H <= 0) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  g_bth_to_bht_kernel<<<dim3((T * H + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(x, T, H, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// partial: B x ceil(N / 4096) doubles of scratch for the two-pass form (NULL: stream-ordered allocation)
static size_t g_phase_partials(int B, long long N) { return (size_t)B * (size_t)((N + kPhChunk - 1) / kPhChunk); }
static int g_phase_impl(const float* f0, const float* f0_up, int B, int T, int hop, float sample_rate, float* f0_up_out,
                        float* phase_out, double* partial, void* stream) {
  if ((!f0) == (!f0_up) || !phase_out || B <= 0 || T <= 0 || hop <= 0 || !(sample_rate > 0.0f)) return NWS_ERR_BAD_ARG;
  const long long N = (long long)T * hop;     // f0_up given: T = N, hop = 1
  if (N > (1ll << 30)) return NWS_ERR_UNSUPPORTED;
  const float scale = (float)T / (float)N;
  hipStream_t st = (hipStream_t)stream;
  const long long G = (N + kPhChunk - 1) / kPhChunk;
  // (NWS_G_PHASE_SERIAL is re-read on every call on purpose - tests/test_gpu_generic.py flips it in-process; a getenv is ~0.1 us
  // of a call that launches two kernels)
  if (G < 2 || G > 65535 || B > 65535 || getenv("NWS_G_PHASE_SERIAL")) {      // short rows: one workgroup per utterance, one launch
    g_phase_kernel<<<B, 1024, 0, st>>>(f0, f0_up, T, (int)N, scale, sample_rate, f0_up_out, phase_out);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  double* P = partial;
  if (!P && (hipMallocAsync(reinterpret_cast<void**>(&P), g_phase_partials(B, N) * sizeof(double), st) != hipSuccess || !P))
    return NWS_ERR_WORKSPACE;
  g_phase_sum_kernel<<<dim3((unsigned)G, B), 256, 0, st>>>(f0, f0_up, T, (int)N, scale, P);
  g_phase_scan_kernel<<<dim3((unsigned)G, B), 256, 0, st>>>(f0, f0_up, T, (int)N, scale, sample_rate, P, f0_up_out, phase_out);
  const hipError_t e = hipGetLastError();
  if (!partial) (void)hipFreeAsync(P, st);
  return e == hipSuccess ? NWS_OK : (int)e;
}

int nws_g_phase(const float* f0, const float* f0_up, int B, int T, int hop, float sample_rate, float* f0_up_out,
                float* phase_out, void* stream) {
  return g_phase_impl(f0, f0_up, B, T, hop, sample_rate, f0_up_out, phase_out, nullptr, stream);
}

int nws_g_upsample(const float* x, int64_t rows, int T, int hop, float* y, void* stream) {
  if (!x || !y || rows <= 0 || T <= 0 || hop <= 0) return NWS_ERR_BAD_ARG;
  if (rows > 65535) return NWS_ERR_UNSUPPORTED;
  const long long N = (long long)T * hop;
  if (N > (1ll << 30)) return NWS_ERR_UNSUPPORTED;
  g_upsample_kernel<<<dim3((unsigned)((N + 255) / 256), (unsigned)rows), 256, 0, (hipStream_t)stream>>>(x, T, (int)N, (float)T / (float)N, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_oscillator(const float* f0_up, const float* phase, const float* phase_u, const float* rand_phase, int K, int B, int N,
                     float sample_rate, float* out, void* stream) {
  if (!f0_up || !phase || !phase_u || !rand_phase || !out || K <= 0 || B <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (K > 65535 || B > 65535) return NWS_ERR_UNSUPPORTED;
  g_oscillator_kernel<<<dim3((N + 255) / 256, K, B), 256, 0, (hipStream_t)stream>>>(f0_up, phase, phase_u, rand_phase, K, N,
                                                                                     sample_rate, out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_conv1x1(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int N, float* y, void* stream) {
  if (!x || !w || !y || B <= 0 || Cin <= 0 || Cout <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (B > 65535 || (Cout + 7) / 8 > 65535) return NWS_ERR_UNSUPPORTED;
  g_conv1x1_kernel<<<dim3((N + 255) / 256, (Cout + 7) / 8, B), 256, 0, (hipStream_t)stream>>>(x, w, bias, Cin, Cout, N, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_shaper_apply(const NwsShaperDesc* d, const float* x, int64_t rows, int64_t N, float* y, void* stream) {
  if (!shaper_ok(d) || !x || !y || rows <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (rows > 65535 || rows % d->n_shapers) return NWS_ERR_UNSUPPORTED;
  const size_t lds = shaper_lds(d);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;
  static unsigned long long attr = 0;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(g_shaper_apply_kernel), attr)) return rc;
  const int gx = (int)((N + 127) / 128 < 4096 ? (N + 127) / 128 : 4096);
  g_shaper_apply_kernel<<<dim3(gx, (unsigned)rows), 128, lds, (hipStream_t)stream>>>(to_dev(d), x, N, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_shaper_table(const NwsShaperDesc* d, int table_size, float table_min, float table_max, float* table_out, void* stream) {
  if (!shaper_ok(d) || d->lut || !table_out || table_size < 2 || !(table_max > table_min)) return NWS_ERR_BAD_ARG;
  if (d->n_shapers > 65535) return NWS_ERR_UNSUPPORTED;
  const size_t lds = shaper_lds(d);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;
  static unsigned long long attr = 0;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(g_shaper_table_kernel), attr)) return rc;
  g_shaper_table_kernel<<<dim3((table_size + 127) / 128, d->n_shapers), 128, lds, (hipStream_t)stream>>>(
      to_dev(d), table_size, table_min, table_max, table_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_film_shaper(const NwsShaperDesc* d, const float* exciter, const float* film, int B, int T, int hop, float* out,
                      void* stream) {
  if (!shaper_ok(d) || !exciter || !film || !out || B <= 0 || T <= 0 || hop <= 0) return NWS_ERR_BAD_ARG;
  if (B > 65535 || d->n_shapers > 65535) return NWS_ERR_UNSUPPORTED;
  const size_t lds = shaper_lds(d);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;
  const int N = T * hop;
  const int gx = (N + 127) / 128 < 2048 ? (N + 127) / 128 : 2048;
  const dim3 grid(gx, d->n_shapers, B);
  const int w = (!d->lut && d->depth >= 2) ? d->width : 0;
  if (w == 4)
    g_film_shaper_kernel<4><<<grid, 128, 0, (hipStream_t)stream>>>(to_dev(d), exciter, film, T, N, (float)T / (float)N, out);
  else if (w == 8)
    g_film_shaper_kernel<8><<<grid, 128, 0, (hipStream_t)stream>>>(to_dev(d), exciter, film, T, N, (float)T / (float)N, out);
  else if (w == 16)
    g_film_shaper_kernel<16><<<grid, 128, 0, (hipStream_t)stream>>>(to_dev(d), exciter, film, T, N, (float)T / (float)N, out);
  else {
    static unsigned long long attr = 0;
    if (int rc = ensure_lds(reinterpret_cast<const void*>(g_film_shaper_kernel<0>), attr)) return rc;
    g_film_shaper_kernel<0><<<grid, 128, lds, (hipStream_t)stream>>>(to_dev(d), exciter, film, T, N, (float)T / (float)N, out);
  }
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// dmat: (fir_len / 2 + 1) x fir_len floats of scratch for the design matrix (NULL: stream-ordered allocation)
static int g_fir_design_impl(const float* H, const float* window, int fir_len, int B, int T, float* fir_out, float* dmat,
                             void* stream) {
  if (!H || !window || !fir_out || B <= 0 || T <= 0 || fir_len < 2 || (fir_len & 1)) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  static const bool valu_only = [] { const char* e = getenv("NWS_G_FIR"); return e && !strcmp(e, "valu"); }();
  const int nb = fir_len / 2 + 1;
  const size_t lds_mfma = (size_t)(nb + 1) * 33 * sizeof(float);
  if (!valu_only && lds_mfma <= 160 * 1024) {
    float* D = dmat;
    if (!D && (hipMallocAsync(reinterpret_cast<void**>(&D), (size_t)nb * fir_len * sizeof(float), st) != hipSuccess || !D))
      return NWS_ERR_WORKSPACE;
    g_fir_dmat_kernel<<<(nb * fir_len + 255) / 256, 256, 0, st>>>(window, fir_len, D);
    static unsigned long long attr = 0;
    int rc = ensure_lds(reinterpret_cast<const void*>(g_fir_design_mfma_kernel), attr);
    if (rc == NWS_OK) g_fir_design_mfma_kernel<<<dim3((T + 31) / 32, B), 256, lds_mfma, st>>>(H, D, fir_len, T, fir_out);
    const hipError_t e = hipGetLastError();
    if (!dmat) (void)hipFreeAsync(D, st);
    if (rc != NWS_OK) return rc;
    return e == hipSuccess ? NWS_OK : (int)e;
  }
  const size_t lds = ((size_t)fir_len + fir_len / 2 + 1) * sizeof(float);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;
  static unsigned long long attr = 0;
  if (int rc = ensure_lds(reinterpret_cast<const void*>(g_fir_design_kernel), attr)) return rc;
  g_fir_design_kernel<<<dim3(T, B), 256, lds, st>>>(H, window, fir_len, T, fir_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_fir_design(const float* H, const float* window, int fir_len, int B, int T, float* fir_out, void* stream) {
  return g_fir_design_impl(H, window, fir_len, B, T, fir_out, nullptr, stream);
}

int nws_g_fir_noise(const float* fir, const float* noise, int fir_len, int hop, int B, int T, const float* add_in,
                    int add_channels, float* out, void* stream) {
  if (!fir || !noise || !out || B <= 0 || T <= 0 || hop <= 0 || fir_len < 2 || (fir_len & 1)) return NWS_ERR_BAD_ARG;
  if (add_in && add_channels <= 0) return NWS_ERR_BAD_ARG;
  // torch.istft needs every sample of the first hop*T covered (fir_len >= hop) and torch.stft's reflect padding needs
  // fir_len / 2 < hop*T - 1
  const long long N = (long long)T * hop;
  if (fir_len < hop || fir_len / 2 >= N - 1 || B > 65535) return NWS_ERR_UNSUPPORTED;
  // matrix-pipe form (rows = utterances, shared noise circulant) when the tap tile fits LDS and a wave's column tiles its
  // registers; NWS_G_FIR=valu keeps the per-sample kernel
  static const bool valu_only = [] { const char* e = getenv("NWS_G_FIR"); return e && !strcmp(e, "valu"); }();
  const size_t lds = ((size_t)32 * (fir_len + 1) + 2 * (size_t)fir_len) * sizeof(float);
  if (!valu_only && lds <= 160 * 1024 && hop <= 128 * kFirTilesPerWave) {
    static unsigned long long attr = 0;
    if (int rc = ensure_lds(reinterpret_cast<const void*>(g_fir_noise_mfma_kernel), attr)) return rc;
    g_fir_noise_mfma_kernel<<<dim3(T, (B + 31) / 32), 256, lds, (hipStream_t)stream>>>(fir, noise, (int)N - 1, fir_len, hop, T, B,
                                                                                       add_in, add_channels, (int)N, out);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  g_fir_noise_kernel<<<dim3((unsigned)((N + 255) / 256), B), 256, 0, (hipStream_t)stream>>>(fir, noise, (int)N - 1, fir_len, hop,
                                                                                            T, add_in, add_channels, (int)N, out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_g_reverb_direct(const float* x, const float* ir, int ir_len, int B, int N, float* y, void* stream) {
  if (!x || !ir || !y || ir_len <= 0 || B <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (B > 65535 || x == y) return NWS_ERR_UNSUPPORTED;
  const int Lc = N > ir_len + 1 ? N : ir_len + 1;
  if (Lc & 1) {
    // odd circular length: the reference's own (non-convolution) result, see g_odd_* above; scratch is stream-ordered
    if (Lc > (1 << 22) || Lc < 3) return NWS_ERR_UNSUPPORTED;   // O(Lc^2) in float64: seconds at 2^20, not a real-time path
    const int M = Lc - 1, K = M / 2 + 1;
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = ((size_t)Lc + M + K + (size_t)B * K) * sizeof(double2);
    double2* buf = nullptr;
    if (hipMallocAsync(reinterpret_cast<void**>(&buf), bytes, st) != hipSuccess || !buf) return NWS_ERR_WORKSPACE;
    double2 *twLo = buf, *twM = twLo + Lc, *H = twM + M, *X = H + K;
    g_odd_tables_kernel<<<(Lc + 255) / 256, 256, 0, st>>>(Lc, twLo, twM);
    g_odd_dft_kernel<<<dim3((K + 127) / 128, 1), 128, 0, st>>>(ir, ir_len, 0, 1, Lc, K, twLo, H);
    g_odd_dft_kernel<<<dim3((K + 127) / 128, B), 128, 0, st>>>(x, N, N, 0, Lc, K, twLo, X);
    g_odd_inverse_kernel<<<dim3((N + 127) / 128, B), 128, 0, st>>>(x, X, H, N, M, K, twM, y);
    const hipError_t e = hipGetLastError();
    (void)hipFreeAsync(buf, st);
    return e == hipSuccess ? NWS_OK : (int)e;
  }
  g_reverb_direct_kernel<<<dim3((N + 255) / 256, B), 256, (1280 + 1024) * sizeof(float), (hipStream_t)stream>>>(x, ir, ir_len, N,
                                                                                                               Lc, y);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// ---- whole forward for any configuration -----------------------------------------------------------------------------
static bool model_ok(const NwsGenericModel* m) {
  if (!m) return false;
  if (m->control_size < 1 || m->gru_hidden < 1 || m->embedding < 1 || m->n_harmonics < 1 || m->n_shapers < 1 || m->hop < 1)
    return false;
  if (m->newt_mlp_depth < 1 || m->newt_mlp_depth > 8 || m->hgen_depth < 1 || m->hgen_depth > 8 || m->hgen_hidden < 1) return false;
  if (m->fir_len < 2 || (m->fir_len & 1) || m->out_channels < 1 || m->ir_len < 1) return false;
  if (!m->gru_w_ih || !m->gru_w_hh || !m->gru_b_ih || !m->gru_b_hh || !m->proj_w || !m->proj_b || !m->mixer_w || !m->mixer_b ||
      !m->newt_out_w || !m->newt_out_b || !m->noise_window || !m->ir)
    return false;
  for (int i = 0; i < m->newt_mlp_depth; ++i)
    if (!m->newt_mlp_w[i] || !m->newt_mlp_b[i] || (i < m->newt_mlp_depth - 1 && (!m->newt_ln_g[i] || !m->newt_ln_b[i]))) return false;
  for (int i = 0; i < m->hgen_depth; ++i)
    if (!m->hgen_w[i] || !m->hgen_b[i] || (i < m->hgen_depth - 1 && (!m->hgen_ln_g[i] || !m->hgen_ln_b[i]))) return false;
  return shaper_ok(&m->shaper) && m->shaper.n_shapers == m->n_shapers;
}

// mixer fragments or transposed mixer (<= (K + 16) x 64) + scale + table pairs (64 x lut_size x 2) of g_exciter_newt_fused
static size_t g_tab_floats(const NwsGenericModel* m) {
  return ((size_t)m->n_harmonics + 16) * 64 + 4 + (m->shaper.lut ? (size_t)128 * m->shaper.lut_size : 0);
}

struct GArena {
  float *gru_bth, *gru_bht, *emb, *film, *H, *fir, *f0_up, *phase, *osc, *exciter, *shaped, *newt, *pre, *tab, *dmat;
  double* partial;
  void* gru_ws;
  size_t gru_ws_bytes;
  bool ok;
};

static GArena g_carve(const NwsGenericModel* m, int B, int T, void* ws, size_t bytes) {
  const size_t N = (size_t)T * m->hop;
  char* p = static_cast<char*>(ws);
  size_t left = bytes;
  bool ok = true;
  auto take = [&](size_t b) -> void* {
    b = al(b);
    if (b > left) {
      ok = false;
      return nullptr;
    }
    void* r = p;
    p += b;
    left -= b;
    return r;
  };
  auto fl = [&](size_t n) { return static_cast<float*>(take(n * sizeof(float))); };
  GArena a{};
  a.gru_ws_bytes = nws_g_gru_workspace_bytes(m->gru_hidden);
  a.gru_ws = take(a.gru_ws_bytes);
  a.gru_bth = fl((size_t)B * T * m->gru_hidden);
  a.gru_bht = fl((size_t)B * T * m->gru_hidden);
  a.emb = fl((size_t)B * T * m->embedding);
  a.film = fl((size_t)B * T * 4 * m->n_shapers);
  a.H = fl((size_t)B * T * (m->fir_len / 2 + 1));
  a.fir = fl((size_t)B * T * m->fir_len);
  a.f0_up = fl((size_t)B * N);
  a.phase = fl((size_t)B * N);
  a.osc = fl((size_t)B * m->n_harmonics * N);
  a.exciter = fl((size_t)B * m->n_shapers * N);
  a.shaped = a.osc;                                  // the oscillator bank is dead once the exciter exists
  if (m->n_shapers > m->n_harmonics) a.shaped = fl((size_t)B * m->n_shapers * N);
  a.newt = fl((size_t)B * m->out_channels * N);
  a.pre = fl((size_t)B * N);
  a.tab = fl(g_tab_floats(m));
  a.dmat = fl((size_t)(m->fir_len / 2 + 1) * m->fir_len);
  a.partial = static_cast<double*>(take(g_phase_partials(B, (long long)N) * sizeof(double)));
  a.ok = ok;
  return a;
}

// launch of g_exciter_newt_kernel; NWS_ERR_UNSUPPORTED when the sizes do not fit it (the caller runs the stage kernels).
// `scratch` (g_tab_floats: its own region of the arena) receives the transposed mixer and the table pairs.
static int g_exciter_newt_fused(const NwsGenericModel* m, const float* f0_up, const float* phase, const float* phase_u,
                                const float* rand_phase, const float* film, int B, int T, int N, float sample_rate, float* scratch,
                                float* newt_out, float* exciter_out, float* shaped, void* stream) {
  const int S = m->n_shapers, K = m->n_harmonics, OC = m->out_channels;
  if (S > 64 || OC > 4 || B > 65535 || getenv("NWS_G_STAGES")) return NWS_ERR_UNSUPPORTED;
  const bool exc_only = m->shaper.lut == nullptr;     // sin-MLP shapers: see EXC_ONLY
  hipStream_t st = (hipStream_t)stream;
  const float scale = (float)T / (float)N;
  const NwsShaperDesc* d = &m->shaper;
  const GShaper P = to_dev(d);
  static const bool valu_mixer = [] { const char* e = getenv("NWS_G_EXCITER"); return e && !strcmp(e, "valu"); }();
  if (!valu_mixer) {
    // matrix-pipe mixer: tiles of 32 samples, `tpw` per wave as long as the launch keeps >= 1024 workgroups
    const int MT = S <= 32 ? 1 : 2, SBM = 32 * MT, K16 = (K + 15) / 16;
    int tpw = 4;
    while (tpw > 1 && (long long)((N + 128 * tpw - 1) / (128 * tpw)) * B < 1024) tpw >>= 1;
    // ... and as long as the FiLM rows of the workgroup's frames fit LDS next to the fragments (short hops: many frames per tile)
    auto lds_of = [&](int t) {
      const size_t frames = (size_t)(128 * t / m->hop + 3);
      return ((size_t)K16 * MT * 512 + 16 * K16 + 5 * (size_t)SBM + (exc_only ? 0 : (size_t)8 * SBM * frames)) * sizeof(float);
    };
    // FOUR workgroups per CU when fewer tiles per wave allow it (120 registers permit 4 waves per SIMD; the fragments alone
    // are 28 KB at the default sizes: 44.7 KB = 3 workgroups with four tiles per wave, 40.6 KB = 4 with two), else two.
    // NWS_G_TILE_LDS=<bytes> moves the first limit (measurements; 81920 = the rule before)
    static const size_t lds_goal = [] { const char* e = getenv("NWS_G_TILE_LDS"); return e ? (size_t)atoi(e) : (size_t)40960; }();
    {
      int t4 = tpw;
      while (t4 > 1 && lds_of(t4) > lds_goal) t4 >>= 1;
      if (lds_of(t4) <= lds_goal) tpw = t4;
    }
    while (tpw > 1 && lds_of(tpw) > 80 * 1024) tpw >>= 1;     // two workgroups per CU
    const int nf = 128 * tpw / m->hop + 3;
    const size_t lds = lds_of(tpw);
    if (lds <= 160 * 1024) {
      // scratch: [fragments K16 x MT x 512 floats | scale, 1 / scale, pad | table pairs]
      gf16x8* frag = reinterpret_cast<gf16x8*>(scratch);
      float* scl = scratch + (size_t)K16 * MT * 512;
      g_mixer_frag_kernel<<<1, 256, 0, st>>>(m->mixer_w, S, K, MT, frag, scl);
      NWS_CHECK_LAUNCH();
      float2* pairs = reinterpret_cast<float2*>(scl + 4);
      if (!exc_only) {
        g_lut_pairs_kernel<<<(SBM * P.lut_size + 255) / 256, 256, 0, st>>>(P.lut, S, P.lut_size, SBM, pairs);
        NWS_CHECK_LAUNCH();
      }
      const dim3 grid((N + 128 * tpw - 1) / (128 * tpw), 1, B);
#define NWS_G_EM(MTV, OCTV)                                                                                                        \
  {                                                                                                                                \
    static unsigned long long attr = 0;                                                                                            \
    if (int rc = ensure_lds(reinterpret_cast<const void*>(g_exciter_newt_mfma_kernel<MTV, OCTV>), attr)) return rc;                \
    g_exciter_newt_mfma_kernel<MTV, OCTV><<<grid, 256, lds, st>>>(P, f0_up, phase, phase_u, rand_phase, frag, scl, m->mixer_b, film, \
                                                                  m->newt_out_w, m->newt_out_b, pairs, K, T, N, scale,             \
                                                                  sample_rate, OC, nf, tpw, rdiv, OCTV == 0 ? exciter_out : newt_out); \
  }
      const int oct = exc_only ? 0 : OC == 1 ? 1 : OC == 2 ? 2 : 4;
      const float rdiv = exc_only ? 0.0f : (float)(1.0 / (double)(P.lut_max - P.lut_min));
      if (MT == 1) {
        if (oct == 0) NWS_G_EM(1, 0) else if (oct == 1) NWS_G_EM(1, 1) else if (oct == 2) NWS_G_EM(1, 2) else NWS_G_EM(1, 4)
      } else {
        if (oct == 0) NWS_G_EM(2, 0) else if (oct == 1) NWS_G_EM(2, 1) else if (oct == 2) NWS_G_EM(2, 2) else NWS_G_EM(2, 4)
      }
#undef NWS_G_EM
      NWS_CHECK_LAUNCH();
      if (exc_only) {
        int rc = nws_g_film_shaper(&m->shaper, exciter_out, film, B, T, m->hop, shaped, stream);
        if (rc != NWS_OK) return rc;
        return nws_g_conv1x1(shaped, m->newt_out_w, m->newt_out_b, B, S, OC, N, newt_out, stream);
      }
      return NWS_OK;
    }
  }
  const int SB = S <= 8 ? 8 : S <= 16 ? 16 : S <= 32 ? 32 : 64;
  const int nf = 256 / m->hop + 3;                  // frames 256 consecutive samples can touch (+ the clamped right neighbour)
  const size_t lds = exc_only ? (size_t)K * sizeof(float) : ((size_t)K + (size_t)4 * S * nf) * sizeof(float);
  if (lds > 160 * 1024) return NWS_ERR_UNSUPPORTED;
  g_mixer_t_kernel<<<(K * SB + 255) / 256, 256, 0, st>>>(m->mixer_w, S, K, SB, scratch);
  NWS_CHECK_LAUNCH();
  const dim3 grid((N + 255) / 256, 1, B);
#define NWS_G_EN(SBV)                                                                                                              \
  {                                                                                                                                \
    static unsigned long long attr = 0;                                                                                            \
    if (int rc = ensure_lds(reinterpret_cast<const void*>(g_exciter_newt_kernel<SBV, false>), attr)) return rc;                    \
    if (exc_only)                                                                                                                  \
      g_exciter_newt_kernel<SBV, true><<<grid, 256, lds, st>>>(P, f0_up, phase, phase_u, rand_phase, scratch, m->mixer_b, film,    \
                                                                 m->newt_out_w, m->newt_out_b, K, T, N, m->hop, scale, sample_rate, \
                                                                 OC, nf, exciter_out);                                             \
    else                                                                                                                           \
      g_exciter_newt_kernel<SBV, false><<<grid, 256, lds, st>>>(P, f0_up, phase, phase_u, rand_phase, scratch, m->mixer_b, film,   \
                                                                  m->newt_out_w, m->newt_out_b, K, T, N, m->hop, scale,            \
                                                                  sample_rate, OC, nf, newt_out);                                  \
  }
  if (SB == 8) NWS_G_EN(8)
  else if (SB == 16) NWS_G_EN(16)
  else if (SB == 32) NWS_G_EN(32)
  else NWS_G_EN(64)
#undef NWS_G_EN
  NWS_CHECK_LAUNCH();
  if (exc_only) {
    // (`shaped` may alias the scratch: the transposed mixer is dead once the exciter kernel above has run)
    int rc = nws_g_film_shaper(&m->shaper, exciter_out, film, B, T, m->hop, shaped, stream);
    if (rc != NWS_OK) return rc;
    return nws_g_conv1x1(shaped, m->newt_out_w, m->newt_out_b, B, S, OC, N, newt_out, stream);
  }
  return NWS_OK;
}

size_t nws_forward_generic_workspace_bytes(const NwsGenericModel* m, int B, int T) {
  if (!m || B <= 0 || T <= 0 || m->hop <= 0) return 0;
  const size_t N = (size_t)T * m->hop;
  auto fb = [](size_t n) { return al(n * sizeof(float)); };
  size_t t = nws_g_gru_workspace_bytes(m->gru_hidden);
  t += 2 * fb((size_t)B * T * m->gru_hidden) + fb((size_t)B * T * m->embedding) + fb((size_t)B * T * 4 * m->n_shapers);
  t += fb((size_t)B * T * (m->fir_len / 2 + 1)) + fb((size_t)B * T * m->fir_len) + 2 * fb((size_t)B * N);
  t += fb((size_t)B * m->n_harmonics * N) + fb((size_t)B * m->n_shapers * N);
  if (m->n_shapers > m->n_harmonics) t += fb((size_t)B * m->n_shapers * N);
  t += fb((size_t)B * m->out_channels * N) + fb((size_t)B * N) + fb(g_tab_floats(m));
  t += fb((size_t)(m->fir_len / 2 + 1) * m->fir_len) + al(g_phase_partials(B, (long long)N) * sizeof(double));
  return t;
}

int nws_forward_generic(const NwsGenericModel* m, const float* f0, const float* control, int B, int C, int T, float sample_rate,
                        const float* phase_u, const float* rand_phase, const float* noise, const NwsReverbPlan* plan,
                        const void* reverb_tables, const void* reverb_spectrum, void* reverb_workspace,
                        size_t reverb_workspace_bytes, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!model_ok(m) || !f0 || !control || !phase_u || !rand_phase || !noise || !out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2 || C < m->control_size) return NWS_ERR_BAD_ARG;
  const GArena a = g_carve(m, B, T, workspace, workspace_bytes);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  const int N = T * m->hop;
  const int S = m->n_shapers, K = m->n_harmonics, nb = m->fir_len / 2 + 1;
  int rc;
#define G(call)                  \
  do {                           \
    rc = (call);                 \
    if (rc != NWS_OK) return rc; \
  } while (0)
  // control encoder (neural_waveshaping.py:69-72, :24-26)
  G(nws_g_gru(m->gru_w_ih, m->gru_w_hh, m->gru_b_ih, m->gru_b_hh, control, B, C, m->control_size, m->gru_hidden, T, nullptr,
              a.gru_bth, nullptr, a.gru_ws, a.gru_ws_bytes, stream));
  G(nws_g_bth_to_bht(a.gru_bth, B, T, m->gru_hidden, a.gru_bht, stream));
  {
    const float* w1[1] = {m->proj_w};
    const float* b1[1] = {m->proj_b};
    const float* none[1] = {nullptr};
    G(nws_td_mlp(a.gru_bht, B, m->gru_hidden, m->embedding, m->embedding, 1, T, w1, b1, none, none, m->ln_eps, m->leaky_slope,
                 a.emb, stream));
  }
  // frame-rate MLPs (shaping.py:68, neural_waveshaping.py:82)
  G(nws_td_mlp(a.emb, B, m->embedding, m->embedding, 4 * S, m->newt_mlp_depth, T, m->newt_mlp_w, m->newt_mlp_b, m->newt_ln_g,
               m->newt_ln_b, m->ln_eps, m->leaky_slope, a.film, stream));
  G(nws_td_mlp(a.emb, B, m->embedding, m->hgen_hidden, nb, m->hgen_depth, T, m->hgen_w, m->hgen_b, m->hgen_ln_g, m->hgen_ln_b,
               m->ln_eps, m->leaky_slope, a.H, stream));
  // exciter (neural_waveshaping.py:75-76, :64-67)
  G(g_phase_impl(f0, nullptr, B, T, m->hop, sample_rate, a.f0_up, a.phase, a.partial, stream));
  // oscillator bank -> harmonic mixer -> FiLM / shapers -> NEWT mixer (generators.py:58-66, neural_waveshaping.py:64-67,
  // shaping.py:67-79): one kernel that keeps everything between the phase and the NEWT output in registers when the sizes allow
  // (<= 64 shapers, <= 4 output channels), the stage kernels otherwise
  rc = g_exciter_newt_fused(m, a.f0_up, a.phase, phase_u, rand_phase, a.film, B, T, N, sample_rate, a.tab, a.newt, a.exciter, a.shaped, stream);
  if (rc == NWS_ERR_UNSUPPORTED) {
    G(nws_g_oscillator(a.f0_up, a.phase, phase_u, rand_phase, K, B, N, sample_rate, a.osc, stream));
    G(nws_g_conv1x1(a.osc, m->mixer_w, m->mixer_b, B, K, S, N, a.exciter, stream));
    G(nws_g_film_shaper(&m->shaper, a.exciter, a.film, B, T, m->hop, a.shaped, stream));
    G(nws_g_conv1x1(a.shaped, m->newt_out_w, m->newt_out_b, B, S, m->out_channels, N, a.newt, stream));
  } else if (rc != NWS_OK) {
    return rc;
  }
  // noise branch + branch sum (generators.py:21-35, neural_waveshaping.py:82-86)
  G(g_fir_design_impl(a.H, m->noise_window, m->fir_len, B, T, a.fir, a.dmat, stream));
  G(nws_g_fir_noise(a.fir, noise, m->fir_len, m->hop, B, T, a.newt, m->out_channels, a.pre, stream));
  // reverb (shaping.py:161-173): four-step FFT when the circular length factors, time-domain form otherwise
  if (plan && reverb_tables && reverb_spectrum && reverb_workspace) {
    G(nws_reverb(plan, reverb_tables, reverb_spectrum, a.pre, B, N, out, reverb_workspace, reverb_workspace_bytes, stream));
  } else {
    G(nws_g_reverb_direct(a.pre, m->ir, m->ir_len, B, N, out, stream));
  }
#undef G
  return NWS_OK;
}

}  // extern "C"
