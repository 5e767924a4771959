// Learned reverb: y = x + circconv_L(x, [0, ir])[:N],  L = max(N, len(ir)+1)
// (Reverb.forward, models/modules/shaping.py:161-173; circular, NOT zero-padded to 2N: SURVEY App. D.4).
//
// Design: a hand-written four-step FFT of length L = N1 * N2 (N2 = largest power of
// two dividing L, <= 1024; L = 64000 -> 125 x 512, L = 32000 -> 125 x 256):
//   (1) column DFT over n1: for N1 = 125 (every standard size) three radix-5 Stockham stages in LDS on a
//       125 x 32-column tile; for any other N1 a dense fp32 MFMA contraction against a cached DFT matrix
//       (exact-fp32 v_mfma_f32_32x32x2_f32, the complex product written as a real [2N1 x 2N1] matrix so
//       that one MFMA step consumes (re, im) of one input row);
//   (2) twiddle + radix-2 Stockham row FFT of size N2 in LDS, pointwise product with the cached IR
//       spectrum, inverse row FFT, conjugate twiddle, 1/L  -- one workgroup per row, one kernel;
//   (3) inverse column DFT (same MFMA kernel, conjugate matrix) + dry signal, only for rows < N/N2.
// Two utterances are packed into one complex transform (z = x_a + i x_b; the IR is real, so
// Re/Im of the circular convolution are the two results): no Hermitian untangling anywhere.
// The spectrum is kept in the transform's own (k1,k2) order, which a pointwise product does not care about.
//
// Lengths the direct form does not serve well (N1 neither 125 nor <= 128: the DFT-matrix column pass is O(N1^2)) run as
// OVERLAP-SAVE on the same kernels: the input is read as an Lc-periodic signal (zeros between N and Lc), block j is an
// L = 125 * 2^k point transform that starts hist = len(ir) samples before output j * (L - hist); its last L - hist points are
// final samples of the length-Lc circular convolution.  The wrap-around lives in the column pass's loads; the row pass sees
// pairs * nblk independent transforms.  So every even Lc has a plan (Reverb.forward takes any N, shaping.py:161-173).
#include <stdlib.h>

#include "nws_common.h"

namespace {

__device__ __forceinline__ int frag_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

struct PlanDev {
  int L, N1, N2, NP, M2;
  int Lc, hist, nblk, P;   // overlap-save: circular length, history per block, blocks per pair, outputs per block (direct: L, 0, 1, L)
};

__host__ __device__ inline int round_up32(int v) { return (v + 31) & ~31; }

inline PlanDev plan_dev(const NwsReverbPlan* p) {
  PlanDev d;
  d.L = p->L;
  d.N1 = p->N1;
  d.N2 = p->N2;
  d.NP = round_up32(p->N1);
  d.M2 = 2 * d.NP;
  d.Lc = p->Lc > 0 ? p->Lc : p->L;
  d.hist = p->Lc > 0 ? p->hist : 0;
  d.nblk = p->Lc > 0 ? p->nblk : 1;
  d.P = d.L - d.hist;
  return d;
}

// table offsets (in floats)
inline size_t off_afwd(const PlanDev&) { return 0; }
inline size_t off_ainv(const PlanDev& d) { return (size_t)d.N1 * 2 * d.M2; }
inline size_t off_tw(const PlanDev& d) { return 2 * (size_t)d.N1 * 2 * d.M2; }
inline size_t off_rowtw(const PlanDev& d) { return off_tw(d) + 2 * (size_t)d.L; }
inline size_t off_tw125(const PlanDev& d) { return off_rowtw(d) + (size_t)d.N2; }
inline size_t off_r8(const PlanDev& d) { return off_tw125(d) + 250 + 6; }   // 16-byte aligned: every offset before it is even, 250 + 6 = 256
inline size_t table_floats(const PlanDev& d) { return off_r8(d) + 2 * 8 * 64 * 2; }

// A[(step*2 + h)*M2 + row]: the real form of the (inverse) DFT matrix, see header comment.
//   forward  F = cos - i sin :  real row k1: [cos, +sin]   imag row k1: [-sin, cos]
//   inverse  F = cos + i sin :  real row n1: [cos, -sin]   imag row n1: [+sin, cos]
__global__ void build_dft_matrix_kernel(float* __restrict__ A, int N1, int NP, int M2, int inverse) {
  const size_t total = (size_t)N1 * 2 * M2;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(e % M2);
    const int h = (int)((e / M2) & 1);
    const int step = (int)(e / (2 * (size_t)M2));
    const int is_im = row >= NP;
    const int idx = is_im ? row - NP : row;
    float v = 0.0f;
    if (idx < N1) {
      const long long m = ((long long)idx * step) % N1;
      double s, c;
      sincospi(2.0 * (double)m / (double)N1, &s, &c);
      if (inverse) s = -s;
      // forward: real row [c, s], imag row [-s, c]
      v = (float)(is_im ? (h == 0 ? -s : c) : (h == 0 ? c : s));
    }
    A[e] = v;
  }
}

__global__ void build_twiddle_kernel(float2* __restrict__ tw, float2* __restrict__ rowtw, float2* __restrict__ tw125,
                                     float2* __restrict__ r8, int L, int N1, int N2) {
  // inner twiddles of the wave-per-row 512-point transform (row512_kernel), lane-major so that a wave loads them coalesced:
  // r8[k][lane] = W512^(lane k),  r8[8 + k][lane] = W64^((lane & 7) k)
  if (blockIdx.x == 1 && N2 == 512) {
    for (int e = threadIdx.x; e < 2 * 8 * 64; e += blockDim.x) {
      const int which = e >> 9, k = (e >> 6) & 7, lane = e & 63;
      const int m = which == 0 ? (lane * k) & 511 : (8 * (lane & 7) * k) & 511;
      double s, c;
      sincospi(2.0 * (double)m / 512.0, &s, &c);
      r8[e] = make_float2((float)c, (float)-s);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 125) {
    double s, c;
    sincospi(2.0 * (double)threadIdx.x / 125.0, &s, &c);
    tw125[threadIdx.x] = make_float2((float)c, (float)-s);
  }
  const size_t total = (size_t)L + N2 / 2;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    double s, c;
    if (e < (size_t)L) {
      const int k1 = (int)(e / N2), n2 = (int)(e % N2);
      const long long m = ((long long)k1 * n2) % L;
      sincospi(2.0 * (double)m / (double)L, &s, &c);
      tw[e] = make_float2((float)c, (float)-s);
    } else {
      const int m = (int)(e - L);
      sincospi(2.0 * (double)m / (double)N2, &s, &c);
      rowtw[m] = make_float2((float)c, (float)-s);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// N1 == 125 (L = 64000, 32000, ...: every standard size): the column DFT as a real FFT, three radix-5
// Stockham stages in LDS on a 125 x 32-column tile (lanes = columns -> every LDS access is 64 consecutive
// float2, conflict-free; global traffic is whole 128 B row segments).  ~100x fewer flops than the GEMM form.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

template <bool INVERSE>
__device__ __forceinline__ void dft5(float2 (&v)[5]) {
  const float c1 = 0.30901699437494745f, c2 = -0.8090169943749473f;
  const float s1 = 0.9510565162951535f, s2 = 0.5877852522924731f;
  const float2 t1 = make_float2(v[1].x + v[4].x, v[1].y + v[4].y), t2 = make_float2(v[2].x + v[3].x, v[2].y + v[3].y);
  const float2 t3 = make_float2(v[1].x - v[4].x, v[1].y - v[4].y), t4 = make_float2(v[2].x - v[3].x, v[2].y - v[3].y);
  const float2 a1 = make_float2(fmaf(c2, t2.x, fmaf(c1, t1.x, v[0].x)), fmaf(c2, t2.y, fmaf(c1, t1.y, v[0].y)));
  const float2 a2 = make_float2(fmaf(c1, t2.x, fmaf(c2, t1.x, v[0].x)), fmaf(c1, t2.y, fmaf(c2, t1.y, v[0].y)));
  const float2 b1 = make_float2(fmaf(s2, t4.x, s1 * t3.x), fmaf(s2, t4.y, s1 * t3.y));
  const float2 b2 = make_float2(fmaf(-s1, t4.x, s2 * t3.x), fmaf(-s1, t4.y, s2 * t3.y));
  v[0] = make_float2(v[0].x + t1.x + t2.x, v[0].y + t1.y + t2.y);
  // forward: V1 = a1 - i b1, V4 = a1 + i b1, V2 = a2 - i b2, V3 = a2 + i b2   (-i (x+iy) = y - ix); inverse: conjugate
  const float sg = INVERSE ? -1.0f : 1.0f;
  v[1] = make_float2(a1.x + sg * b1.y, a1.y - sg * b1.x);
  v[4] = make_float2(a1.x - sg * b1.y, a1.y + sg * b1.x);
  v[2] = make_float2(a2.x + sg * b2.y, a2.y - sg * b2.x);
  v[3] = make_float2(a2.x - sg * b2.y, a2.y + sg * b2.x);
}

// three Stockham radix-5 stages over bufA -> bufB -> bufA -> bufB; returns with the result in bufB (natural order)
// kColThreads = 800 = 25 x 32: one radix-5 butterfly per thread and stage (25 butterflies x 32 columns), 5 elements per
// thread for the loads and stores - the tile's work divides exactly, and a CU holds 2 x 12.5 waves instead of 2 x 4 (the
// 256-thread form ran the 25 butterflies of a column on 8 thread groups: 4 rounds, the last one 1/8 occupied)
constexpr int kColThreads = 800;
constexpr int kColPer = 125 * 32 / kColThreads;   // 5
template <bool INVERSE>
__device__ __forceinline__ void fft125_tile(float2* bufA, float2* bufB, const float2* tw125, int tid) {
  const int c = tid & 31, g = tid >> 5;
  float2* in = bufA;
  float2* out = bufB;
#pragma unroll
  for (int Ns = 1; Ns < 125; Ns *= 5) {
    const int twstep = 25 / Ns;
    for (int j = g; j < 25; j += kColThreads / 32) {
      const int k = j % Ns;
      float2 v[5];
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        v[r] = in[(j + 25 * r) * 32 + c];
        if (Ns > 1 && r > 0) {
          float2 t = tw125[r * k * twstep];
          if (INVERSE) t.y = -t.y;
          v[r] = cmulf(v[r], t);
        }
      }
      dft5<INVERSE>(v);
      const int j0 = (j / Ns) * Ns * 5 + k;
#pragma unroll
      for (int r = 0; r < 5; ++r) out[(j0 + r * Ns) * 32 + c] = v[r];
    }
    __syncthreads();
    float2* t = in;
    in = out;
    out = t;
  }
}

template <bool OLS>
__global__ __launch_bounds__(kColThreads) void col125_fwd_kernel(PlanDev d, const float2* __restrict__ tw125_g,
                                                         const float* __restrict__ x, int B, int N, long long x_stride,
                                                         float* __restrict__ Ure, float* __restrict__ Uim) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* bufA = reinterpret_cast<float2*>(smem_raw);
  float2* bufB = bufA + 125 * 32;
  float2* tw125 = bufB + 125 * 32;
  const int tid = threadIdx.x;
  const int p = blockIdx.y;
  const int c0 = blockIdx.x * 32;
  if (tid < 125) tw125[tid] = tw125_g[tid];
  const float* x0 = 2 * p < B ? x + (size_t)(2 * p) * x_stride : nullptr;
  const float* x1 = 2 * p + 1 < B ? x + (size_t)(2 * p + 1) * x_stride : nullptr;
  const int blk = OLS ? blockIdx.z : 0;
  const size_t slot = (size_t)p * d.nblk + blk;
  // all loads issued before the first LDS write (memory-latency bound: bytes in flight are what counts)
  {
    float2 v[kColPer];
#pragma unroll
    for (int i = 0; i < kColPer; ++i) {
      const int e = tid + kColThreads * i;
      int n = d.N2 * (e >> 5) + c0 + (e & 31);
      if (OLS) {   // position n of block blk = sample (blk P - hist + n) of the Lc-periodic signal (N .. Lc-1: zeros)
        n += blk * d.P - d.hist;
        n = n < 0 ? n + d.Lc : (int)((unsigned)n % (unsigned)d.Lc);
      }
      const bool in = n < N;
      v[i] = make_float2((in && x0) ? x0[n] : 0.0f, (in && x1) ? x1[n] : 0.0f);
    }
#pragma unroll
    for (int i = 0; i < kColPer; ++i) bufA[tid + kColThreads * i] = v[i];
  }
  __syncthreads();
  fft125_tile<false>(bufA, bufB, tw125, tid);
#pragma unroll
  for (int i = 0; i < kColPer; ++i) {
    const int e = tid + kColThreads * i;
    const size_t o = (slot * 125 + (e >> 5)) * d.N2 + c0 + (e & 31);
    const float2 v = bufB[e];
    Ure[o] = v.x;
    Uim[o] = v.y;
  }
}

template <bool OLS>
__global__ __launch_bounds__(kColThreads) void col125_inv_kernel(PlanDev d, const float2* __restrict__ tw125_g,
                                                         const float* __restrict__ Ure, const float* __restrict__ Uim,
                                                         const float* __restrict__ x, int B, int N,
                                                         float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* bufA = reinterpret_cast<float2*>(smem_raw);
  float2* bufB = bufA + 125 * 32;
  float2* tw125 = bufB + 125 * 32;
  const int tid = threadIdx.x;
  const int p = blockIdx.y;
  const int c0 = blockIdx.x * 32;
  if (tid < 125) tw125[tid] = tw125_g[tid];
  const int blk = OLS ? blockIdx.z : 0;
  const size_t slot = (size_t)p * d.nblk + blk;
  {
    float2 v[kColPer];
#pragma unroll
    for (int i = 0; i < kColPer; ++i) {
      const int e = tid + kColThreads * i;
      const size_t o = (slot * 125 + (e >> 5)) * d.N2 + c0 + (e & 31);
      v[i] = make_float2(Ure[o], Uim[o]);
    }
#pragma unroll
    for (int i = 0; i < kColPer; ++i) bufA[tid + kColThreads * i] = v[i];
  }
  const int rows_out = (N + d.N2 - 1) / d.N2;
  const bool has1 = 2 * p + 1 < B;
  // the dry signal is independent of the transform: fetch it before the FFT so that its latency hides under it
  // (overlap-save: position i of block blk is output blk P + i - hist; the first hist positions are wrapped history)
  float dry0[kColPer], dry1[kColPer];
  long long nout[kColPer];
#pragma unroll
  for (int i = 0; i < kColPer; ++i) {
    const int e = tid + kColThreads * i;
    const int pos = d.N2 * (e >> 5) + c0 + (e & 31);
    const long long n = OLS ? (long long)blk * d.P + pos - d.hist : (long long)pos;
    const bool keep = OLS ? (pos >= d.hist && n < N) : (e < rows_out * 32 && n < N);
    nout[i] = keep ? n : -1;
    const bool in = x != nullptr && keep;
    const size_t o0 = (size_t)(2 * p) * N + (keep ? n : 0);
    dry0[i] = in ? x[o0] : 0.0f;
    dry1[i] = (in && has1) ? x[o0 + N] : 0.0f;
  }
  __syncthreads();
  fft125_tile<true>(bufA, bufB, tw125, tid);
#pragma unroll
  for (int i = 0; i < kColPer; ++i) {
    const int e = tid + kColThreads * i;
    if (nout[i] >= 0) {
      const float2 v = bufB[e];
      const size_t o0 = (size_t)(2 * p) * N + nout[i];
      y[o0] = dry0[i] + v.x;
      if (has1) y[o0 + N] = dry1[i] + v.y;
    }
  }
}

constexpr size_t kCol125Lds = (2 * 125 * 32 + 125) * sizeof(float2);

// ---- column DFT (forward): real utterances -> planar U[p][k1][n2] ----
__global__ __launch_bounds__(256) void col_fwd_kernel(const float* __restrict__ A, PlanDev d, const float* __restrict__ x,
                                                      int B, int N, long long x_stride, float* __restrict__ Ure,
                                                      float* __restrict__ Uim) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int mt = blockIdx.y * 4 + wave;
  if (mt >= d.M2 / 32) return;
  const int p = blockIdx.z;
  const int c = blockIdx.x * 32 + col;
  const int utt = 2 * p + half;
  const float* xs = utt < B ? x + (size_t)utt * x_stride : nullptr;
  const int rows_in = (N + d.N2 - 1) / d.N2 < d.N1 ? (N + d.N2 - 1) / d.N2 : d.N1;
  const float* a_ptr = A + (size_t)half * d.M2 + 32 * mt + col;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 4
  for (int n1 = 0; n1 < rows_in; ++n1) {
    const long long n = (long long)d.N2 * n1 + c;
    const float bv = (xs != nullptr && n < N) ? xs[n] : 0.0f;
    const float av = a_ptr[(size_t)n1 * 2 * d.M2];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * mt + frag_row(r, half);
    const int is_im = row >= d.NP;
    const int k1 = is_im ? row - d.NP : row;
    if (k1 < d.N1) {
      float* dst = is_im ? Uim : Ure;
      dst[((size_t)p * d.N1 + k1) * d.N2 + c] = acc[r];
    }
  }
}

// ---- inverse column DFT + dry signal: planar U'[p][k1][n2] -> y (B,N) ----
__global__ __launch_bounds__(256) void col_inv_kernel(const float* __restrict__ A, PlanDev d,
                                                      const float* __restrict__ Ure, const float* __restrict__ Uim,
                                                      const float* __restrict__ x, int B, int N, float* __restrict__ y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, col = lane & 31;
  const int rows_out = (N + d.N2 - 1) / d.N2;
  const int nt = (rows_out + 31) / 32;
  const int idx = blockIdx.y * 4 + wave;
  if (idx >= 2 * nt) return;
  const int mt = idx < nt ? idx : d.NP / 32 + (idx - nt);
  const int p = blockIdx.z;
  const int c = blockIdx.x * 32 + col;
  const float* src = (half == 0 ? Ure : Uim) + (size_t)p * d.N1 * d.N2 + c;
  const float* a_ptr = A + (size_t)half * d.M2 + 32 * mt + col;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 4
  for (int k1 = 0; k1 < d.N1; ++k1) {
    const float bv = src[(size_t)k1 * d.N2];
    const float av = a_ptr[(size_t)k1 * 2 * d.M2];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 32 * mt + frag_row(r, half);
    const int is_im = row >= d.NP;
    const int n1 = is_im ? row - d.NP : row;
    const int utt = 2 * p + is_im;
    const long long n = (long long)d.N2 * n1 + c;
    if (n1 < d.N1 && utt < B && n < N) {
      const size_t o = (size_t)utt * N + n;
      y[o] = (x != nullptr ? x[o] : 0.0f) + acc[r];
    }
  }
}

// ---- row pass: twiddle, FFT_N2, x IR spectrum, IFFT_N2, conj twiddle, 1/L (in place on planar U) ----
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}

// Stockham autosort FFT of one row in LDS.  Two radix-2 stages (p and 2p) are fused into one radix-4 stage: the same
// butterflies and twiddles, half the LDS round trips and workgroup barriers (N2 = 512: 4 radix-4 + 1 radix-2 stage
// instead of 9).  rowtw[m] = exp(-2 pi i m / N2), m < N2/2.
template <bool INVERSE>
__device__ __forceinline__ float2* stockham(float2* x, float2* y, const float2* __restrict__ rowtw, int N2, int tid,
                                            int nthreads) {
  const int t = N2 >> 1, q = N2 >> 2;
  int p = 1;
  for (; 4 * p <= N2; p <<= 2) {
    const int step1 = t / p, step2 = q / p;  // N2/(2p), N2/(4p)
    for (int i = tid; i < q; i += nthreads) {
      const int k = i & (p - 1);
      const int base = ((i - k) << 2) + k;
      float2 w = rowtw[k * step1], w2 = rowtw[k * step2];
      if (INVERSE) {
        w.y = -w.y;
        w2.y = -w2.y;
      }
      const float2 a = x[i], bq = x[i + q];
      const float2 c = cmul(w, x[i + t]), dq = cmul(w, x[i + t + q]);
      const float2 a0 = make_float2(a.x + c.x, a.y + c.y), a1 = make_float2(a.x - c.x, a.y - c.y);
      const float2 b0 = cmul(w2, make_float2(bq.x + dq.x, bq.y + dq.y));
      const float2 b1r = cmul(w2, make_float2(bq.x - dq.x, bq.y - dq.y));
      // times -i (forward) / +i (inverse)
      const float2 b1 = INVERSE ? make_float2(-b1r.y, b1r.x) : make_float2(b1r.y, -b1r.x);
      y[base] = make_float2(a0.x + b0.x, a0.y + b0.y);
      y[base + 2 * p] = make_float2(a0.x - b0.x, a0.y - b0.y);
      y[base + p] = make_float2(a1.x + b1.x, a1.y + b1.y);
      y[base + 3 * p] = make_float2(a1.x - b1.x, a1.y - b1.y);
    }
    __syncthreads();
    float2* tmp = x;
    x = y;
    y = tmp;
  }
  for (; p < N2; p <<= 1) {  // one radix-2 stage left when log2(N2) is odd
    const int tw_step = t / p;
    for (int i = tid; i < t; i += nthreads) {
      const int k = i & (p - 1);
      const int j = ((i - k) << 1) + k;
      float2 w = rowtw[k * tw_step];
      if (INVERSE) w.y = -w.y;
      const float2 u0 = x[i];
      const float2 u1 = cmul(w, x[i + t]);
      y[j] = make_float2(u0.x + u1.x, u0.y + u1.y);
      y[j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
    }
    __syncthreads();
    float2* tmp = x;
    x = y;
    y = tmp;
  }
  return x;
}

template <bool SPECTRUM_ONLY, int kMaxPer>
__global__ __launch_bounds__(256) void row_kernel(PlanDev d, float* __restrict__ Ure, float* __restrict__ Uim,
                                                  const float2* __restrict__ tw, const float2* __restrict__ rowtw_g,
                                                  const float* __restrict__ Hre, const float* __restrict__ Him,
                                                  float* __restrict__ Sre, float* __restrict__ Sim) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float2* buf0 = reinterpret_cast<float2*>(smem_raw);
  float2* buf1 = buf0 + d.N2;
  float2* rowtw = buf1 + d.N2;  // N2/2 entries
  const int tid = threadIdx.x;
  const int k1 = blockIdx.x;
  const int p = blockIdx.y;
  const size_t base = ((size_t)p * d.N1 + k1) * d.N2;
  const size_t hbase = (size_t)k1 * d.N2;
  // everything this row needs from memory is requested up front (N2 <= 256 kMaxPer elements per thread): the IR
  // spectrum is only used between the two transforms, its latency hides under the forward one
  float2 z[kMaxPer], t4[kMaxPer], h4[kMaxPer];
#pragma unroll
  for (int e = 0; e < kMaxPer; ++e) {
    const int i = tid + 256 * e;
    const bool in = i < d.N2;
    z[e] = in ? make_float2(Ure[base + i], Uim[base + i]) : make_float2(0.0f, 0.0f);
    t4[e] = in ? tw[hbase + i] : make_float2(0.0f, 0.0f);
    h4[e] = (!SPECTRUM_ONLY && in) ? make_float2(Hre[hbase + i], Him[hbase + i]) : make_float2(0.0f, 0.0f);
  }
  for (int i = tid; i < d.N2 / 2; i += 256) rowtw[i] = rowtw_g[i];
#pragma unroll
  for (int e = 0; e < kMaxPer; ++e)
    if (tid + 256 * e < d.N2) buf0[tid + 256 * e] = cmul(z[e], t4[e]);
  __syncthreads();
  float2* cur = stockham<false>(buf0, buf1, rowtw, d.N2, tid, 256);
  if (SPECTRUM_ONLY) {
    for (int i = tid; i < d.N2; i += 256) {
      Sre[hbase + i] = cur[i].x;
      Sim[hbase + i] = cur[i].y;
    }
    return;
  }
  float2* other = cur == buf0 ? buf1 : buf0;
#pragma unroll
  for (int e = 0; e < kMaxPer; ++e)
    if (tid + 256 * e < d.N2) cur[tid + 256 * e] = cmul(cur[tid + 256 * e], h4[e]);
  __syncthreads();
  cur = stockham<true>(cur, other, rowtw, d.N2, tid, 256);
  const float inv_l = 1.0f / (float)d.L;
#pragma unroll
  for (int e = 0; e < kMaxPer; ++e) {
    const int i = tid + 256 * e;
    if (i < d.N2) {
      const float2 tc = make_float2(t4[e].x, -t4[e].y);
      const float2 r = cmul(cur[i], tc);
      Ure[base + i] = r.x * inv_l;
      Uim[base + i] = r.y * inv_l;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// N2 == 512 (L = 64000: every 4 s batch): ONE WAVE PER ROW, eight points per lane, the row transform as 8 x 8 x 8.
//   n = 64 j + l (lane l holds j = 0..7: eight coalesced 256 B loads),  k = k1 + 8 (k2a + 8 k2b):
//     pass 1  Y[l][k1]  = W512^(l k1)  DFT8_j  x[64 j + l]                       (registers)
//     exchange: lane (k1, b) <- Y[8 a + b][k1], a = 0..7                         (LDS, row stride 72: conflict-free)
//     pass 2  Z[b][k2a] = W64^(b k2a)  DFT8_a  Y[8 a + b][k1]
//     exchange: lane (k1, k2a) <- Z[b][k2a], b = 0..7                            (LDS, Latin-square swizzle: conflict-free)
//     pass 3  X[k1 + 8 k2a + 64 k2b] = DFT8_b Z[b][k2a]
// The spectrum stays in that (lane, register) order - the pointwise product does not care, the cached IR spectrum is written by
// the same transform (SPECTRUM_ONLY) - and the inverse runs the three passes backwards with conjugate twiddles, so neither
// direction reorders anything.  No workgroup barrier anywhere (a wave's LDS traffic is ordered by itself); the 256-thread
// Stockham form above needed ten per row and ran at 1.9 TB/s (17.3 us for 32.8 MB at B = 64).  Scalar fp32 butterflies: the
// "times -i" of a packed complex type is exactly the operand swizzle the build refuses (LABBOOK.md "5.2").
// ---------------------------------------------------------------------------------------------
struct C8 {
  float re[8], im[8];
};

// in-place DFT of 8 points; INV: conjugate twiddles (no scaling)
template <bool INV>
__device__ __forceinline__ void dft8(C8& v) {
  const float kS = 0.70710678118654752f;
  float ar[8], ai[8];
  // radix-2 DIT: evens (0, 2, 4, 6) and odds (1, 3, 5, 7) as two DFT4
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    const float x0r = v.re[o], x0i = v.im[o], x1r = v.re[2 + o], x1i = v.im[2 + o];
    const float x2r = v.re[4 + o], x2i = v.im[4 + o], x3r = v.re[6 + o], x3i = v.im[6 + o];
    const float s0r = x0r + x2r, s0i = x0i + x2i, d0r = x0r - x2r, d0i = x0i - x2i;
    const float s1r = x1r + x3r, s1i = x1i + x3i, d1r = x1r - x3r, d1i = x1i - x3i;
    // forward: -i d1 = (d1i, -d1r);  inverse: +i d1 = (-d1i, d1r)
    const float tr = INV ? -d1i : d1i, ti = INV ? d1r : -d1r;
    ar[4 * o + 0] = s0r + s1r; ai[4 * o + 0] = s0i + s1i;
    ar[4 * o + 2] = s0r - s1r; ai[4 * o + 2] = s0i - s1i;
    ar[4 * o + 1] = d0r + tr;  ai[4 * o + 1] = d0i + ti;
    ar[4 * o + 3] = d0r - tr;  ai[4 * o + 3] = d0i - ti;
  }
  // odd half times W8^k: W8 = (1 - i)/sqrt2, W8^2 = -i, W8^3 = (-1 - i)/sqrt2 (conjugates for the inverse)
  float br[4], bi[4];
  br[0] = ar[4]; bi[0] = ai[4];
  if (!INV) {
    br[1] = (ar[5] + ai[5]) * kS;  bi[1] = (ai[5] - ar[5]) * kS;
    br[2] = ai[6];                 bi[2] = -ar[6];
    br[3] = (ai[7] - ar[7]) * kS;  bi[3] = -(ar[7] + ai[7]) * kS;
  } else {
    br[1] = (ar[5] - ai[5]) * kS;  bi[1] = (ai[5] + ar[5]) * kS;
    br[2] = -ai[6];                bi[2] = ar[6];
    br[3] = -(ar[7] + ai[7]) * kS; bi[3] = (ar[7] - ai[7]) * kS;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v.re[k] = ar[k] + br[k];     v.im[k] = ai[k] + bi[k];
    v.re[k + 4] = ar[k] - br[k]; v.im[k + 4] = ai[k] - bi[k];
  }
}

constexpr int kR8Stride = 72;                      // floats per k1 row of the first exchange (64 + 8: bank = 8 k1 + b)
constexpr int kR8Floats = 2 * 8 * kR8Stride;       // re | im planes of one wave

// address of Z[k1][b][k2a] in the second exchange: bank = 8 ((k1 + b) & 7) + ((b + k2a) & 7) is a bijection of the 64 lanes
// both for the writers (k1, b) of one k2a and for the readers (k1, k2a) of one b
__device__ __forceinline__ int r8_addr2(int k1, int b, int k2a) { return 64 * k1 + 8 * ((k1 + b) & 7) + ((b + k2a) & 7); }

template <bool SPECTRUM_ONLY>
__global__ __launch_bounds__(256) void row512_kernel(PlanDev d, int rows, float* __restrict__ Ure, float* __restrict__ Uim,
                                                     const float2* __restrict__ tw, const float2* __restrict__ r8,
                                                     const float* __restrict__ Hre, const float* __restrict__ Him,
                                                     float* __restrict__ Sre, float* __restrict__ Sim) {
  __shared__ float lds[4][kR8Floats];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;           // = p * N1 + k1
  if (row >= rows) return;
  const int k1row = row % d.N1;
  const size_t base = (size_t)row * 512;
  const size_t hbase = (size_t)k1row * 512;
  float* Lre = lds[wave];
  float* Lim = Lre + 8 * kR8Stride;
  const int hi3 = lane >> 3, lo3 = lane & 7;       // lane = (k1, b) resp. (k1, k2a) after the exchanges

  C8 z, t4, h;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    z.re[j] = Ure[base + lane + 64 * j];
    z.im[j] = Uim[base + lane + 64 * j];
    const float2 t = tw[hbase + lane + 64 * j];
    t4.re[j] = t.x;
    t4.im[j] = t.y;
  }
  if (!SPECTRUM_ONLY) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h.re[j] = Hre[hbase + lane + 64 * j];
      h.im[j] = Him[hbase + lane + 64 * j];
    }
  }
  // twiddles of the two inner steps, k = 1..7: W512^(lane k) and W64^(lo3 k), lane-major tables (coalesced loads)
  float w1r[8], w1i[8], w2r[8], w2i[8];
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    const float2 a = r8[k * 64 + lane], c = r8[(8 + k) * 64 + lane];
    w1r[k] = a.x; w1i[k] = a.y;
    w2r[k] = c.x; w2i[k] = c.y;
  }
  // four-step twiddle of the column pass (tw[k1][n2])
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float r = fmaf(z.re[j], t4.re[j], -(z.im[j] * t4.im[j])), i = fmaf(z.re[j], t4.im[j], z.im[j] * t4.re[j]);
    z.re[j] = r;
    z.im[j] = i;
  }

  auto twiddle = [](C8& v, const float (&wr)[8], const float (&wi)[8], bool conj) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const float c = wr[k], s = conj ? -wi[k] : wi[k];
      const float r = fmaf(v.re[k], c, -(v.im[k] * s)), i = fmaf(v.re[k], s, v.im[k] * c);
      v.re[k] = r;
      v.im[k] = i;
    }
  };
  // ---- forward ----
  dft8<false>(z);
  twiddle(z, w1r, w1i, false);
#pragma unroll
  for (int k = 0; k < 8; ++k) {          // writer lane l, register k1 = k
    Lre[k * kR8Stride + lane] = z.re[k];
    Lim[k * kR8Stride + lane] = z.im[k];
  }
#pragma unroll
  for (int a = 0; a < 8; ++a) {          // reader lane (k1 = hi3, b = lo3), register a
    z.re[a] = Lre[hi3 * kR8Stride + 8 * a + lo3];
    z.im[a] = Lim[hi3 * kR8Stride + 8 * a + lo3];
  }
  dft8<false>(z);
  twiddle(z, w2r, w2i, false);
#pragma unroll
  for (int k = 0; k < 8; ++k) {          // writer lane (k1, b), register k2a = k
    const int ad = r8_addr2(hi3, lo3, k);
    Lre[ad] = z.re[k];
    Lim[ad] = z.im[k];
  }
#pragma unroll
  for (int b = 0; b < 8; ++b) {          // reader lane (k1, k2a = lo3), register b
    const int ad = r8_addr2(hi3, b, lo3);
    z.re[b] = Lre[ad];
    z.im[b] = Lim[ad];
  }
  dft8<false>(z);
  if (SPECTRUM_ONLY) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      Sre[hbase + lane + 64 * j] = z.re[j];
      Sim[hbase + lane + 64 * j] = z.im[j];
    }
    return;
  }
  // ---- x IR spectrum (same (lane, register) order) ----
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float r = fmaf(z.re[j], h.re[j], -(z.im[j] * h.im[j])), i = fmaf(z.re[j], h.im[j], z.im[j] * h.re[j]);
    z.re[j] = r;
    z.im[j] = i;
  }
  // ---- inverse: the same three passes backwards ----
  dft8<true>(z);                          // over k2b -> b
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int ad = r8_addr2(hi3, b, lo3);
    Lre[ad] = z.re[b];
    Lim[ad] = z.im[b];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ad = r8_addr2(hi3, lo3, k);
    z.re[k] = Lre[ad];
    z.im[k] = Lim[ad];
  }
  twiddle(z, w2r, w2i, true);
  dft8<true>(z);                          // over k2a -> a
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    Lre[hi3 * kR8Stride + 8 * a + lo3] = z.re[a];
    Lim[hi3 * kR8Stride + 8 * a + lo3] = z.im[a];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    z.re[k] = Lre[k * kR8Stride + lane];
    z.im[k] = Lim[k * kR8Stride + lane];
  }
  twiddle(z, w1r, w1i, true);
  dft8<true>(z);                          // over k1 -> j
  const float inv_l = 1.0f / (float)d.L;
#pragma unroll
  for (int j = 0; j < 8; ++j) {           // conjugate four-step twiddle, 1/L
    const float r = fmaf(z.re[j], t4.re[j], z.im[j] * t4.im[j]), i = fmaf(z.im[j], t4.re[j], -(z.re[j] * t4.im[j]));
    Ure[base + lane + 64 * j] = r * inv_l;
    Uim[base + lane + 64 * j] = i * inv_l;
  }
}

// ir_ = [0, ir] (models/modules/shaping.py:162)
__global__ void build_ir_kernel(const float* __restrict__ ir, int ir_len, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i <= ir_len) out[i] = i == 0 ? 0.0f : ir[i - 1];
}

// streaming (linear) reverb: y = x + wet[0:M] + tail[0:M];  tail'[k] = tail[k+M] + wet[M+k]   (k < TL; tail beyond TL is 0)
__global__ void reverb_tail_kernel(const float* __restrict__ x, const float* __restrict__ wet, const float* __restrict__ tail_in,
                                   float* __restrict__ tail_out, float* __restrict__ y, int M, int L, int TL) {
  const int b = blockIdx.y;
  const float* w = wet + (size_t)b * L;
  const float* ti = tail_in + (size_t)b * TL;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M + TL; i += gridDim.x * blockDim.x) {
    if (i < M) {
      y[(size_t)b * M + i] = x[(size_t)b * M + i] + w[i] + (i < TL ? ti[i] : 0.0f);
    } else {
      const int k = i - M;
      tail_out[(size_t)b * TL + k] = (k + M < TL ? ti[k + M] : 0.0f) + (M + k < L ? w[M + k] : 0.0f);
    }
  }
}

// ---- short buffers: the circular convolution in the time domain ----
// y[n] = x[n] + sum_{k<N} x[k] irz[(n - k) mod L], irz = [0, ir] zero-padded to L (kept behind the spectrum).  For the
// short streaming buffers of scripts/time_buffer_sizes.py (N <= 1024 << L = 32000) that is N^2 MACs from LDS in ONE launch
// instead of three latency-bound FFT passes over 32000 points.  Four partial sums per thread; a direct fp32 sum of <= 1024
// products is at least as close to the reference's result as a fp32 FFT.
constexpr int kDirectMaxN = 1024;
__global__ __launch_bounds__(256) void reverb_direct_kernel(const float* __restrict__ x, const float* __restrict__ irz, int N,
                                                            int L, float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* xs = reinterpret_cast<float*>(smem_raw);   // N inputs
  float* hs = xs + N;                               // hs[d + N] = irz[d mod L] for the lags d = n - k in (-N, N)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int n = blockIdx.x * 256 + tid;
  for (int i = tid; i < N; i += 256) xs[i] = x[(size_t)b * N + i];
  for (int i = tid; i < 2 * N; i += 256) {
    const int d = i - N;
    hs[i] = irz[d >= 0 ? d : d + L];
  }
  __syncthreads();
  if (n >= N) return;
  float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
  const float* h = hs + n + N;   // h[-k] = irz[(n - k) mod L]
  int k = 0;
  for (; k + 3 < N; k += 4) {
    a0 = fmaf(xs[k], h[-k], a0);
    a1 = fmaf(xs[k + 1], h[-k - 1], a1);
    a2 = fmaf(xs[k + 2], h[-k - 2], a2);
    a3 = fmaf(xs[k + 3], h[-k - 3], a3);
  }
  for (; k < N; ++k) a0 = fmaf(xs[k], h[-k], a0);
  y[(size_t)b * N + n] = xs[n] + ((a0 + a1) + (a2 + a3));
}

// irz = [0, ir] zero-padded to L
__global__ void build_irz_kernel(const float* __restrict__ ir, int ir_len, int L, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L) out[i] = (i >= 1 && i <= ir_len) ? ir[i - 1] : 0.0f;
}

// row pass of `pairs` packed transforms: wave-per-row radix-8 kernel for N2 = 512, the Stockham workgroup-per-row form otherwise
template <bool SPECTRUM_ONLY>
void launch_rows(const PlanDev& d, int pairs, float* Ure, float* Uim, const float* t, const float* Sre_in, const float* Sim_in,
                        float* Sre_out, float* Sim_out, hipStream_t st) {
  const float2* tw = reinterpret_cast<const float2*>(t + off_tw(d));
  const float2* rowtw = reinterpret_cast<const float2*>(t + off_rowtw(d));
  if (d.N2 == 512) {
    const int rows = pairs * d.N1;
    row512_kernel<SPECTRUM_ONLY><<<(rows + 3) / 4, 256, 0, st>>>(d, rows, Ure, Uim, tw, reinterpret_cast<const float2*>(t + off_r8(d)), Sre_in, Sim_in,
                                                                Sre_out, Sim_out);
  } else {
    // `pairs` counts transforms (utterance pairs x overlap-save blocks): grid.y in slices of 65535
    for (int q0 = 0; q0 < pairs; q0 += 65535) {
      const int nq = pairs - q0 < 65535 ? pairs - q0 : 65535;
      float* ure = Ure + (size_t)q0 * d.L;
      float* uim = Uim + (size_t)q0 * d.L;
      const size_t lds = (size_t)(2 * d.N2 + d.N2 / 2) * sizeof(float2);
      if (d.N2 <= 1024)
        row_kernel<SPECTRUM_ONLY, 4><<<dim3(d.N1, nq), 256, lds, st>>>(d, ure, uim, tw, rowtw, Sre_in, Sim_in, Sre_out, Sim_out);
      else
        row_kernel<SPECTRUM_ONLY, 8><<<dim3(d.N1, nq), 256, lds, st>>>(d, ure, uim, tw, rowtw, Sre_in, Sim_in, Sre_out, Sim_out);
    }
  }
}

constexpr int kMaxN2 = 2048;          // row transform sizes: 32 .. 2048 (40 KB of LDS per row at 2048)
constexpr int kDenseMaxN1 = 128;      // DFT-matrix column pass: O(N1^2) per column, cheaper than overlap-save up to about here
constexpr int kDenseLastResort = 8192;

bool plan_ok(const NwsReverbPlan* p) {
  if (!(p && p->L > 0 && p->N1 > 0 && p->N2 >= 32 && p->N2 <= kMaxN2 && (p->N2 & (p->N2 - 1)) == 0 &&
        (long long)p->N1 * p->N2 == p->L))
    return false;
  if (p->Lc == 0) return p->hist == 0 && p->nblk == 1;
  return p->N1 == 125 && p->Lc > p->hist && p->hist >= 0 && p->hist < p->L && p->nblk >= 1 && p->nblk <= 65535 && (p->Lc & 1) == 0;
}

}  // namespace

extern "C" {

int nws_reverb_plan(int N, int ir_len_plus1, NwsReverbPlan* plan) {
  if (!plan || N <= 0 || ir_len_plus1 <= 0 || N > (1 << 30) || ir_len_plus1 > (1 << 30)) return NWS_ERR_BAD_ARG;
  const int Lc = N > ir_len_plus1 ? N : ir_len_plus1;
  *plan = NwsReverbPlan{};
  int n2 = 1;
  while ((Lc % (n2 * 2)) == 0 && n2 < kMaxN2) n2 *= 2;
  // a smaller row size that leaves exactly 125 columns (e.g. 128000 = 125 x 1024 although 2048 | 256000 = 125 x 2048 is fine too)
  for (int m = n2; m >= 32; m >>= 1)
    if (Lc / m == 125 && Lc % m == 0) n2 = m;
  const int n1 = n2 >= 32 ? Lc / n2 : 0;
  int dense_max = kDenseMaxN1;
  // (the two NWS_REVERB_* switches are re-read per call on purpose: tests and tools/reverb_lengths.py flip them in-process; the
  // plan is host-only and the Python engine caches it per length, so this is not on a per-forward path)
  if (const char* e = getenv("NWS_REVERB_DENSE_MAX")) dense_max = atoi(e);   // measurements only
  if (n2 >= 32 && (n1 == 125 || n1 <= dense_max)) {
    plan->L = Lc;
    plan->N1 = n1;
    plan->N2 = n2;
    plan->nblk = 1;
    return NWS_OK;
  }
  // overlap-save on 125 x 2^k transforms; NWS_REVERB_OLS_N2 pins the row size (measurements)
  if ((Lc & 1) == 0) {
    const int hist = ir_len_plus1 - 1;
    int forced = 0;
    if (const char* e = getenv("NWS_REVERB_OLS_N2")) forced = atoi(e);
    double best = 0.0;
    int best_n2 = 0;
    for (int m = 256; m <= kMaxN2; m <<= 1) {
      const long long Lt = 125LL * m;
      const long long P = Lt - hist;
      if (P < Lt / 4) continue;                       // at least a quarter of every transform is output
      const long long nb = (N + P - 1) / P;
      if (nb > 65535) continue;
      if (forced && m != forced) continue;
      // measured per-point cost of a transform (tools/reverb_lengths.py, B = 64): the wave-per-row row pass of N2 = 512 against
      // the workgroup-per-row Stockham form of the other sizes
      const double cost = (double)nb * (double)Lt * (m == 512 ? 1.0 : m <= 1024 ? 1.35 : 1.55);
      if (best_n2 == 0 || cost < best) {
        best = cost;
        best_n2 = m;
      }
    }
    if (best_n2) {
      const long long P = 125LL * best_n2 - hist;
      plan->L = 125 * best_n2;
      plan->N1 = 125;
      plan->N2 = best_n2;
      plan->Lc = Lc;
      plan->hist = hist;
      plan->nblk = (int)((N + P - 1) / P);
      return NWS_OK;
    }
  }
  // an impulse response too long for the largest block: the DFT-matrix column pass at any N1 it can hold
  if (n2 >= 32 && n1 <= kDenseLastResort) {
    plan->L = Lc;
    plan->N1 = n1;
    plan->N2 = n2;
    plan->nblk = 1;
    return NWS_OK;
  }
  return NWS_ERR_UNSUPPORTED;
}

int nws_reverb_plan_serves(const NwsReverbPlan* plan, int N, int ir_len_plus1) {
  if (!plan_ok(plan) || N <= 0) return 0;
  if (plan->Lc == 0) return N <= plan->L && (ir_len_plus1 <= 0 || ir_len_plus1 <= plan->L);
  const int lc = ir_len_plus1 > 0 ? (N > ir_len_plus1 ? N : ir_len_plus1) : (N > plan->hist + 1 ? N : plan->hist + 1);
  if (lc != plan->Lc) return 0;
  if (ir_len_plus1 > 0 && ir_len_plus1 - 1 != plan->hist) return 0;
  const long long P = (long long)plan->L - plan->hist;
  return (N + P - 1) / P <= plan->nblk;
}

size_t nws_reverb_table_bytes(const NwsReverbPlan* plan) {
  if (!plan_ok(plan)) return 0;
  return table_floats(plan_dev(plan)) * sizeof(float);
}

size_t nws_reverb_spectrum_bytes(const NwsReverbPlan* plan) {
  if (!plan_ok(plan)) return 0;
  return 3 * (size_t)plan->L * sizeof(float);   // Sre | Sim | irz (time domain, for the direct form of short buffers)
}

size_t nws_reverb_workspace_bytes(const NwsReverbPlan* plan, int B) {
  if (!plan_ok(plan) || B <= 0) return 0;
  const size_t pairs = (size_t)(B + 1) / 2;
  // planar U (2 * pairs * nblk * L) ; the IR-spectrum build needs 3 L (ir_, Ure, Uim)
  const size_t f = 2 * pairs * (size_t)plan->nblk * (size_t)plan->L;
  const size_t g = 3 * (size_t)plan->L;
  return (f > g ? f : g) * sizeof(float);
}

static int ensure_col125_attrs() {
  static unsigned long long attr_devices = 0;
  if (nws_first_use_on_device(attr_devices)) {
    const void* fns[4] = {reinterpret_cast<const void*>(col125_fwd_kernel<false>), reinterpret_cast<const void*>(col125_fwd_kernel<true>),
                          reinterpret_cast<const void*>(col125_inv_kernel<false>), reinterpret_cast<const void*>(col125_inv_kernel<true>)};
    for (const void* fn : fns) {
      const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCol125Lds);
      if (e != hipSuccess) return (int)e;
    }
  }
  return NWS_OK;
}

int nws_reverb_build_tables(const NwsReverbPlan* plan, void* tables, void* stream) {
  if (!plan_ok(plan) || !tables) return NWS_ERR_BAD_ARG;
  if (int rc = ensure_col125_attrs()) return rc;
  const PlanDev d = plan_dev(plan);
  float* t = static_cast<float*>(tables);
  hipStream_t st = (hipStream_t)stream;
  build_dft_matrix_kernel<<<1024, 256, 0, st>>>(t + off_afwd(d), d.N1, d.NP, d.M2, 0);
  NWS_CHECK_LAUNCH();
  build_dft_matrix_kernel<<<1024, 256, 0, st>>>(t + off_ainv(d), d.N1, d.NP, d.M2, 1);
  NWS_CHECK_LAUNCH();
  build_twiddle_kernel<<<512, 256, 0, st>>>(reinterpret_cast<float2*>(t + off_tw(d)),
                                            reinterpret_cast<float2*>(t + off_rowtw(d)),
                                            reinterpret_cast<float2*>(t + off_tw125(d)),
                                            reinterpret_cast<float2*>(t + off_r8(d)), d.L, d.N1, d.N2);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

static size_t row_lds_bytes(const PlanDev& d) { return (size_t)(2 * d.N2 + d.N2 / 2) * sizeof(float2); }

int nws_reverb_ir_spectrum(const NwsReverbPlan* plan, const void* tables, const float* ir, int ir_len, void* spectrum,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!plan_ok(plan) || !tables || !ir || !spectrum || !workspace || ir_len <= 0) return NWS_ERR_BAD_ARG;
  if (ir_len + 1 > plan->L) return NWS_ERR_BAD_ARG;
  if (plan->Lc > 0 && ir_len != plan->hist) return NWS_ERR_BAD_ARG;
  if (workspace_bytes < nws_reverb_workspace_bytes(plan, 1)) return NWS_ERR_WORKSPACE;
  PlanDev d = plan_dev(plan);
  d.Lc = d.L;   // the spectrum of [0, ir] zero-padded to the TRANSFORM length: one plain transform in either mode
  d.hist = 0;
  d.nblk = 1;
  d.P = d.L;
  const float* t = static_cast<const float*>(tables);
  hipStream_t st = (hipStream_t)stream;
  float* irp = static_cast<float*>(workspace);
  float* Ure = irp + d.L;
  float* Uim = Ure + d.L;
  float* Sre = static_cast<float*>(spectrum);
  float* Sim = Sre + d.L;
  build_ir_kernel<<<(ir_len + 1 + 255) / 256, 256, 0, st>>>(ir, ir_len, irp);
  NWS_CHECK_LAUNCH();
  build_irz_kernel<<<(d.L + 255) / 256, 256, 0, st>>>(ir, ir_len, d.L, Sim + d.L);
  NWS_CHECK_LAUNCH();
  if (d.N1 == 125) {
    col125_fwd_kernel<false><<<dim3(d.N2 / 32, 1), kColThreads, kCol125Lds, st>>>(d, reinterpret_cast<const float2*>(t + off_tw125(d)),
                                                                           irp, 1, ir_len + 1, 0, Ure, Uim);
  } else {
    const dim3 g1(d.N2 / 32, (d.M2 / 32 + 3) / 4, 1);
    col_fwd_kernel<<<g1, 256, 0, st>>>(t + off_afwd(d), d, irp, 1, ir_len + 1, 0, Ure, Uim);
  }
  NWS_CHECK_LAUNCH();
  launch_rows<true>(d, 1, Ure, Uim, t, nullptr, nullptr, Sre, Sim, st);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_reverb(const NwsReverbPlan* plan, const void* tables, const void* spectrum, const float* x, int B, int N,
               float* y, void* workspace, size_t workspace_bytes, void* stream) {
  if (!plan_ok(plan) || !tables || !spectrum || !x || !y || !workspace || B <= 0 || N <= 0) return NWS_ERR_BAD_ARG;
  if (!nws_reverb_plan_serves(plan, N, 0)) return NWS_ERR_BAD_ARG;
  if (workspace_bytes < nws_reverb_workspace_bytes(plan, B)) return NWS_ERR_WORKSPACE;
  const PlanDev d = plan_dev(plan);
  const int pairs = (B + 1) / 2;
  if (pairs > 65535) return NWS_ERR_UNSUPPORTED;
  const float* t = static_cast<const float*>(tables);
  hipStream_t st = (hipStream_t)stream;
  float* Ure = static_cast<float*>(workspace);
  float* Uim = Ure + (size_t)pairs * d.nblk * d.L;
  const float* Sre = static_cast<const float*>(spectrum);
  const float* Sim = Sre + d.L;
  const float2* tw125 = reinterpret_cast<const float2*>(t + off_tw125(d));
  if (plan->Lc > 0) {
    // overlap-save: nblk transforms per pair, only as many as N needs (a plan made for N serves shorter calls with the same Lc)
    PlanDev e = d;
    e.nblk = (int)(((long long)N + d.P - 1) / d.P);
    Uim = Ure + (size_t)pairs * e.nblk * d.L;
    col125_fwd_kernel<true><<<dim3(d.N2 / 32, pairs, e.nblk), kColThreads, kCol125Lds, st>>>(e, tw125, x, B, N, (long long)N, Ure, Uim);
    NWS_CHECK_LAUNCH();
    launch_rows<false>(e, pairs * e.nblk, Ure, Uim, t, Sre, Sim, nullptr, nullptr, st);
    NWS_CHECK_LAUNCH();
    col125_inv_kernel<true><<<dim3(d.N2 / 32, pairs, e.nblk), kColThreads, kCol125Lds, st>>>(e, tw125, Ure, Uim, x, B, N, y);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  if (N <= kDirectMaxN && B <= 65535) {
    reverb_direct_kernel<<<dim3((N + 255) / 256, B), 256, (size_t)3 * N * sizeof(float), st>>>(x, Sim + d.L, N, d.L, y);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }

  if (d.N1 == 125) {
    col125_fwd_kernel<false><<<dim3(d.N2 / 32, pairs), kColThreads, kCol125Lds, st>>>(d, tw125, x, B, N, (long long)N, Ure, Uim);
  } else {
    const dim3 g1(d.N2 / 32, (d.M2 / 32 + 3) / 4, pairs);
    col_fwd_kernel<<<g1, 256, 0, st>>>(t + off_afwd(d), d, x, B, N, (long long)N, Ure, Uim);
  }
  NWS_CHECK_LAUNCH();
  launch_rows<false>(d, pairs, Ure, Uim, t, Sre, Sim, nullptr, nullptr, st);
  NWS_CHECK_LAUNCH();
  if (d.N1 == 125) {
    col125_inv_kernel<false><<<dim3(d.N2 / 32, pairs), kColThreads, kCol125Lds, st>>>(d, tw125, Ure, Uim, x, B, N, y);
  } else {
    const int rows_out = (N + d.N2 - 1) / d.N2;
    const int nt = (rows_out + 31) / 32;
    const dim3 g3(d.N2 / 32, (2 * nt + 3) / 4, pairs);
    col_inv_kernel<<<g3, 256, 0, st>>>(t + off_ainv(d), d, Ure, Uim, x, B, N, y);
  }
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

// One chunk of a LINEAR (streaming) reverb: the chunk's wet part is a circular convolution of length plan->L >= M + ir_len,
// which cannot wrap; its first M samples plus the carried tail are emitted, the rest is accumulated into the new tail.
int nws_reverb_linear_chunk(const NwsReverbPlan* plan, const void* tables, const void* spectrum, const float* x, int B, int M,
                            const float* tail_in, float* tail_out, int tail_len, float* y, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!plan_ok(plan) || !tables || !spectrum || !x || !y || !tail_in || !tail_out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || M <= 0 || tail_len <= 0 || M + tail_len > plan->L || plan->Lc != 0) return NWS_ERR_BAD_ARG;
  const PlanDev d = plan_dev(plan);
  const int pairs = (B + 1) / 2;
  const size_t need = (2 * (size_t)pairs * d.L + (size_t)B * d.L) * sizeof(float);
  if (workspace_bytes < need) return NWS_ERR_WORKSPACE;
  const float* t = static_cast<const float*>(tables);
  hipStream_t st = (hipStream_t)stream;
  float* Ure = static_cast<float*>(workspace);
  float* Uim = Ure + (size_t)pairs * d.L;
  float* wet = Uim + (size_t)pairs * d.L;  // (B, L)
  const float* Sre = static_cast<const float*>(spectrum);
  const float* Sim = Sre + d.L;
  const float2* tw125 = reinterpret_cast<const float2*>(t + off_tw125(d));
  if (d.N1 == 125) {
    col125_fwd_kernel<false><<<dim3(d.N2 / 32, pairs), kColThreads, kCol125Lds, st>>>(d, tw125, x, B, M, (long long)M, Ure, Uim);
  } else {
    const dim3 g1(d.N2 / 32, (d.M2 / 32 + 3) / 4, pairs);
    col_fwd_kernel<<<g1, 256, 0, st>>>(t + off_afwd(d), d, x, B, M, (long long)M, Ure, Uim);
  }
  NWS_CHECK_LAUNCH();
  launch_rows<false>(d, pairs, Ure, Uim, t, Sre, Sim, nullptr, nullptr, st);
  NWS_CHECK_LAUNCH();
  if (d.N1 == 125) {
    col125_inv_kernel<false><<<dim3(d.N2 / 32, pairs), kColThreads, kCol125Lds, st>>>(d, tw125, Ure, Uim, nullptr, B, d.L, wet);
  } else {
    const int nt = (d.N1 + 31) / 32;
    const dim3 g3(d.N2 / 32, (2 * nt + 3) / 4, pairs);
    col_inv_kernel<<<g3, 256, 0, st>>>(t + off_ainv(d), d, Ure, Uim, nullptr, B, d.L, wet);
  }
  NWS_CHECK_LAUNCH();
  reverb_tail_kernel<<<dim3(64, B), 256, 0, st>>>(x, wet, tail_in, tail_out, y, M, d.L, tail_len);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
