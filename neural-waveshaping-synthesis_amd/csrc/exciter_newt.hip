// Fused harmonic exciter + waveshaper bank for gfx950 (wave64, fp32 MFMA).
//
// Replaces, per utterance b and 128-sample hop j (one workgroup = 4 waves x 32 samples):
//   F.upsample(f0)                      models/neural_waveshaping.py:75
//   HarmonicOscillator.forward          models/modules/generators.py:58-66
//   harmonic_mixer  Conv1d(101->64,1)   models/neural_waveshaping.py:66
//   NEWT.forward / FastNEWT.shaping_fn  models/modules/shaping.py:67-79, :136-151
//
// Design: the (B,101,N) oscillator bank and the (B,64,N) exciter of the
// reference never exist in HBM.  Each lane evaluates one sin() per MFMA step directly in the
// B-operand layout of v_mfma_f32_32x32x2_f32 (lane l -> harmonic 2s+(l>>5), sample l&31); the
// 101->64 mixer weights are the A operand, staged once per workgroup in LDS (transposed, row
// stride 65 -> conflict-free).  The 64x32 accumulator tile then goes straight through
// FiLM -> LUT (or sin-MLP) -> FiLM -> 64->1 mix in registers; one coalesced 128 B store per wave.
#include "nws_common.h"

namespace {

constexpr int kK = NWS_N_HARMONICS;         // 101
constexpr int kKPad = 102;                  // 51 MFMA steps of 2 harmonics
constexpr int kS = NWS_N_SHAPERS;           // 64
constexpr int kWtStride = 65;               // LDS row stride of the transposed mixer
constexpr int kTile = 128;                  // samples per workgroup (= control hop)
constexpr float kTau = 6.283185307179586f;  // fl32(math.tau)
constexpr float kPi = 3.141592653589793f;   // fl32(math.pi)

enum Mode { kModeLut = 0, kModeExact = 1, kModeExciterOnly = 2 };

// ---------------------------------------------------------------------------------------------
// carry[b][c] = sum_{n<32c} f0_up[b][n] in float64 (torch's CPU cumsum accumulates in double and
// rounds each element to fp32).  One workgroup per utterance.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void phase_carry_kernel(const float* __restrict__ f0,
                                                           const float* __restrict__ f0_up, int T,
                                                           double* __restrict__ carry) {
  const int b = blockIdx.x;
  const int N = T * NWS_HOP;
  const int nchunks = N / 32;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ double wave_tot[16];
  double running = 0.0;
  for (int base = 0; base < nchunks; base += 1024) {
    const int c = base + threadIdx.x;
    double s = 0.0;
    if (c < nchunks) {
      const int n0 = c * 32;
      if (f0_up != nullptr) {
        const float4* p = reinterpret_cast<const float4*>(f0_up + (size_t)b * N + n0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 v = p[i];
          s += (double)v.x;
          s += (double)v.y;
          s += (double)v.z;
          s += (double)v.w;
        }
      } else {
        const float* x = f0 + (size_t)b * T;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
          const NwsLerp L = nws_lerp_coeff(n0 + i, T);
          s += (double)nws_lerp(x[L.i0], x[L.i1], L.w0, L.w1);
        }
      }
    }
    double v = s;  // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double t = __shfl_up(v, off, 64);
      if (lane >= off) v += t;
    }
    if (lane == 63) wave_tot[wave] = v;
    double excl = __shfl_up(v, 1, 64);
    if (lane == 0) excl = 0.0;
    __syncthreads();
    double wp = 0.0, total = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const double t = wave_tot[i];
      if (i < wave) wp += t;
      total += t;
    }
    if (c < nchunks) carry[(size_t)b * nchunks + c] = running + wp + excl;
    running += total;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// exact shaper s at scalar argument x (TrainableNonlinearity, models/modules/shaping.py:36-37;
// grouped 1x1 convs == 64 independent 1->8->8->8->1 sin-MLPs).  Weights in LDS.
// ---------------------------------------------------------------------------------------------
struct ShaperLds {
  float in_scale[64];
  float w0[512], b0[512];
  float w2[4096], b2[512];
  float w4[4096], b4[512];
  float w6[512], b6[64];
};

__device__ __forceinline__ float exact_shaper(const ShaperLds& W, int s, float x) {
  const float a = W.in_scale[s] * x;
  float h1[8], h2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h1[i] = nws_sinf(fmaf(W.w0[s * 8 + i], a, W.b0[s * 8 + i]));
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float acc = W.b2[s * 8 + o];
    const float* wr = &W.w2[(s * 8 + o) * 8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(wr[i], h1[i], acc);
    h2[o] = nws_sinf(acc);
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    float acc = W.b4[s * 8 + o];
    const float* wr = &W.w4[(s * 8 + o) * 8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(wr[i], h2[i], acc);
    h1[o] = nws_sinf(acc);
  }
  float acc = W.b6[s];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = fmaf(W.w6[s * 8 + i], h1[i], acc);
  return nws_sinf(acc);
}

__device__ __forceinline__ void load_shaper_lds(ShaperLds& L, const NwsWeights& w, int tid, int nthreads) {
  for (int i = tid; i < 64; i += nthreads) {
    L.in_scale[i] = w.shaper_in_scale[i];
    L.b6[i] = w.shaper_b6[i];
  }
  for (int i = tid; i < 512; i += nthreads) {
    L.w0[i] = w.shaper_w0[i];
    L.b0[i] = w.shaper_b0[i];
    L.b2[i] = w.shaper_b2[i];
    L.b4[i] = w.shaper_b4[i];
    L.w6[i] = w.shaper_w6[i];
  }
  for (int i = tid; i < 4096; i += nthreads) {
    L.w2[i] = w.shaper_w2[i];
    L.w4[i] = w.shaper_w4[i];
  }
}

// FastNEWT.shaping_fn (models/modules/shaping.py:136-151), quirks kept: index scale size/(max-min)
// against a linspace grid of step (max-min)/(size-1); fract taken against the CLAMPED lower index.
__device__ __forceinline__ float lut_shaper(const float* __restrict__ row, int size, float tmin, float trange, float x) {
  const float idx = __fdiv_rn((float)size * (x - tmin), trange);
  float fl = floorf(idx);
  fl = fmaxf(fl, 0.0f);
  fl = fminf(fl, (float)(size - 1));
  const int lo = (int)fl;
  const int up = lo + 1 < size ? lo + 1 : size - 1;
  const float fract = idx - fl;
  const float lv = row[lo];
  const float uv = row[up];
  return (uv - lv) * fract + lv;
}

struct ExcLds {
  float wt[kKPad * kWtStride];  // wt[kk][s] = mixer_w[s][kk]
  float mix_b[kS];
  float out_w[kS];
  float shift[kKPad];
  float film[3][NWS_FILM_CH];   // frames j-1, j, j+1 (clamped)
};

template <int MODE>
__global__ __launch_bounds__(256) void exciter_newt_kernel(NwsWeights w, const float* __restrict__ f0,
                                                           const float* __restrict__ f0_up,
                                                           const double* __restrict__ carry,
                                                           const float* __restrict__ phase_u,
                                                           const float* __restrict__ rand_phase,
                                                           const float* __restrict__ film, int T, float sample_rate,
                                                           float* __restrict__ exciter_out,
                                                           float* __restrict__ newt_out) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ExcLds& L = *reinterpret_cast<ExcLds*>(smem_raw);
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw + ((sizeof(ExcLds) + 15) & ~size_t(15)));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int half = lane >> 5;
  const int col = lane & 31;
  const int j = blockIdx.x;  // hop
  const int b = blockIdx.y;
  const int N = T * NWS_HOP;

  // ---- stage the workgroup constants in LDS ----
  for (int e = tid; e < kS * kK; e += 256) {
    const int s = e / kK, kk = e - s * kK;
    L.wt[kk * kWtStride + s] = w.mixer_w[e];
  }
  if (tid < kS) {
    L.wt[kK * kWtStride + tid] = 0.0f;  // padded harmonic 102
    L.mix_b[tid] = w.mixer_b[tid];
    if (MODE != kModeExciterOnly) L.out_w[tid] = w.newt_out_w[tid];
  }
  // _create_phase_shift (generators.py:54-56): fl(fl(u * rand_phase) - fl32(pi))
  if (tid < kKPad) L.shift[tid] = tid < kK ? phase_u[tid] * rand_phase[tid] - kPi : 0.0f;
  if (MODE != kModeExciterOnly) {
    for (int e = tid; e < 3 * NWS_FILM_CH; e += 256) {
      const int q = e >> 8, c = e & 255;
      int f = j - 1 + q;
      f = f < 0 ? 0 : (f > T - 1 ? T - 1 : f);
      L.film[q][c] = film[((size_t)b * T + f) * NWS_FILM_CH + c];
    }
  }
  if (MODE == kModeExact) load_shaper_lds(SH, w, tid, 256);

  // ---- per-sample phase: fp64 prefix sum -> fp32 rounding chain of the reference ----
  const int n = j * kTile + wave * 32 + col;
  const NwsLerp lc = nws_lerp_coeff(n, T);
  float f0n;
  if (f0_up != nullptr) {
    f0n = f0_up[(size_t)b * N + n];
  } else {
    const float* x = f0 + (size_t)b * T;
    f0n = nws_lerp(x[lc.i0], x[lc.i1], lc.w0, lc.w1);
  }
  double cs = (double)f0n;
#pragma unroll
  for (int off = 1; off < 32; off <<= 1) {
    const double t = __shfl_up(cs, off, 32);
    if (col >= off) cs += t;
  }
  cs += carry[(size_t)b * (N / 32) + (n >> 5)];
  const float csum = (float)cs;                                 // fl32 of the double prefix sum
  const float phase = __fdiv_rn(kTau * csum, sample_rate);      // math.tau * cumsum / sample_rate
  const float nyquist = sample_rate * 0.5f;

  __syncthreads();

  // ---- 101 harmonics -> 64 shapers on the matrix cores ----
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    acc0[r] = 0.0f;
    acc1[r] = 0.0f;
  }
  const float* wt_lane = &L.wt[half * kWtStride + col];
  for (int s = 0; s < kKPad / 2; ++s) {
    const int kk = 2 * s + half;
    const float kf = (float)(kk + 1);
    const bool live = (f0n * kf) < nyquist;  // anti-alias mask on the upsampled F0 (generators.py:50-52)
    if (__all(!live)) break;                 // k*f0 only grows with k: everything above is masked too
    const float arg = kf * phase + L.shift[kk];
    const float v = live ? nws_sinf(arg) : 0.0f;
    const float a0 = wt_lane[s * 2 * kWtStride];
    const float a1 = wt_lane[s * 2 * kWtStride + 32];
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, v, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, v, acc1, 0, 0, 0);
  }

  // accumulator element r of M-tile m: shaper 32m + (r&3) + 8(r>>2) + 4*half, sample `col`
  if (exciter_out != nullptr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s0 = (r & 3) + 8 * (r >> 2) + 4 * half;
      exciter_out[((size_t)b * kS + s0) * N + n] = acc0[r] + L.mix_b[s0];
      exciter_out[((size_t)b * kS + s0 + 32) * N + n] = acc1[r] + L.mix_b[s0 + 32];
    }
  }
  if (MODE == kModeExciterOnly) return;

  // ---- FiLM -> shaper -> FiLM -> 64->1 mix, all in registers ----
  const float* p0 = L.film[lc.i0 - (j - 1)];
  const float* p1 = L.film[lc.i1 - (j - 1)];
  const float trange = w.lut_max - w.lut_min;
  float partial = 0.0f;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s0 = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float x = (m == 0 ? acc0[r] : acc1[r]) + L.mix_b[s0];
      const float g_i = nws_lerp(p0[s0], p1[s0], lc.w0, lc.w1);
      const float b_i = nws_lerp(p0[64 + s0], p1[64 + s0], lc.w0, lc.w1);
      const float g_n = nws_lerp(p0[128 + s0], p1[128 + s0], lc.w0, lc.w1);
      const float b_n = nws_lerp(p0[192 + s0], p1[192 + s0], lc.w0, lc.w1);
      const float xi = g_i * x + b_i;  // FiLM (models/modules/dynamic.py:8)
      float sh;
      if (MODE == kModeLut) {
        sh = lut_shaper(w.lut + (size_t)s0 * w.lut_size, w.lut_size, w.lut_min, trange, xi);
      } else {
        sh = exact_shaper(SH, s0, xi);
      }
      const float y = g_n * sh + b_n;
      partial = fmaf(L.out_w[s0], y, partial);
    }
  }
  const float total = partial + nws_swap_halves(partial) + w.newt_out_b[0];
  if (half == 0) newt_out[(size_t)b * N + n] = total;
}

// element-wise shaper application on (B,64,N) (stage tests / FastNEWT table construction)
template <int MODE>
__global__ __launch_bounds__(256) void shaper_apply_kernel(NwsWeights w, const float* __restrict__ x, int64_t N,
                                                           float* __restrict__ y) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw);
  if (MODE == kModeExact) load_shaper_lds(SH, w, threadIdx.x, 256);
  __syncthreads();
  const int64_t row = blockIdx.y;  // b*64 + s
  const int s = (int)(row & 63);
  const float trange = w.lut_max - w.lut_min;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
    const float v = x[row * N + i];
    y[row * N + i] = MODE == kModeLut ? lut_shaper(w.lut + (size_t)s * w.lut_size, w.lut_size, w.lut_min, trange, v)
                                      : exact_shaper(SH, s, v);
  }
}

__global__ __launch_bounds__(256) void shaper_table_kernel(NwsWeights w, int size, float tmin, float tmax,
                                                           float* __restrict__ table) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  ShaperLds& SH = *reinterpret_cast<ShaperLds*>(smem_raw);
  load_shaper_lds(SH, w, threadIdx.x, 256);
  __syncthreads();
  const int s = blockIdx.y;
  // torch.linspace(min, max, size) in fp32 (ATen RangeFactories, symmetric form; bit-exact with the
  // CPU kernel as probed in the build container): step = (max-min)/(size-1); first half
  // fma(step, i, min), second half fma(-step, size-1-i, max).
  const float step = __fdiv_rn(tmax - tmin, (float)(size - 1));
  const int halfway = size / 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < size; i += gridDim.x * 256) {
    const float xv = i < halfway ? fmaf(step, (float)i, tmin) : fmaf(-step, (float)(size - 1 - i), tmax);
    table[(size_t)s * size + i] = exact_shaper(SH, s, xv);
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

int nws_phase_carry(const float* f0, const float* f0_up, int B, int T, double* carry, void* stream) {
  if ((!f0 && !f0_up) || !carry || B <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  phase_carry_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(f0, f0_up, T, carry);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_exciter_newt(const NwsWeights* w, const float* f0, const float* f0_up, const double* carry,
                     const float* phase_u, const float* rand_phase, const float* film, int B, int T,
                     float sample_rate, float* exciter_out, float* newt_out, void* stream) {
  if (!weights_ok(w) || (!f0 && !f0_up) || !carry || !phase_u || !rand_phase || B <= 0 || T <= 0) return NWS_ERR_BAD_ARG;
  if (!exciter_out && !newt_out) return NWS_ERR_BAD_ARG;
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
      exciter_newt_kernel<kModeLut><<<grid, 256, base, st>>>(*w, f0, f0_up, carry, phase_u, rand_phase, film, T,
                                                             sample_rate, exciter_out, newt_out);
    } else {
      if (!w->shaper_w0 || !w->shaper_w2 || !w->shaper_w4 || !w->shaper_w6) return NWS_ERR_BAD_ARG;
      exciter_newt_kernel<kModeExact><<<grid, 256, base + sizeof(ShaperLds), st>>>(
          *w, f0, f0_up, carry, phase_u, rand_phase, film, T, sample_rate, exciter_out, newt_out);
    }
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
