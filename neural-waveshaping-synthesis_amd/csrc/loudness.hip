// Perceptual-loudness feature (the step BEFORE the synthesis path):
//   neural_waveshaping_synthesis/data/utils/loudness_extraction.py:10-67 with gin/data/urmp_4second_crepe.gin:11-14:
//   |librosa.stft(audio, n_fft 1024, hop 128, hann, center/reflect)| -> amplitude_to_db(ref=np.max, amin=epsilon, top_db 80)
//   -> mean over bins -> (L + 80) / 80.   (The A-weighting of :25-38 is computed and then NOT applied by the reference.)
//
// Design: the windowed real DFT of every frame is ONE GEMM against a constant matrix with the hann window folded in,
//   S[row][t] = sum_n D[row][n] * x_pad[hop t + n],   rows interleaved (2k = Re, 2k+1 = Im of bin k),
// on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate: the reference's own transform is a float32 FFT).
// A workgroup = 32 frames x 4 M-tiles; the 31 hop + n_fft samples the 32 overlapping frames are cut from are staged in LDS
// once (reflect padding resolved there), skewed by one word per 128 so that the 32 lanes of a B-operand read (stride hop)
// fall into different banks.  Re/Im of a bin are adjacent accumulator registers of one lane -> power in-lane; the
// spectrogram is kept as (B, bins, frames) so that both its write and the second pass are coalesced along frames.
// The dB reference is the maximum over the WHOLE spectrogram of an utterance: pass 1 leaves it in one word per utterance
// (atomicMax on the float bits, valid for non-negative values), pass 2 clips, averages and normalises.
#include "nws_common.h"

namespace {

constexpr int kFrames = 32;  // frames per workgroup (MFMA N)

__device__ __forceinline__ int reflect_index(long long i, int N) {  // numpy "reflect" (no edge repeat), one fold each side
  if (i < 0) i = -i;
  if (i >= N) i = 2LL * (N - 1) - i;
  return (int)(i < 0 ? 0 : i);
}

__device__ __forceinline__ int skew(int j) { return j + (j >> 7); }

// rows 2k / 2k+1 = hann[n] cos(2 pi k n / n_fft) / -hann[n] sin(...), K permuted as k(s, h) = (n_fft/2) h + s is NOT applied
// here (plain row-major); rows beyond 2 (n_fft/2 + 1) are zero.  Evaluated in double.
__global__ void dft_matrix_kernel(int n_fft, int rows_pad, float* __restrict__ out) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long long)rows_pad * n_fft) return;
  const int row = (int)(e / n_fft), n = (int)(e - (long long)row * n_fft);
  const int k = row >> 1;
  float v = 0.0f;
  if (k <= n_fft / 2) {
    const double w = 0.5 - 0.5 * cospi(2.0 * (double)n / (double)n_fft);
    const long long kn = ((long long)k * n) % n_fft;  // exact phase reduction
    double s, c;
    sincospi(2.0 * (double)kn / (double)n_fft, &s, &c);
    v = (float)((row & 1) ? -w * s : w * c);
  }
  out[e] = v;
}

__global__ __launch_bounds__(256) void loudness_power_kernel(const float* __restrict__ audio, int N, int n_fft, int hop,
                                                             int frames, int frames_pad, const float* __restrict__ dft,
                                                             int m_tiles, float* __restrict__ power,
                                                             unsigned* __restrict__ max_bits) {
  extern __shared__ float xs[];  // skewed window of the frame tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh = lane >> 5, col = lane & 31;
  const int t0 = blockIdx.x * kFrames;
  const int mt = blockIdx.y * 4 + wave;
  const int b = blockIdx.z;
  const float* x = audio + (size_t)b * N;
  const int span = (kFrames - 1) * hop + n_fft;
  const long long first = (long long)hop * t0 - n_fft / 2;  // center=True: frame t covers [hop t - n_fft/2, hop t + n_fft/2)
  for (int j = tid; j < span; j += 256) xs[skew(j)] = x[reflect_index(first + j, N)];
  __syncthreads();
  if (mt >= m_tiles) return;

  const int half_k = n_fft / 2;
  const float* arow = dft + (size_t)(32 * mt + col) * n_fft + (size_t)half_k * kh;
  const int boff = hop * col + half_k * kh;
  f32x16 acc = {};
  for (int s0 = 0; s0 < half_k; s0 += 8) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow + s0), a1 = *reinterpret_cast<const float4*>(arow + s0 + 4);
    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], xs[skew(boff + s0 + i)], acc, 0, 0, 0);
  }
  // rows (r, r+1), r even = (Re, Im) of bin 16 mt + (r&3)/2 + 4 (r>>2) + 2 kh; column = frame t0 + col
  const int t = t0 + col;
  const int bins = n_fft / 2 + 1;
  float mx = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const int bin = 16 * mt + ((r & 3) >> 1) + 4 * (r >> 2) + 2 * kh;
    const float p = fmaf(acc[r], acc[r], acc[r + 1] * acc[r + 1]);
    if (bin < bins && t < frames) {
      power[((size_t)b * bins + bin) * frames_pad + t] = p;
      mx = fmaxf(mx, p);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) atomicMax(&max_bits[b], __float_as_uint(mx));
}

__global__ void loudness_db_kernel(const float* __restrict__ power, const unsigned* __restrict__ max_bits, int bins,
                                   int frames, int frames_pad, float amin, float top_db, int normalise,
                                   float* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= frames) return;
  const float k10 = 3.0102999566398120f;  // 10 / log2(10)
  const float floor_p = amin * amin;
  const float ref_db = k10 * __log2f(fmaxf(floor_p, __uint_as_float(max_bits[b])));
  // log_spec.max() is max(floor, max power) - ref = 0 by construction, so the top_db clip sits at -top_db
  const float* p = power + (size_t)b * bins * frames_pad + t;
  float s = 0.0f;
  for (int k = 0; k < bins; ++k) {
    const float db = k10 * __log2f(fmaxf(floor_p, p[(size_t)k * frames_pad])) - ref_db;
    s += fmaxf(db, -top_db);
  }
  float l = s / (float)bins;
  if (normalise) l = (l + 80.0f) / 80.0f;
  out[(size_t)b * frames + t] = l;
}

__host__ int rows_padded(int n_fft) { return ((2 * (n_fft / 2 + 1)) + 31) / 32 * 32; }
// LDS tile of one workgroup: the (kFrames - 1) hop + n_fft samples its 32 overlapping frames are cut from (+ skew words)
constexpr size_t kLoudnessLdsCap = 160 * 1024;
__host__ size_t tile_lds_bytes(int n_fft, int hop) {
  const size_t span = (size_t)(32 - 1) * hop + n_fft;
  return (span + (span >> 7) + 1) * sizeof(float);
}
__host__ bool fft_ok(int n_fft, int hop) {
  // the tile must fit the CU's 160 KB of LDS: e.g. n_fft 2048 allows hop <= 1250, n_fft 1024 hop <= 1024
  return n_fft >= 64 && n_fft <= 2048 && (n_fft & (n_fft - 1)) == 0 && hop >= 1 && hop <= n_fft &&
         tile_lds_bytes(n_fft, hop) <= kLoudnessLdsCap;
}

}  // namespace

extern "C" {

size_t nws_loudness_dft_bytes(int n_fft) {
  if (!fft_ok(n_fft, 1)) return 0;
  return (size_t)rows_padded(n_fft) * n_fft * sizeof(float);
}

int nws_loudness_dft_matrix(int n_fft, float* dft_out, void* stream) {
  if (!fft_ok(n_fft, 1) || !dft_out) return NWS_ERR_BAD_ARG;
  const long long n = (long long)rows_padded(n_fft) * n_fft;
  dft_matrix_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(n_fft, rows_padded(n_fft), dft_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_loudness_frames(int N, int hop) { return (N <= 0 || hop <= 0) ? 0 : 1 + N / hop; }

size_t nws_loudness_workspace_bytes(int B, int N, int n_fft, int hop) {
  if (B <= 0 || N <= 0 || !fft_ok(n_fft, hop)) return 0;
  const size_t frames_pad = ((size_t)nws_loudness_frames(N, hop) + 31) / 32 * 32;
  return ((size_t)B * (n_fft / 2 + 1) * frames_pad) * sizeof(float) + (((size_t)B * sizeof(unsigned) + 255) & ~size_t(255));
}

int nws_loudness(const float* audio, int B, int N, int n_fft, int hop, const float* dft, float amin, float top_db,
                 int normalise, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!audio || !dft || !out || !workspace || B <= 0 || !fft_ok(n_fft, hop)) return NWS_ERR_BAD_ARG;
  if (N <= n_fft / 2) return NWS_ERR_BAD_ARG;  // reflect padding needs more than n_fft/2 samples (as in the reference)
  if (!(amin > 0.0f) || !(top_db >= 0.0f)) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  if (workspace_bytes < nws_loudness_workspace_bytes(B, N, n_fft, hop)) return NWS_ERR_WORKSPACE;
  const int frames = nws_loudness_frames(N, hop);
  const int frames_pad = (frames + 31) / 32 * 32;
  const int bins = n_fft / 2 + 1;
  const size_t max_bytes = ((size_t)B * sizeof(unsigned) + 255) & ~size_t(255);
  unsigned* max_bits = static_cast<unsigned*>(workspace);
  float* power = reinterpret_cast<float*>(static_cast<char*>(workspace) + max_bytes);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(max_bits, 0, (size_t)B * sizeof(unsigned), st);
  if (e != hipSuccess) return (int)e;
  const int m_tiles = rows_padded(n_fft) / 32;
  const size_t lds = tile_lds_bytes(n_fft, hop);
  static unsigned long long attr_devices = 0;
  if (nws_first_use_on_device(attr_devices)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(loudness_power_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kLoudnessLdsCap);
    if (e != hipSuccess) return (int)e;
  }
  const dim3 grid(frames_pad / kFrames, (m_tiles + 3) / 4, B);
  loudness_power_kernel<<<grid, 256, lds, st>>>(audio, N, n_fft, hop, frames, frames_pad, dft, m_tiles, power, max_bits);
  NWS_CHECK_LAUNCH();
  const dim3 g2((frames + 255) / 256, B);
  loudness_db_kernel<<<g2, 256, 0, st>>>(power, max_bits, bins, frames, frames_pad, amin, top_db, normalise, out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
