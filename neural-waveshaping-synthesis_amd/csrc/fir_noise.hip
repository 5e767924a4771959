// Time-varying FIR filtered noise (FIRNoiseSynth.forward, models/modules/generators.py:30-35).
//
// The reference multiplies a rectangular-window STFT (n_fft 256, hop 128, center/reflect) of one
// shared U[0,1) noise vector by the per-frame filter spectrum and inverts with istft(center=False):
// per frame that is a 256-point CIRCULAR convolution of the noise frame with the frame's FIR
// (SURVEY.md App. A.6), overlap-added and divided by the overlap count (1 for n < 128, else 2).
//
// Design: one wave produces one 128-sample output hop of one utterance.  Lanes
// 0-31 compute the first half of frame t's circular convolution (4 outputs each), lanes 32-63 the
// second half of frame t-1's; the two contributions meet with one half-swap.  Taps and noise are
// staged in LDS per workgroup (4 consecutive hops share 5 frames).  Each lane keeps a sliding
// 8-tap register window of the taps and of a copy delayed by one sample (so that tap PAIRS are
// even-aligned for both output parities): two ds_read_b128 of taps + one broadcast ds_read_b128 of
// noise feed 8 v_pk_fma_f32 = 16 MACs.  The NEWT branch is added here (cat + sum(1), models/neural_waveshaping.py:85-86).
#include "nws_common.h"

namespace {

constexpr int kL = NWS_FIR_LEN;  // 256
constexpr int kHop = NWS_HOP;    // 128
constexpr int kHopsPerBlock = 4;

struct NoiseLds {
  float taps[kHopsPerBlock + 1][kL];          // fir of frames t0-1 .. t0+3
  float taps1[kHopsPerBlock + 1][kL];         // the same taps delayed by one: taps1[k] = taps[(k-1) & 255]
  float sig[(kHopsPerBlock + 1) * kHop + kHop];  // padded noise [128(t0-1), 128(t0+3)+256)
};

// reflect-padded noise (torch.stft center=True, pad_mode="reflect", pad 128 each side).  One-shot forward: origin = 128,
// len = N-1.  Streaming windows pass the absolute noise stream with origin 0 and a len that only bites at the stream's end.
__device__ __forceinline__ float padded_noise(const float* __restrict__ noise, int len, int origin, int i) {
  int s = i - origin;
  if (s < 0) s = -s;
  if (s > len - 1) s = 2 * (len - 1) - s;
  s = s < 0 ? 0 : s;
  return noise[s];
}

__global__ __launch_bounds__(256) void fir_noise_kernel(const float* __restrict__ fir, const float* __restrict__ noise,
                                                        const float* __restrict__ add_in, int T, int len, int origin,
                                                        float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) NoiseLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, q = lane & 31;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * kHopsPerBlock;
  const int N = T * kHop;

  for (int e = tid; e < (kHopsPerBlock + 1) * kL; e += 256) {
    const int fr = e >> 8, k = e & 255;
    const int t = t0 - 1 + fr;
    const float v = (t >= 0 && t < T) ? fir[((size_t)b * T + t) * kL + k] : 0.0f;
    L.taps[fr][k] = v;
    L.taps1[fr][(k + 1) & 255] = v;
  }
  for (int e = tid; e < (kHopsPerBlock + 1) * kHop + kHop; e += 256) {
    const int i = (t0 - 1) * kHop + e;  // index into the padded noise, valid range [0, N+255)
    L.sig[e] = (i >= 0 && i < N + kL - 1) ? padded_noise(noise, len, origin, i) : 0.0f;
  }
  __syncthreads();

  const int t = t0 + wave;  // output hop
  if (t >= T) return;
  // half 0: frame t, outputs y_t[4q .. 4q+3];  half 1: frame t-1, outputs y_{t-1}[128+4q .. ]
  const int slot = wave + 1 - half;            // frame slot in LDS (frame t0-1+slot)
  const int nb = half * kHop + 4 * q;          // first output index inside the frame
  const float* f = &L.sig[slot * kHop];        // frame samples f[0..255]
  const float* h = L.taps[slot];

  // y[nb+i] = sum_m f[m] h[(nb+i-m) & 255], four outputs per lane, TWO taps per packed FMA:
  //   y_i += {f[m+1], f[m]} * {A[je], A[je+1]}   (m even; lanes: f[m+1] h[n_i-m-1]  and  f[m] h[n_i-m])
  // the pair (A[je], A[je+1]) must be even-aligned: for odd i it is (h[j-1], h[j]) of the natural array, for even i
  // (h[j-1], h[j]) = (h1[j], h1[j+1]) of the copy delayed by one sample.  With base = nb - m0 (multiple of 4) the pairs
  // needed per 4 taps are at base-2, base, base+2 of each array: a sliding window of two quads (hi = [base, base+4),
  // lo = [base-4, base)), one new ds_read_b128 per array per iteration.
  const float* h1 = L.taps1[slot];
  f32x2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f}, a2 = {0.0f, 0.0f}, a3 = {0.0f, 0.0f};
  float4 hi0 = *reinterpret_cast<const float4*>(&h[nb & 255]);
  float4 hi1 = *reinterpret_cast<const float4*>(&h1[nb & 255]);
#pragma unroll 4
  for (int m0 = 0; m0 < kL; m0 += 4) {
    const float4 lo0 = *reinterpret_cast<const float4*>(&h[(nb - m0 - 4) & 255]);
    const float4 lo1 = *reinterpret_cast<const float4*>(&h1[(nb - m0 - 4) & 255]);
    const float4 fv = *reinterpret_cast<const float4*>(&f[m0]);
    const f32x2 fs0 = {fv.y, fv.x}, fs1 = {fv.w, fv.z};
    // odd outputs (natural taps)
    a1 = fma2(fs0, f32x2{hi0.x, hi0.y}, a1);
    a3 = fma2(fs0, f32x2{hi0.z, hi0.w}, a3);
    a1 = fma2(fs1, f32x2{lo0.z, lo0.w}, a1);
    a3 = fma2(fs1, f32x2{hi0.x, hi0.y}, a3);
    // even outputs (taps delayed by one)
    a0 = fma2(fs0, f32x2{hi1.x, hi1.y}, a0);
    a2 = fma2(fs0, f32x2{hi1.z, hi1.w}, a2);
    a0 = fma2(fs1, f32x2{lo1.z, lo1.w}, a0);
    a2 = fma2(fs1, f32x2{hi1.x, hi1.y}, a2);
    hi0 = lo0;
    hi1 = lo1;
  }
  const float y0 = a0.x + a0.y, y1 = a1.x + a1.y, y2 = a2.x + a2.y, y3 = a3.x + a3.y;
  // overlap-add of the two frames covering this hop, divided by the overlap count
  const float o0 = y0 + nws_swap_halves(y0);
  const float o1 = y1 + nws_swap_halves(y1);
  const float o2 = y2 + nws_swap_halves(y2);
  const float o3 = y3 + nws_swap_halves(y3);
  if (half == 0) {
    const float inv = t == 0 ? 1.0f : 0.5f;
    const size_t o = (size_t)b * N + (size_t)t * kHop + 4 * q;
    float4 r = make_float4(o0 * inv, o1 * inv, o2 * inv, o3 * inv);
    if (add_in != nullptr) {
      const float4 a = *reinterpret_cast<const float4*>(&add_in[o]);
      r.x = a.x + r.x;
      r.y = a.y + r.y;
      r.z = a.z + r.z;
      r.w = a.w + r.w;
    }
    *reinterpret_cast<float4*>(&out[o]) = r;
  }
}

}  // namespace

extern "C" int nws_fir_noise_window(const float* fir, const float* noise, int noise_len, int origin, const float* add_in,
                                    int B, int T, float* out, void* stream) {
  if (!fir || !noise || !out || B <= 0 || T <= 0 || noise_len < 2 || origin < 0) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  const dim3 grid((T + kHopsPerBlock - 1) / kHopsPerBlock, B);
  fir_noise_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(fir, noise, add_in, T, noise_len, origin, out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_fir_noise(const float* fir, const float* noise, const float* add_in, int B, int T, float* out,
                             void* stream) {
  return nws_fir_noise_window(fir, noise, T * kHop - 1, kL / 2, add_in, B, T, out, stream);
}
