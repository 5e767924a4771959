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
#include <type_traits>

#include "nws_common.h"

namespace {

constexpr int kL = NWS_FIR_LEN;  // 256
constexpr int kHop = NWS_HOP;    // 128
constexpr int kHalf = NWS_FIR_HALF;  // taps per stored row: h[128 .. 255]
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

  // fir holds the upper half-taps u[d] = h[128 + d]; the row is symmetric about tap 128 and h[0] = 0 (include/nws_hip.h)
  for (int e = tid; e < (kHopsPerBlock + 1) * kL; e += 256) {
    const int fr = e >> 8, k = e & 255;
    const int t = t0 - 1 + fr;
    const int d = k >= kHalf ? k - kHalf : kHalf - k;      // k = 0 -> d = 128: the zero tap
    const float v = (t >= 0 && t < T && d < kHalf) ? fir[((size_t)b * T + t) * kHalf + d] : 0.0f;
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

// ---------------------------------------------------------------------------------------------------------------------
// Batched form on the matrix cores (B >= 16).  All utterances share ONE noise vector (generators.py:30), so for frame t
// the 256x256 circulant C_t[n][k] = f_t[(n-k) & 255] of the noise frame is the same GEMM operand for every utterance:
//     Y_t (B x 256) = H_t (B x 256 taps) * C_t^T .
// One workgroup = one output hop x 32 utterances; the hop needs rows 0..127 of frame t's circulant and rows 128..255 of
// frame t-1's, i.e. ONE accumulation over K = 512: D[b][j] = sum_k h_t[b][k] f_t[(j-k)&255] + sum_k h_{t-1}[b][k] f_{t-1}[(128+j-k)&255].
// fp32 accuracy from fp16 MFMAs by the two-term split (x = hi + lo, three products); taps are pre-scaled by a per-utterance
// power of two and the noise by 2^10 so that the lo parts stay out of the fp16 subnormal range.
//  * A operand (utterance rows): taps as fp16 hi/lo half rows in LDS, row stride 272 B (bank-conflict-free b128 reads);
//    HBM holds the upper 128 taps of every frame only (mirror-symmetric rows, include/nws_hip.h).
//  * B operand (sample columns): lane (j, khalf) needs 8 CONSECUTIVE entries of the reversed noise frame starting at
//    (k0 + 8 khalf - j) & 255 -- an arbitrary offset, but its low three bits are (-j) & 7, fixed per lane: eight copies of
//    the reversed frame, copy c shifted by c, make every read an aligned ds_read_b128 of one 16-byte block of its copy.
//    Round 5: a ds_read_b128 is served in four NON-contiguous 16-lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...;
//    MI355X_MICROARCH, LDS), not in runs of 16 lanes: the padded copy stride of rounds 3-4 (544 B, "conflict-free") put two
//    lanes of every group on one bank (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.283, the only hot kernel above 0.06).  Every
//    plain stride leaves a 2-way conflict (exhaustive check, tools/lds_groups_fir.py); rotating the 32 blocks of copy c by
//    kRot[c] = {0, 1, 5, 9, 13, 5, 9, 13} blocks makes all four groups hit 16 distinct 16-byte slots for every k0 and wave.
//    Copies are exactly 512 B now (a read never straddles a block, so the 32 wrap-around bytes per copy are gone).
//  * D rows are utterances, columns samples: each accumulator register stores 2 x 128 B contiguous segments.
constexpr int kUtt = 32;
constexpr int kCopyHalfs = 256;      // one copy = the 256 entries of the reversed frame = 32 blocks of 16 B
// block rotation of copy c (in blocks of 8 halfs), packed one nibble per copy: {0, 1, 5, 9, 13, 5, 9, 13}
constexpr unsigned kCopyRot = 0xD95D9510u;
__device__ __forceinline__ int copy_rot_halfs(int c) { return (int)((kCopyRot >> (4 * c)) & 15u) * 8; }
constexpr int kRowHalfs = 136;       // half a tap row (128 taps) + pad: 272 B stride, conflict-free ds_read_b128
constexpr float kNoiseScale = 1024.0f;  // noise (U[0,1) in the reference; anything within +-32 is fine) times 2^10
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

struct NoiseMfmaLds {
  _Float16 rhi[2][8][kCopyHalfs];  // [frame t | frame t-1 advanced by 128][shift c][v] = R[(v + c) & 255]
  _Float16 rlo[2][8][kCopyHalfs];
  _Float16 hhi[kUtt][kRowHalfs];   // HALF the taps (128) of the frame being accumulated, times the utterance's scale:
  _Float16 hlo[kUtt][kRowHalfs];   // staging half rows keeps the block at 36 KB of LDS = 4 workgroups per CU
  __attribute__((aligned(16))) float unscale[kUtt];   // 1 / (tap scale * noise scale) per utterance
  float win[4 * kHop];             // padded noise [128 (t-1), 128 (t-1) + 384), times the noise scale: frames t-1 and t (+ 128 of
                                   // padding: every thread stores two entries, so both loads are requested up front)
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

// (hi, lo) fp16 split of two values: v_cvt_pk_f16_f32 for hi, one v_fma_mix{lo,hi}_f16 per lo (exact residual rounded once)
__device__ __forceinline__ void split16x2(float a, float b, f16x2& hi, f16x2& lo) {
  hi = __builtin_convertvector(f32x2{a, b}, f16x2);
  const unsigned hp = __builtin_bit_cast(unsigned, hi);
  unsigned lp;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lp) : "v"(hp), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lp) : "v"(hp), "v"(b));
  lo = __builtin_bit_cast(f16x2, lp);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// maximum of non-negative values over each 32-lane half of the wave, in every lane of the half
__device__ __forceinline__ float half_max(float v) {
  v = fmaxf(v, dpp_f32<0xb1, 0xf>(v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f32<0x4e, 0xf>(v));   // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f32<0x141, 0xf>(v));  // row_half_mirror
  v = fmaxf(v, dpp_f32<0x140, 0xf>(v));  // row_mirror: every lane holds its 16-lane row's maximum
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));   // rows 0|1 and 2|3 exchanged: the half's maximum
}
__global__ __launch_bounds__(256, 4) void fir_noise_mfma_kernel(const float* __restrict__ fir, const float* __restrict__ noise,
                                                                const float* __restrict__ add_in, int B, int T, int len,
                                                                int origin, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) NoiseMfmaLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kh = lane >> 5, col = lane & 31;
  // Hop t stages the tap rows of frames t and t-1, so every row is wanted by two workgroups.  Workgroups are dealt to the eight
  // XCDs round-robin (block b -> XCD b % 8, MI355X_MICROARCH; placement only matters for speed): with t = blockIdx.x the two
  // always sat on different XCDs and both fetched the row from HBM (32.8 MB per launch instead of 16.4, L2 hit 0.24).  The hop
  // range is cut into eight contiguous chunks, one per XCD; neighbours in t are then neighbours in launch order on ONE L2.
  const int per_xcd = gridDim.x >> 3;                         // gridDim.x = 8 ceil(T / 8)
  const int t = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (t >= T) return;
  const int b0 = blockIdx.y * kUtt;
  const int N = T * kHop;

  // taps: wave w covers rows r = w + 4 q (q = 0..7).  One load instruction fetches HALF rows of TWO rows: lanes 0..31 the
  // taps [128 h + 4 p, +4) of row w + 4 (2 it), lanes 32..63 the same taps of row w + 4 (2 it + 1); v[it] holds the first
  // halves (h = 0), v[it + 4] the second.  Staging a half frame then has all 64 lanes busy (a row per instruction left 32
  // idle in every one of the 416 staging instructions), and the per-row scale search runs on two rows at once.
  const int p32 = lane & 31;
  const int my_row = wave + 4 * kh;   // this lane's row of it = 0; row(it) = wave + 4 (2 it + kh) = my_row + 8 it
  // The stored row is the upper half u[d] = h[128 + d] (include/nws_hip.h); the lower half is its mirror image:
  // h[4p + i] = u[128 - 4p - i] with u[128] := 0 (h[0] = 0).  Lane p fetches the aligned quad A = u[124 - 4p .. 127 - 4p] and takes
  // u[128 - 4p] = the first element of lane p-1's quad (one DPP lane shift): both halves
  // of a row come out of the SAME 512 B of HBM.
  auto load_rows = [&](int frame, float4 (&v)[8]) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int b = b0 + my_row + 8 * it;
      const bool ok = frame >= 0 && frame < T && b < B;
      // always-in-bounds addresses and a select afterwards: as `ok ? *ptr : zero` hipcc predicated every COMPONENT on its own
      // (64 global_load_dword per lane instead of 16 global_load_dwordx4: four instructions over the same cache lines)
      const int bc = b < B ? b : B - 1, fc = frame < 0 ? 0 : (frame < T ? frame : T - 1);
      // (two index expressions off the same row start: written as `src + 124 - 4 p` hipcc put a temporary on the stack)
      const float* src = &fir[((size_t)bc * T + fc) * kHalf + 4 * p32];
      const float* src_lo = &fir[((size_t)bc * T + fc) * kHalf + 4 * (31 - p32)];
      const float4 hi4 = *reinterpret_cast<const float4*>(src), lo4 = *reinterpret_cast<const float4*>(src_lo);
      // zeroed by a select on the loaded values (not by a multiplication: 0 * Inf = NaN would leak a clamped row's Inf / NaN into
      // a frame that has to be silent, and into the utterance's scale search)
      const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      v[it + 4] = ok ? hi4 : zero4;
      v[it] = ok ? lo4 : zero4;     // the raw quad A; mirrored when staged
    }
  };
  // the 384 noise samples both frames are cut from, once (reflect padding resolved here), pre-scaled.  Requested BEFORE the
  // tap rows: the vector-memory counter retires in order, so waiting for these two leaves the 16 row loads in flight
  float wv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + 256 * q, i = kHop * (t - 1) + e;
    const bool in = e < 3 * kHop && i >= 0 && i < N + kL - 1;
    const float nv = padded_noise(noise, len, origin, in ? i : origin);   // unconditional: both loads issue back to back
    wv[q] = in ? nv * kNoiseScale : 0.0f;
  }
  float4 cur[8], prv[8];
  load_rows(t, cur);
  load_rows(t - 1, prv);
  L.win[tid] = wv[0];
  L.win[tid + 256] = wv[1];

  // (LDS-only barrier: __syncthreads() would also wait for the tap rows requested above, and the staging of the noise copies
  // below - LDS and vector work only - is what their latency is there to hide)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // win complete

  // reversed noise frames, eight shifted copies each: frame 0: R[u] = f_t[(-u) & 255] = win[128 + ((-u) & 255)];
  // frame 1: R[u] = f_{t-1}[(128 - u) & 255] = win[(128 - u) & 255].  One thread fills 8 consecutive v (one 16-byte chunk).
  // (lane -> chunk with the shift c fastest: the 8 lanes of one v0 read 15 CONSECUTIVE win entries between them and the next
  // group continues 8 further on, so a half-wave's ds_read_b32 touch 32 distinct banks; with v0 fastest the lanes were 8
  // dwords apart - 4 banks, 8-way conflicts - which is where the 0.37 LDS bank-conflict rate of round 2 came from)
#pragma unroll
  for (int fr = 0; fr < 2; ++fr) {           // 2 x 8 copies x 32 blocks = two chunks per thread
    // thread -> (copy c, block): c fastest, the block skewed by sigma(c) = (c - kRot[c]) mod 8 = {0, 0, 5, 2, 7, 0, 5, 2}: the 8 lanes
    // of a ds_write_b128 group then land on 8 distinct 16-byte slots ((block + kRot[c]) mod 8 = (tid / 8 + c) mod 8), and the
    // 32 lanes of a ds_read_b32 group still read 32 distinct banks of `win` (offsets 8 sigma(c) + c mod 32 are distinct within a
    // set of eight, the four sets sit 8 apart)
    const int c = tid & 7, v0 = 8 * (((tid >> 3) + (int)((0x25072500u >> (4 * c)) & 15u)) & 31);
    f16x8 h8, l8;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int u0 = (v0 + 2 * q + c) & 255, u1 = (v0 + 2 * q + 1 + c) & 255;
      const float a = fr == 0 ? L.win[kHop + ((-u0) & 255)] : L.win[(kHop - u0) & 255];
      const float bb = fr == 0 ? L.win[kHop + ((-u1) & 255)] : L.win[(kHop - u1) & 255];
      f16x2 hi, lo;
      split16x2(a, bb, hi, lo);
      h8[2 * q] = hi.x;
      h8[2 * q + 1] = hi.y;
      l8[2 * q] = lo.x;
      l8[2 * q + 1] = lo.y;
    }
    const int vp = (v0 + copy_rot_halfs(c)) & 255;       // the block's rotated place inside its copy
    *reinterpret_cast<f16x8*>(&L.rhi[fr][c][vp]) = h8;
    *reinterpret_cast<f16x8*>(&L.rlo[fr][c][vp]) = l8;
  }
  // one power-of-two scale per utterance (both frames): largest |tap| -> [2^14, 2^15).  Keeps hi AND lo of every tap that
  // matters clear of the fp16 subnormals whatever the filter gain (-120 dB noise floors included); exact to undo.
  float scale[4];   // of row my_row + 8 it (each 32-lane half has its own rows)
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    auto amax4 = [](const float4& a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); };
    float mx = fmaxf(amax4(cur[it + 4]), amax4(prv[it + 4]));     // the upper halves hold every distinct tap of the two rows
    mx = half_max(mx);   // over the 32 lanes that hold this row
    int ex = (int)((__float_as_uint(mx) >> 23) & 0xff);  // biased exponent of the row maximum
    ex = ex < 16 ? 16 : (ex > 250 ? 250 : ex);
    scale[it] = __uint_as_float((unsigned)(268 - ex) << 23);          // 2^(14 - e)
    if (p32 == 0) L.unscale[my_row + 8 * it] = __uint_as_float((unsigned)(ex - 14) << 23) * (1.0f / kNoiseScale);  // 2^(e - 14) / 2^10
  }
  auto stage_rows = [&](const float4 (&v)[8], const int khalf) {  // taps [128 khalf, 128 khalf + 128) of all 32 rows
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = my_row + 8 * it;
      float4 t4 = v[it + 4 * khalf];
      if (khalf == 0) {   // lower half: h[4p + i] = u[128 - 4p - i] from the quad A = u[124 - 4p .. 127 - 4p] (see load_rows)
        float up = dpp_f32<0x138, 0xf>(t4.x);      // wave_shr:1: lane l <- lane l-1 (lane 0 keeps the 0 of `old`)
        up = p32 == 0 ? 0.0f : up;                 // u[128] := 0 (h[0] = 0); also cuts the shift across the two rows of a wave
        t4 = make_float4(up, t4.w, t4.z, t4.y);
      }
      f16x2 h01, l01, h23, l23;
      split16x2(t4.x * scale[it], t4.y * scale[it], h01, l01);
      split16x2(t4.z * scale[it], t4.w * scale[it], h23, l23);
      const f16x4 h = {h01.x, h01.y, h23.x, h23.y}, l = {l01.x, l01.y, l23.x, l23.y};
      *reinterpret_cast<f16x4*>(&L.hhi[r][4 * p32]) = h;
      *reinterpret_cast<f16x4*>(&L.hlo[r][4 * p32]) = l;
    }
  };
  stage_rows(cur, 0);
  __syncthreads();

  f32x16 acc;
  const int j = 32 * wave + col;  // output sample inside the hop
  const int c = (-j) & 7;
  const int jr = j - copy_rot_halfs(c);   // the copy's block rotation folded into the lane's offset (a multiple of 8: the & ~7 commutes)
  auto accumulate = [&](const int fr, const int khalf, auto first_tag) {
    constexpr bool kFirst = decltype(first_tag)::value;
    const _Float16* rh = &L.rhi[fr][c][0];
    const _Float16* rl = &L.rlo[fr][c][0];
#pragma unroll 4
    for (int ks = 0; ks < kL / 32; ++ks) {
      const int k = 16 * ks + 8 * kh;                       // tap inside the staged half
      const int s8 = ((kL / 2 * khalf + k - jr) & 255) & ~7;
      const f16x8 ahi = *reinterpret_cast<const f16x8*>(&L.hhi[col][k]);
      const f16x8 alo = *reinterpret_cast<const f16x8*>(&L.hlo[col][k]);
      const f16x8 bhi = *reinterpret_cast<const f16x8*>(&rh[s8]);
      const f16x8 blo = *reinterpret_cast<const f16x8*>(&rl[s8]);
      if (kFirst && ks == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, f32x16{}, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc, 0, 0, 0);
    }
  };
  accumulate(0, 0, std::true_type{});
  __syncthreads();
  stage_rows(cur, 1);
  __syncthreads();
  accumulate(0, 1, std::false_type{});
  __syncthreads();
  stage_rows(prv, 0);
  __syncthreads();
  accumulate(1, 0, std::false_type{});
  __syncthreads();
  stage_rows(prv, 1);
  // the 16 values of the other branch this lane adds (cat + sum(1)): requested here, all at once, so that they arrive under
  // the last 24 MFMAs (as "if (b < B) out = add_in[o] + v" per row hipcc serialised 16 load -> wait -> store round trips to
  // memory at the end of every wave: 40 % of the kernel's time)
  float addv[16];
  const bool has_add = add_in != nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
    const int b = b0 + row < B ? b0 + row : B - 1;          // clamped: the load is always in bounds, the store is masked
    addv[r] = has_add ? add_in[(size_t)b * N + (size_t)t * kHop + j] : 0.0f;
  }
  __syncthreads();
  accumulate(1, 1, std::false_type{});

  const float ola = t == 0 ? 1.0f : 0.5f;  // overlap-add count: 1 in the first hop, else 2
  float vout[16];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 us = *reinterpret_cast<const float4*>(&L.unscale[8 * g + 4 * kh]);   // rows 8 g + 4 kh + (0..3)
    const float u4[4] = {us.x, us.y, us.z, us.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = acc[4 * g + i] * (ola * u4[i]);
      vout[4 * g + i] = has_add ? addv[4 * g + i] + v : v;
    }
  }
  // every value is final before the first store: 16 stores back to back, nothing to wait for in between (with the load, or
  // a per-row branch, next to each store hipcc put an s_waitcnt vmcnt(0) - the previous store's acknowledgement - in front of it)
  float* o = out + (size_t)(b0 + 4 * kh) * N + (size_t)t * kHop + j;
  if (b0 + kUtt <= B) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2)) * N] = vout[r];
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);
      if (b0 + 4 * kh + row < B) o[(size_t)row * N] = vout[r];
    }
  }
}

}  // namespace

extern "C" int nws_fir_noise_window(const float* fir, const float* noise, int noise_len, int origin, const float* add_in,
                                    int B, int T, float* out, void* stream) {
  if (!fir || !noise || !out || B <= 0 || T <= 0 || noise_len < 2 || origin < 0) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  if (B >= 16) {  // shared-noise circulant GEMM on the matrix cores
    const dim3 grid(8 * ((T + 7) / 8), (B + kUtt - 1) / kUtt);    // see the XCD note in the kernel
    fir_noise_mfma_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(fir, noise, add_in, B, T, noise_len, origin, out);
    NWS_CHECK_LAUNCH();
    return NWS_OK;
  }
  const dim3 grid((T + kHopsPerBlock - 1) / kHopsPerBlock, B);
  fir_noise_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(fir, noise, add_in, T, noise_len, origin, out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

extern "C" int nws_fir_noise(const float* fir, const float* noise, const float* add_in, int B, int T, float* out,
                             void* stream) {
  return nws_fir_noise_window(fir, noise, T * kHop - 1, kL / 2, add_in, B, T, out, stream);
}
