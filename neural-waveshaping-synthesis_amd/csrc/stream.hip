// Stateful streaming step (the counterpart of the reference's stateless scripts/time_buffer_sizes.py:50-72):
// one call = K new control frames of B parallel streams in, 128 K audio samples out, ALL state device-resident behind fixed
// pointers, so that a steady-state hop is a fixed sequence of launches that a hipGraph replays (four for hops of <= 256 samples:
// see nws_stream_step; seven for longer chunks).
//
// What is carried (LABBOOK.md "7"): the GRU state h; the previous chunk's last frame (F0, FiLM row, FIR half-taps) so
// that every kernel of the one-shot forward runs unchanged on the window [previous frame | K new frames]; the float64 phase
// sum through the last emitted sample (spliced exactly into the window's carries); 64 samples of noise-branch residue (the
// noise branch runs half a hop ahead of the oscillator branch); the last 384 samples of the noise stream; and, for the
// reverb, the last 31 999 reverb INPUT samples in a ring.  The learned reverb (models/modules/shaping.py:161-173) is applied
// as a LINEAR convolution of the stream: per hop wet[n] = sum_m ir[m-1] x[n-m] is summed directly in the time domain by 125
// workgroups per utterance (256 taps each) and reduced in a fixed order - 8.2 M MACs per utterance and hop instead of the
// 64 000-point transform pair the overlap-add form paid every hop (round 2: p50 155 us, p99 379 us per 256-sample hop).
// Chunks beyond 2048 samples take the four-step FFT of reverb_fft.hip on [history | new samples].
#include <stdlib.h>

#include "nws_common.h"

namespace {

constexpr int kRing = NWS_STREAM_RING; // reverb-input ring per utterance (>= 31 999 history + the longest chunk)
constexpr int kTapsPerPart = 256;
constexpr int kMaxDirect = 2048;      // samples per step the time-domain reverb serves

size_t al(size_t b) { return (b + 255) & ~size_t(255); }

struct Layout {
  size_t h, h_next, prev_f0, prev_film, prev_fir, S, residue, ring, counters, gru_flag, nzwin;  // state
  size_t gru_out, film_new, fir_new, f0_w, film_w, fir_w, carry, newt_w, noise_w, pre, partial, x_lin, y_lin, rv_ws;   // scratch
  size_t rv_ws_bytes, total;
  int parts;
};

Layout layout(int B, int max_frames, int ir_len, size_t fft_ws_bytes) {
  Layout L{};
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += al(bytes);
    return at;
  };
  const size_t K = max_frames, Tw = K + 1, Nw = 128 * Tw;
  L.h = take((size_t)B * NWS_HIDDEN * 4);
  L.h_next = take((size_t)B * NWS_HIDDEN * 4);
  L.prev_f0 = take((size_t)B * 4);
  L.prev_film = take((size_t)B * NWS_FILM_CH * 4);
  L.prev_fir = take((size_t)B * NWS_FIR_HALF * 4);
  L.S = take((size_t)B * 8);
  L.residue = take((size_t)B * 64 * 4);
  L.ring = take((size_t)B * kRing * 4);
  L.counters = take(64);
  L.gru_flag = take((size_t)B * 8);
  L.nzwin = take((Nw + 128 + 8) * 4);
  L.gru_out = take((size_t)B * K * NWS_HIDDEN * 4);
  L.film_new = take((size_t)B * K * NWS_FILM_CH * 4);
  L.fir_new = take((size_t)B * K * NWS_FIR_HALF * 4);
  L.f0_w = take((size_t)B * Tw * 4);
  L.film_w = take((size_t)B * Tw * NWS_FILM_CH * 4);
  L.fir_w = take((size_t)B * Tw * NWS_FIR_HALF * 4);
  L.carry = take((size_t)B * 4 * Tw * 8);
  L.newt_w = take((size_t)B * Nw * 4);
  L.noise_w = take((size_t)B * Nw * 4);
  L.pre = take((size_t)B * Nw * 4);
  L.parts = (ir_len + kTapsPerPart - 1) / kTapsPerPart;
  L.partial = take((size_t)L.parts * B * (Nw < (size_t)kMaxDirect ? Nw : (size_t)kMaxDirect) * 4);
  if (Nw > (size_t)kMaxDirect) {
    L.x_lin = take((size_t)B * (ir_len + Nw) * 4);
    L.y_lin = take((size_t)B * (ir_len + Nw) * 4);
    L.rv_ws_bytes = fft_ws_bytes;
    L.rv_ws = take(fft_ws_bytes);
  }
  L.total = o;
  return L;
}

// ---- prep: windows, carries spliced with the carried phase sum, previous-frame state, GRU state hand-over, noise window ----
// blocks 0..B-1: one utterance each; block B: the shared noise window
__global__ __launch_bounds__(256) void stream_prep_kernel(const float* __restrict__ f0_new, const float* __restrict__ film_new,
                                                          const float* __restrict__ fir_new, int K, int first, int final, int B,
                                                          float* __restrict__ prev_f0, float* __restrict__ prev_film,
                                                          float* __restrict__ prev_fir, double* __restrict__ S,
                                                          float* __restrict__ h, const float* __restrict__ h_next,
                                                          float* __restrict__ f0_w, float* __restrict__ film_w,
                                                          float* __restrict__ fir_w, double* __restrict__ carry,
                                                          float* __restrict__ nzwin, const float* __restrict__ noise_new,
                                                          int nz_shift, int nz_keep, int n_new, const float* __restrict__ noise_all,
                                                          int noise_all_len, long long* __restrict__ counters) {
  __shared__ double wave_tot[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int Tw = first ? K : K + 1;
  if (b == B) {
    // positions advance here, at the head of the NEXT step: the previous step's last kernel left its sample / frame counts as
    // pending (its other workgroups were still reading the position).  Nobody else in this launch reads the counters.
    const NwsStreamNoiseWin Z{nzwin, noise_new, noise_all, counters, nz_shift, nz_keep, n_new, noise_all_len, first, K};
    nws_stream_noise_window_block<256>(Z, tid);
    return;
  }
  // window = [previous frame] + new frames
  const int off = first ? 0 : 1;
  for (int i = tid; i < Tw; i += 256) f0_w[(size_t)b * Tw + i] = (i < off) ? prev_f0[b] : f0_new[(size_t)b * K + i - off];
  for (int i = tid; i < Tw * NWS_FILM_CH; i += 256) {
    const int t = i / NWS_FILM_CH, c = i - t * NWS_FILM_CH;
    film_w[(size_t)b * Tw * NWS_FILM_CH + i] = (t < off) ? prev_film[(size_t)b * NWS_FILM_CH + c] : film_new[((size_t)b * K + t - off) * NWS_FILM_CH + c];
  }
  for (int i = tid; i < Tw * NWS_FIR_HALF; i += 256) {
    const int t = i / NWS_FIR_HALF, c = i - t * NWS_FIR_HALF;
    fir_w[(size_t)b * Tw * NWS_FIR_HALF + i] = (t < off) ? prev_fir[(size_t)b * NWS_FIR_HALF + c] : fir_new[((size_t)b * K + t - off) * NWS_FIR_HALF + c];
  }
  if (tid < NWS_HIDDEN) h[(size_t)b * NWS_HIDDEN + tid] = h_next[(size_t)b * NWS_HIDDEN + tid];
  __syncthreads();   // the window is complete (global writes of this block are visible to it) and prev_* has been consumed
  if (tid == 0) prev_f0[b] = f0_new[(size_t)b * K + K - 1];
  for (int c = tid; c < NWS_FILM_CH; c += 256) prev_film[(size_t)b * NWS_FILM_CH + c] = film_new[((size_t)b * K + K - 1) * NWS_FILM_CH + c];
  for (int c = tid; c < NWS_FIR_HALF; c += 256) prev_fir[(size_t)b * NWS_FIR_HALF + c] = fir_new[((size_t)b * K + K - 1) * NWS_FIR_HALF + c];
  // exclusive float64 prefix sums of the window's upsampled F0 at 32-sample granularity (the lerp clamps at the window edges;
  // those 64 + 64 samples are only emitted at the true ends of the stream)
  nws_phase_carry_block<4>(f0_w, nullptr, Tw, carry, b, tid, wave_tot);
  __syncthreads();
  const int nch = 4 * Tw;
  double* cb = carry + (size_t)b * nch;
  if (!first) {
    // sample 64 of the window is the first one not yet emitted: its running sum must continue the carried one.  Sums of
    // fp32 values in fp64 are exact, so the spliced carries are bit-identical to the one-shot forward's
    const double base = cb[2], s0 = S[b];
    __syncthreads();
    for (int c = tid; c < nch; c += 256) cb[c] = s0 + (cb[c] - base);
    __syncthreads();
  }
  if (!final && tid == 0) S[b] = cb[nch - 2];    // through the last emitted sample (128 Tw - 64)
}

// pre-reverb value of emitted sample i of this step: nws_common.h (shared with the recurrence launch's extra workgroups)
using PreSrc = NwsPreSrc;
__device__ __forceinline__ float pre_value(const PreSrc& P, int b, int i) { return nws_pre_value(P, b, i); }

// ---- combine (FFT path of long chunks only): pre-reverb signal -> linear buffer + the reverb-input ring -----------------
__global__ __launch_bounds__(256) void stream_combine_kernel(PreSrc P, int M, float* __restrict__ pre, float* __restrict__ ring,
                                                             const long long* __restrict__ counters) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const float v = pre_value(P, b, i);
  pre[(size_t)b * M + i] = v;
  const long long pos = counters[0];
  ring[(size_t)b * kRing + ((pos + i) & (kRing - 1))] = v;
}

// ---- time-domain reverb: partial sums over 256 taps for 256 outputs --------------------------------------------------
// partial[p][b][j] = sum_{i < 256} ir[256 p + i] * x[pos + j - 1 - 256 p - i]      (tap m = 1 + 256 p + i, ir_[m] = ir[m-1])
// x: the ring for samples of earlier steps, pre_value() for this step's own (the ring receives them in the reduce kernel)
__global__ __launch_bounds__(256) void stream_reverb_partial_kernel(const float* __restrict__ ring, PreSrc P, const float* __restrict__ ir,
                                                                    int ir_len, int M, int B, float* __restrict__ partial,
                                                                    const long long* __restrict__ counters) {
  __shared__ __attribute__((aligned(16))) float xs[512];
  __shared__ __attribute__((aligned(16))) float hs[256];
  const int p = blockIdx.x, j0 = blockIdx.y * 256, b = blockIdx.z;
  const int tid = threadIdx.x;
  const float v = nws_stream_reverb_partial<false>(ring, P, ir, ir_len, M, b, p, j0, counters[0], tid, xs, hs);
  if (j0 + tid < M) partial[((size_t)p * B + b) * M + j0 + tid] = v;
}

// ---- reduce (fixed order) + dry signal + everything that closes the step: ring, noise residue, pending counters ----------
// INLINE0 (hops of <= 256 samples: one block per utterance): parts 1 .. parts - 1 were summed by the extra workgroups of the
// hop's FIRST launch (control_gru.hip) - they read earlier hops' reverb input only; part 0, the one that reaches this hop's own
// samples, is summed here by the same code in the same order, so the result is bit-identical with the three-launch form
struct StreamTail {
  const float* film_w;
  const float* fir_w;
  const float* h_next;
  float* prev_film;
  float* prev_fir;
  float* h;
  int Tw;
  // four-launch hop: nobody has applied the previous step's pending counts (the workgroup that does so in the five-launch hop
  // shares its launch with readers of the pending form there): positions are read as counters[0] + counters[2], and the LAST
  // workgroup of this kernel to finish leaves the counters with nothing pending
  int apply_counters;
};
template <bool INLINE0>
__global__ __launch_bounds__(256) void stream_reverb_reduce_kernel(const float* __restrict__ partial, int parts, PreSrc P, int M, int B,
                                                                   int K, int final, int tail_from, float* __restrict__ out,
                                                                   float* __restrict__ pre_out, float* __restrict__ ring,
                                                                   float* __restrict__ residue, long long* __restrict__ counters,
                                                                   const float* __restrict__ ir, int ir_len, StreamTail tail) {
  const int b = blockIdx.y;
  if (INLINE0 && tail.film_w != nullptr) {
    // fused hop (no prep kernel): what the next hop's head needs of this one - the last frame's FiLM row and FIR taps, the GRU state
    const int c = threadIdx.x;
    tail.prev_film[(size_t)b * NWS_FILM_CH + c] = tail.film_w[((size_t)b * tail.Tw + tail.Tw - 1) * NWS_FILM_CH + c];
    if (c < NWS_FIR_HALF) tail.prev_fir[(size_t)b * NWS_FIR_HALF + c] = tail.fir_w[((size_t)b * tail.Tw + tail.Tw - 1) * NWS_FIR_HALF + c];
    if (c < NWS_HIDDEN) tail.h[(size_t)b * NWS_HIDDEN + c] = tail.h_next[(size_t)b * NWS_HIDDEN + c];
  }
  const int j = blockIdx.x * 256 + threadIdx.x;
  const bool apply = INLINE0 && tail.apply_counters != 0;
  const long long pos = counters[0] + (apply ? counters[2] : 0);
  float part0 = 0.0f;
  if (INLINE0) {
    __shared__ __attribute__((aligned(16))) float xs[512];
    __shared__ __attribute__((aligned(16))) float hs[256];
    part0 = nws_stream_reverb_partial<false>(ring, P, ir, ir_len, M, b, 0, 0, pos, threadIdx.x, xs, hs);
  }
  if (j < M) {
    const float dry = pre_value(P, b, j);
    // 125 independent loads: five fixed-order groups of 25 in flight at a time (one dependent load per iteration made this
    // kernel the longest of the hop)
    float acc = 0.0f;
    int p = 0;
    for (; p + 25 <= parts; p += 25) {
      float t[25];
#pragma unroll
      for (int q = 0; q < 25; ++q) t[q] = (INLINE0 && p + q == 0) ? part0 : partial[((size_t)(p + q) * B + b) * M + j];
#pragma unroll
      for (int q = 0; q < 25; ++q) acc += t[q];
    }
    for (; p < parts; ++p) acc += (INLINE0 && p == 0) ? part0 : partial[((size_t)p * B + b) * M + j];
    out[(size_t)b * M + j] = dry + acc;
    pre_out[(size_t)b * M + j] = dry;
    ring[(size_t)b * kRing + ((pos + j) & (kRing - 1))] = dry;
  }
  // the noise residue of the next chunk: thread j < 64 of the first block is the only one that read residue[b][j] above
  if (!final && blockIdx.x == 0 && threadIdx.x < 64) residue[(size_t)b * 64 + threadIdx.x] = P.noise_w[(size_t)b * P.Nw + tail_from + threadIdx.x];
  if (apply) {
    // every workgroup has read the position (above, into `pos`) before it counts itself done; the one that completes the count
    // is the only one left that touches the counters
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(&counters[4]), 1ull);
      if (done == (unsigned long long)(gridDim.x * gridDim.y) - 1) {
        __threadfence();
        counters[0] = pos + M;
        counters[1] = counters[1] + counters[3] + K;
        counters[2] = 0;
        counters[3] = 0;
        counters[4] = 0;
      }
    }
    return;
  }
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {   // applied by the next step's prep kernel
    counters[2] = M;
    counters[3] = K;
  }
}

__global__ void stream_close_kernel(const float* __restrict__ noise_w, int Nw, int tail_from, int final, int B, int M, int K,
                                    float* __restrict__ residue, long long* __restrict__ counters) {
  const int b = blockIdx.x;
  if (b < B) {
    if (!final && threadIdx.x < 64) residue[(size_t)b * 64 + threadIdx.x] = noise_w[(size_t)b * Nw + tail_from + threadIdx.x];
    return;
  }
  if (threadIdx.x == 0) {
    counters[2] = M;
    counters[3] = K;
  }
}

// ---- FFT path for long chunks: [31 999 samples of history | M new samples] from the ring; the tail of the result back out ----
// after_step: the step's pending sample count has not been applied to counters[0] yet (see stream_prep_kernel)
__global__ __launch_bounds__(256) void stream_gather_kernel(const float* __restrict__ ring, int hist, int n_new, int zero_from,
                                                            int after_step, float* __restrict__ x_lin,
                                                            const long long* __restrict__ counters) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int N = hist + n_new;
  if (i >= N) return;
  const long long a = counters[0] + (after_step ? counters[2] : 0) - hist + i;    // absolute sample index
  x_lin[(size_t)b * N + i] = (a >= 0 && i < zero_from) ? ring[(size_t)b * kRing + (a & (kRing - 1))] : 0.0f;
}

__global__ __launch_bounds__(256) void stream_take_kernel(const float* __restrict__ y_lin, int N, int from, int M, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < M) out[(size_t)b * M + i] = y_lin[(size_t)b * N + from + i];
}

__global__ void stream_reset_kernel(char* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0;
}

}  // namespace

extern "C" {

size_t nws_stream_state_bytes(int B, int max_frames, int ir_len, const NwsReverbPlan* plan) {
  if (B <= 0 || max_frames <= 0 || ir_len <= 0) return 0;
  return layout(B, max_frames, ir_len, plan ? nws_reverb_workspace_bytes(plan, B) : 0).total;
}

int nws_stream_reset(void* state, size_t state_bytes, void* stream) {
  if (!state || state_bytes == 0) return NWS_ERR_BAD_ARG;
  stream_reset_kernel<<<512, 256, 0, (hipStream_t)stream>>>(static_cast<char*>(state), state_bytes);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_stream_out_samples(int K, int first, int final) {
  const int Tw = first ? K : K + 1;
  return (final ? 128 * Tw : 128 * Tw - 64) - (first ? 0 : 64);
}

// One chunk of K frames.  frames_seen: control frames pushed before this chunk; nz_prev_start: absolute index of the first
// sample the previous step left in the noise window (nws_stream_noise_start of that step; 0 for the first).  The host
// mirrors these counters: they only decide launch arguments that are the same in every steady-state step (previous and
// current chunk of the same K, not first, not final), so a captured graph of such a step stays valid; positions that DO
// advance every step (ring position, absolute frame index of an injected noise stream) live in device memory.
// noise_new (drawn stream): the next nws_stream_noise_draws(...) samples of the excitation; noise_all (parity runs): the
// reference's whole draw (128 F - 1 samples); exactly one of the two.  pre_out: optional (B, M) tap of the reverb input.
// plan / tables / spectrum: the L = 64 000 reverb plan, needed only for chunks beyond 2048 samples (else NULL).
long long nws_stream_noise_start(int first, long long frames_seen) {
  const long long A0 = first ? 0 : frames_seen - 1;
  return A0 <= 0 ? 0 : 128 * (A0 - 1);
}

int nws_stream_noise_draws(int K, int first, long long frames_seen) {
  const long long A0 = first ? 0 : frames_seen - 1;
  const int Tw = first ? K : K + 1;
  const long long need_upto = 128 * (A0 + Tw - 1) + 129;
  const long long have = first ? 0 : 128 * (frames_seen - 1) + 129;
  return (int)(need_upto - have);
}

int nws_stream_step(const NwsWeights* w, const float* fir_design, const NwsReverbPlan* plan, const void* reverb_tables,
                    const void* reverb_spectrum, void* state, size_t state_bytes, int B, int max_frames, const float* f0,
                    const float* control, int C, int K, int first, int final, long long frames_seen, long long nz_prev_start,
                    float sample_rate, const float* phase_u, const float* rand_phase, const float* noise_new,
                    const float* noise_all, int noise_all_len, const float* ir, int ir_len, float* out, float* pre_out,
                    void* stream) {
  if (!w || !fir_design || !state || !f0 || !control || !phase_u || !rand_phase || !ir || !out) return NWS_ERR_BAD_ARG;
  if (B <= 0 || K <= 0 || K > max_frames || C < 2 || ir_len <= 0 || ir_len >= kRing / 2) return NWS_ERR_BAD_ARG;
  if ((noise_new == nullptr) == (noise_all == nullptr)) return NWS_ERR_BAD_ARG;
  if ((first != 0) != (frames_seen == 0)) return NWS_ERR_BAD_ARG;
  if (B > 65535) return NWS_ERR_UNSUPPORTED;
  const Layout L = layout(B, max_frames, ir_len, plan ? nws_reverb_workspace_bytes(plan, B) : 0);
  if (L.total > state_bytes) return NWS_ERR_WORKSPACE;
  char* base = static_cast<char*>(state);
  auto F = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
  long long* counters = reinterpret_cast<long long*>(base + L.counters);
  hipStream_t st = (hipStream_t)stream;
  const int Tw = first ? K : K + 1, Nw = 128 * Tw;
  const int lo = first ? 0 : 64, hi = final ? Nw : Nw - 64, M = hi - lo;
  if (M <= 0) return NWS_ERR_BAD_ARG;
  int rc;
  // 1. control encoder on the new frames (state carried)   2. frame MLPs
  // a hop of <= 256 samples: the reverb's history parts (all but the first: earlier hops' reverb input only) ride on the
  // recurrence launch as extra workgroups, off the hop's critical path (NWS_STREAM_SPLIT_REVERB=0: the three-launch form)
  static const bool split_env = [] {
    const char* e = getenv("NWS_STREAM_SPLIT_REVERB");
    return e == nullptr || e[0] != '0';
  }();
  const bool split_reverb = split_env && M <= 256 && L.parts >= 2;
  // ... and then nothing is left for a prep launch either (NWS_STREAM_FUSE_HEAD=0 keeps it): the per-utterance head (F0 window, first
  // rows of the FiLM / tap windows, spliced carries) depends on nothing the hop computes - more workgroups of the recurrence launch;
  // the frame-MLP kernel writes its rows straight into the windows and carries one workgroup for the shared noise window and the
  // pending counters; the closing kernel hands the last frame and the GRU state to the next hop.  Five launches.
  static const bool head_env = [] {
    const char* e = getenv("NWS_STREAM_FUSE_HEAD");
    return e == nullptr || e[0] != '0';
  }();
  const bool fuse_head = head_env && split_reverb && K <= 32 && w->mlp_frags != nullptr;
  // ... and with one or two new frames the frame MLPs join the recurrence launch as well (two workgroups per utterance that fetch
  // their first weights while the recurrence runs and wait for its rows; NWS_STREAM_FUSE_MLP=0 keeps their launch).  Four launches.
  static const bool mlp_env = [] {
    const char* e = getenv("NWS_STREAM_FUSE_MLP");
    const char* f = getenv("NWS_MLP_FEW");
    return (e == nullptr || e[0] != '0') && (f == nullptr || f[0] != '0');
  }();
  const bool fuse_mlp = mlp_env && fuse_head && K <= 2;
  const long long nz_start = nws_stream_noise_start(first, frames_seen);
  const long long have = first ? 0 : 128 * (frames_seen - 1) + 129;       // absolute end of what the window holds now
  const int n_new = noise_new ? nws_stream_noise_draws(K, first, frames_seen) : 0;
  const int nz_keep = first ? 0 : (int)(have - nz_start);
  const int nz_shift = first ? 0 : (int)(nz_start - nz_prev_start);
  if (nz_shift < 0 || nz_keep < 0) return NWS_ERR_BAD_ARG;
  NwsStreamSide side{};
  side.B = B;
  if (split_reverb) {
    side.ring = F(L.ring);
    side.ir = ir;
    side.partial = F(L.partial);
    side.counters = counters;
    side.ir_len = ir_len;
    side.M = M;
    side.parts = L.parts;
  }
  if (fuse_head) {
    side.f0_new = f0;
    side.prev_f0 = F(L.prev_f0);
    side.prev_film = F(L.prev_film);
    side.prev_fir = F(L.prev_fir);
    side.S = reinterpret_cast<double*>(base + L.S);
    side.f0_w = F(L.f0_w);
    side.film_w = F(L.film_w);
    side.fir_w = F(L.fir_w);
    side.carry = reinterpret_cast<double*>(base + L.carry);
    side.K = K;
    side.first = first;
    side.final = final;
  }
  if (fuse_mlp) {
    side.gru_flag = reinterpret_cast<long long*>(base + L.gru_flag);
    side.counters = counters;
    side.counters_rw = counters;
    side.gru_out = F(L.gru_out);
    side.out_T = Tw;
    side.out_off = first ? 0 : 1;
    side.win = NwsStreamNoiseWin{F(L.nzwin), noise_new, noise_all, counters, nz_shift, nz_keep, n_new, noise_all_len, first, K};
  }
  rc = nws_control_gru_stream(w, control, B, C, K, first ? nullptr : F(L.h), F(L.gru_out), F(L.h_next), split_reverb ? &side : nullptr,
                              stream);
  if (rc != NWS_OK) return rc;
  if (fuse_mlp) {
    // (nothing: the frame MLPs, the noise window and the windows' rows were part of the first launch)
  } else if (fuse_head) {
    const NwsStreamNoiseWin Z{F(L.nzwin), noise_new, noise_all, counters, nz_shift, nz_keep, n_new, noise_all_len, first, K};
    rc = nws_frame_mlps_stream(w, F(L.gru_out), B, K, F(L.film_w), F(L.fir_w), Tw, first ? 0 : 1, &Z, stream);
    if (rc != NWS_OK) return rc;
  } else {
    rc = nws_frame_mlps(w, F(L.gru_out), fir_design, B, K, nullptr, F(L.film_new), nullptr, F(L.fir_new), stream);
    if (rc != NWS_OK) return rc;
    // 3. windows, spliced carries, state hand-over, noise window
    stream_prep_kernel<<<B + 1, 256, 0, st>>>(f0, F(L.film_new), F(L.fir_new), K, first, final, B, F(L.prev_f0), F(L.prev_film),
                                             F(L.prev_fir), reinterpret_cast<double*>(base + L.S), F(L.h), F(L.h_next), F(L.f0_w),
                                             F(L.film_w), F(L.fir_w), reinterpret_cast<double*>(base + L.carry), F(L.nzwin), noise_new,
                                             nz_shift, nz_keep, n_new, noise_all, noise_all_len, counters);
    NWS_CHECK_LAUNCH();
  }
  // 4. oscillator + waveshapers on the window   5. noise branch on the same window
  rc = nws_exciter_newt(w, F(L.f0_w), nullptr, reinterpret_cast<double*>(base + L.carry), phase_u, rand_phase, F(L.film_w), B, Tw,
                        sample_rate, nullptr, F(L.newt_w), stream);
  if (rc != NWS_OK) return rc;
  // frame t of the window covers nzwin[128 t - origin, +256): the stream's first frame reaches 128 samples before its start
  // (reflected like torch.stft, generators.py:31); the one-shot draw has 128 F - 1 samples: its end reflects in the final chunk
  const long long A0 = first ? 0 : frames_seen - 1;
  const int origin = A0 == 0 ? 128 : 0;
  const int n_have = (int)(128 * (A0 + Tw - 1) + 129 - nz_start);
  const int n_len = final ? (int)(128 * (A0 + Tw) - 1 - nz_start) : n_have;
  rc = nws_fir_noise_window(F(L.fir_w), F(L.nzwin), n_len, origin, nullptr, B, Tw, F(L.noise_w), stream);
  if (rc != NWS_OK) return rc;
  // 6./7. emitted samples (oscillator branch [lo, hi) + noise branch: 64 samples of residue first, then this window's hops)
  // and the linear reverb of the stream
  const int R0 = first ? 0 : 64, noise_off = first ? 0 : 128;
  const PreSrc P{F(L.newt_w), F(L.noise_w), F(L.residue), Nw, lo, R0, noise_off};
  float* pre = pre_out ? pre_out : F(L.pre);
  const int tail_from = noise_off + M - R0;
  if (split_reverb) {
    // one launch: part 0 + the fixed-order reduction over it and the parts the first launch left, closing the step
    StreamTail tail{};
    if (fuse_head) tail = StreamTail{F(L.film_w), F(L.fir_w), F(L.h_next), F(L.prev_film), F(L.prev_fir), F(L.h), Tw, fuse_mlp ? 1 : 0};
    stream_reverb_reduce_kernel<true><<<dim3(1, B), 256, 0, st>>>(F(L.partial), L.parts, P, M, B, K, final, tail_from, out, pre, F(L.ring),
                                                                  F(L.residue), counters, ir, ir_len, tail);
    NWS_CHECK_LAUNCH();
  } else if (M <= kMaxDirect) {
    // two launches: 125 x B workgroups of partial sums, then the fixed-order reduction that also closes the step
    stream_reverb_partial_kernel<<<dim3(L.parts, (M + 255) / 256, B), 256, 0, st>>>(F(L.ring), P, ir, ir_len, M, B, F(L.partial), counters);
    NWS_CHECK_LAUNCH();
    stream_reverb_reduce_kernel<false><<<dim3((M + 255) / 256, B), 256, 0, st>>>(F(L.partial), L.parts, P, M, B, K, final, tail_from, out,
                                                                                 pre, F(L.ring), F(L.residue), counters, ir, ir_len,
                                                                                 StreamTail{});
    NWS_CHECK_LAUNCH();
  } else {
    if (!plan || !reverb_tables || !reverb_spectrum || L.x_lin == 0) return NWS_ERR_BAD_ARG;
    const int N = ir_len + M;
    if (N > plan->L || plan->Lc != 0) return NWS_ERR_BAD_ARG;
    stream_combine_kernel<<<dim3((M + 255) / 256, B), 256, 0, st>>>(P, M, pre, F(L.ring), counters);
    NWS_CHECK_LAUNCH();
    stream_gather_kernel<<<dim3((N + 255) / 256, B), 256, 0, st>>>(F(L.ring), ir_len, M, N, 0, F(L.x_lin), counters);
    NWS_CHECK_LAUNCH();
    rc = nws_reverb(plan, reverb_tables, reverb_spectrum, F(L.x_lin), B, N, F(L.y_lin), base + L.rv_ws, L.rv_ws_bytes, stream);
    if (rc != NWS_OK) return rc;
    stream_take_kernel<<<dim3((M + 255) / 256, B), 256, 0, st>>>(F(L.y_lin), N, ir_len, M, out);
    NWS_CHECK_LAUNCH();
    stream_close_kernel<<<B + 1, 64, 0, st>>>(F(L.noise_w), Nw, tail_from, final, B, M, K, F(L.residue), counters);
    NWS_CHECK_LAUNCH();
  }
  return NWS_OK;
}

// What a linear reverb still rings out after the last chunk: tail[k] = sum_m ir_[m] x[end + k - m], k < ir_len + 1, from the
// ring's history (four-step FFT on [history | zeros]); plan: the L = 64 000 plan.
int nws_stream_reverb_tail(const NwsReverbPlan* plan, const void* reverb_tables, const void* reverb_spectrum, void* state,
                           size_t state_bytes, int B, int max_frames, int ir_len, float* tail_out /* (B, ir_len + 1) */,
                           void* workspace, size_t workspace_bytes, void* stream) {
  if (!plan || !reverb_tables || !reverb_spectrum || !state || !tail_out || !workspace || B <= 0) return NWS_ERR_BAD_ARG;
  // (the state blob may have been sized without the FFT scratch of long chunks: only the state part is touched here)
  const Layout L = layout(B, max_frames, ir_len, 0);
  if (L.nzwin > state_bytes) return NWS_ERR_WORKSPACE;
  const int N = 2 * ir_len + 1;
  if (N > plan->L || plan->Lc != 0) return NWS_ERR_BAD_ARG;
  const size_t lin = al((size_t)B * N * sizeof(float));
  const size_t rvb = nws_reverb_workspace_bytes(plan, B);
  if (workspace_bytes < 2 * lin + rvb) return NWS_ERR_WORKSPACE;
  char* base = static_cast<char*>(state);
  char* ws = static_cast<char*>(workspace);
  float* x_lin = reinterpret_cast<float*>(ws);
  float* y_lin = reinterpret_cast<float*>(ws + lin);
  hipStream_t st = (hipStream_t)stream;
  // the last step's sample count is still pending: history = [pos + pending - ir_len, pos + pending), then ir_len + 1 zeros
  stream_gather_kernel<<<dim3((N + 255) / 256, B), 256, 0, st>>>(reinterpret_cast<float*>(base + L.ring), ir_len, ir_len + 1, ir_len, 1,
                                                                 x_lin, reinterpret_cast<long long*>(base + L.counters));
  NWS_CHECK_LAUNCH();
  const int rc = nws_reverb(plan, reverb_tables, reverb_spectrum, x_lin, B, N, y_lin, ws + 2 * lin, rvb, stream);
  if (rc != NWS_OK) return rc;
  stream_take_kernel<<<dim3((ir_len + 1 + 255) / 256, B), 256, 0, st>>>(y_lin, N, ir_len, ir_len + 1, tail_out);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
