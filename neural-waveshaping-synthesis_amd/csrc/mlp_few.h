// Frame MLPs for a handful of frames (a streaming hop: 1 .. 4 new control frames per utterance).
//
//   emb = proj(gru_out);  film = newt.mlp(emb);  H = h_generator(emb);  fir = D H     (frame_mlps.hip has the citations)
//
// frame_mlps16_kernel runs two frames through seven dependent MFMA phases of ONE workgroup (32-frame tiles, 745 KB of weight
// fragments through one CU): 14.6 us of a 60 us hop.  With so few frames the work is reading the weights once: this form is
// a matrix-vector product per layer - one workgroup of four waves per (utterance, path), lane (row, half) of wave mt owns output
// channel 32 mt + row and the k of its half, reads its weights as the SAME fp16 (hi, lo) fragment pairs the MFMA kernels use
// (NwsWeights.mlp_frags, first table: lane li of fragment (mt, ks) = row li & 31, k = 16 ks + 8 (li >> 5) + 0..7; hi + lo is
// the weight to 22 bits), a layer ahead of its use, against fp32 activations in LDS (NF frames side by side per channel: one
// packed FMA per weight and frame pair).  Exact fp32 products of 22-bit weights: the class of the two-term MFMA form, not
// bit-identical with it (tests/test_gpu_streaming.py compares both forms with the same CPU reference).
// newt.mlp and h_generator are independent after the embedding: one workgroup each (both compute proj).
// Where its ~10 us go (tools/mlp_few_timeline.py, cycles of one workgroup): a hidden layer 3 700 = ~1 000 for requesting the next
// layer's 64 KB of fragments (four waves through one CU's 64 B / clock), ~1 600 for the product (32 ds_read_b128 per lane and product:
// the four waves' 128 KB of activation reads are the LDS pipe's 1 000 cycles - lanes of a half read the same address, the
// bandwidth is per lane all the same), ~1 100 LayerNorm (two dependent 32-lane reductions, two barriers).  A rolled loop over the
// layers (a quarter of the code) measured SLOWER (26-29 K against 19-22 K cycles: moving the prefetched fragments into the loop's
// registers waits for them) - it is not instruction fetch.  A K-split over the waves (every wave all four M-tiles of a quarter of K: a
// quarter of the LDS reads, one more exchange per layer) is what is left; not built.
#pragma once

#include "nws_common.h"

typedef _Float16 nws_f16x8 __attribute__((ext_vector_type(8)));

// first fragment table of NwsWeights.mlp_frags (frame_mlps.hip: frag_map), units of 16 bytes:
// 0 proj | 1-3 newt hidden | 4 newt out (8 M-tiles) | 5-7 hgen hidden | 8 hgen out (129 rows -> 5 M-tiles) | 9 FIR design rows 128..255 (9 K-steps)
__host__ __device__ constexpr int nws_few_frag_base(int id) {
  return id == 0 ? 0 : id <= 3 ? 4096 * id : id == 4 ? 16384 : id <= 7 ? 24576 + 4096 * (id - 5) : id == 8 ? 36864 : 41984;
}

constexpr int kFewK = 144;   // activation rows in LDS: 128 channels, 129 bands padded to the FIR design's 9 K-steps

template <int KS>
struct NwsFewFrag {
  nws_f16x8 hi[KS], lo[KS];
};

template <int KS>
__device__ __forceinline__ void nws_few_load(NwsFewFrag<KS>& A, const nws_f16x8* __restrict__ frags, int base, int mt, int li) {
  const nws_f16x8* a = frags + base + (size_t)mt * KS * 128 + li;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    A.hi[ks] = a[ks * 128];
    A.lo[ks] = a[ks * 128 + 64];
  }
}

// fp32 value of fragment element j: hi[j] + lo[j] in ONE v_fma_mix_f32 (f16 * 1.0 + f16, exact: 22 significant bits) instead of
// two conversions and an add - per weight that is a third of the vector work of a layer
template <int J>
__device__ __forceinline__ float nws_few_weight(const nws_f16x8& hi, const nws_f16x8& lo) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const unsigned h = __builtin_bit_cast(u32x4, hi)[J >> 1], l = __builtin_bit_cast(u32x4, lo)[J >> 1];
  float d;
  if (J & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(l));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(h), "v"(l));
  return d;
}

template <int CTRL>
__device__ __forceinline__ float nws_few_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// NF = 2: accumulators as one pair; NF = 4: two pairs.  xs[k][NF] fp32.
template <int NF>
struct NwsFewAcc {
  f32x2 p[NF / 2];
};

// acc[f] = sum over this lane's k of w[k] x[k][f]; then the two halves of a row are added: every lane of a row holds the row's sum
template <int KS, int NF>
__device__ __forceinline__ void nws_few_dot(const NwsFewFrag<KS>& A, const float* __restrict__ xs, int half, NwsFewAcc<NF>& acc) {
#pragma unroll
  for (int q = 0; q < NF / 2; ++q) acc.p[q] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float* xk = xs + (16 * ks + 8 * half) * NF;
    const float wv[8] = {nws_few_weight<0>(A.hi[ks], A.lo[ks]), nws_few_weight<1>(A.hi[ks], A.lo[ks]), nws_few_weight<2>(A.hi[ks], A.lo[ks]),
                         nws_few_weight<3>(A.hi[ks], A.lo[ks]), nws_few_weight<4>(A.hi[ks], A.lo[ks]), nws_few_weight<5>(A.hi[ks], A.lo[ks]),
                         nws_few_weight<6>(A.hi[ks], A.lo[ks]), nws_few_weight<7>(A.hi[ks], A.lo[ks])};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 w2 = splat2(wv[j]);
      if (NF == 2) {
        acc.p[0] = fma2(w2, *reinterpret_cast<const f32x2*>(xk + j * NF), acc.p[0]);
      } else {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xk + j * NF);
        acc.p[0] = fma2(w2, f32x2{x4.x, x4.y}, acc.p[0]);
        acc.p[1] = fma2(w2, f32x2{x4.z, x4.w}, acc.p[1]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NF / 2; ++q) {
    acc.p[q].x += nws_swap_halves(acc.p[q].x);
    acc.p[q].y += nws_swap_halves(acc.p[q].y);
  }
}

// sum over the 32 lanes of a half (both halves hold the same values): four DPP steps inside the rows of 16 (quad swaps, then
// rotations by 4 and 8: every lane ends with its row's sum), one cross-row exchange
__device__ __forceinline__ float nws_few_sum32(float v) {
  v += nws_few_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += nws_few_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += nws_few_dpp<0x124>(v);   // row_ror:4
  v += nws_few_dpp<0x128>(v);   // row_ror:8
  return v + __shfl_xor(v, 16, 64);
}

struct NwsFewLds {
  float xa[kFewK * 4];
  float xb[kFewK * 4];
  float red[2][4][4];     // [mean | M2][wave][frame]
};

// One hidden layer: v = W x + b -> LayerNorm over the 128 channels of each frame (per-wave mean / M2, Chan's merge: the
// scheme of frame_mlps.hip) -> LeakyReLU(0.01) -> xout[channel][frame].  Ends with a barrier.
template <int NF>
__device__ __forceinline__ void nws_few_hidden(NwsFewLds& L, const NwsFewFrag<8>& A, const float* xin, float* xout, float bo, float go,
                                               float bto, int mt, int row, int half) {
  const int o = 32 * mt + row;
  NwsFewAcc<NF> acc;
  nws_few_dot<8, NF>(A, xin, half, acc);
  float v[NF], mw[NF], m2[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    v[f] = ((f & 1) ? acc.p[f >> 1].y : acc.p[f >> 1].x) + bo;
    mw[f] = nws_few_sum32(v[f]) * (1.0f / 32.0f);
    const float d = v[f] - mw[f];
    m2[f] = nws_few_sum32(d * d);
  }
  if (row == 0 && half == 0) {
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      L.red[0][mt][f] = mw[f];
      L.red[1][mt][f] = m2[f];
    }
  }
  __syncthreads();
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const float m0 = L.red[0][0][f], m1 = L.red[0][1][f], m2w = L.red[0][2][f], m3 = L.red[0][3][f];
    const float mean = ((m0 + m1) + (m2w + m3)) * 0.25f;
    const float d0 = m0 - mean, d1 = m1 - mean, d2 = m2w - mean, d3 = m3 - mean;
    const float between = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    const float within = (L.red[1][0][f] + L.red[1][1][f]) + (L.red[1][2][f] + L.red[1][3][f]);
    const float var = fmaf(32.0f, between, within) * (1.0f / NWS_HIDDEN);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float y = (v[f] - mean) * rstd * go + bto;
    v[f] = fmaxf(y, 0.01f * y);
  }
  if (half == 0) {
    if (NF == 2) *reinterpret_cast<f32x2*>(xout + o * NF) = f32x2{v[0], v[1]};
    else *reinterpret_cast<f32x4*>(xout + o * NF) = f32x4{v[0], v[1], v[NF > 2 ? 2 : 0], v[NF > 2 ? 3 : 0]};
  }
  __syncthreads();
}

// path 0: proj -> newt.mlp -> FiLM rows;  path 1: proj -> h_generator -> H -> FIR half-taps.  256 threads; T <= NF frames of
// utterance b; rows go to row out_off + t of windows of out_T rows per utterance.
// `wait_for_input()` runs behind the first two layers' fragment requests and in front of the first read of gru_out (returns true at
// once for a launch of its own; the role inside the recurrence launch waits for its utterance's recurrence there, false = gave up).
template <int NF, typename Wait>
__device__ __forceinline__ void nws_mlp_few_path(NwsFewLds& L, const NwsWeights& w, const float* __restrict__ gru_out, int T, int b,
                                                 int path, float* __restrict__ film_w, float* __restrict__ fir_w, int out_T,
                                                 int out_off, int tid, Wait&& wait_for_input, long long* probe = nullptr) {
  // probe (measurements: tools/mlp_few_timeline.py): s_memtime of thread 0 of utterance 0 at the phase boundaries, 16 slots per path
  int pslot = 0;
  auto tick = [&] {
    if (probe != nullptr && b == 0 && tid == 0) probe[16 * path + pslot] = (long long)__builtin_readcyclecounter();
    ++pslot;
  };
  tick();
  const nws_f16x8* F = reinterpret_cast<const nws_f16x8*>(w.mlp_frags);
  const int li = tid & 63, mt = tid >> 6, row = li & 31, half = li >> 5;
  // this lane's biases and LayerNorm gains of every layer FIRST: a value loaded behind a layer's fragment requests is the youngest
  // load in flight when it is needed, and waiting for it (vmcnt(0)) waits for the whole next layer's fragments - the prefetch gone
  const float* const* hb = path ? w.hgen_b : w.newt_mlp_b;
  const float* const* hg = path ? w.hgen_ln_g : w.newt_ln_g;
  const float* const* hbt = path ? w.hgen_ln_b : w.newt_ln_b;
  const int oc = 32 * mt + row;
  const float b_proj = w.proj_b[oc];
  const float b_h[3] = {hb[0][oc], hb[1][oc], hb[2][oc]};
  const float g_h[3] = {hg[0][oc], hg[1][oc], hg[2][oc]};
  const float t_h[3] = {hbt[0][oc], hbt[1][oc], hbt[2][oc]};
  const float b_out0 = hb[3][oc];
  const float b_out1 = path ? w.hgen_b[3][128] : w.newt_mlp_b[3][oc + 128];
  NwsFewFrag<8> A, An;
  nws_few_load<8>(A, F, nws_few_frag_base(0), mt, li);                      // proj
  nws_few_load<8>(An, F, nws_few_frag_base(path ? 5 : 1), mt, li);          // first hidden layer
  tick();
  // (false: the input never arrived - the role's bounded wait gave up.  The rows of this hop are then NaN, loudly, not stale)
  const bool have_input = wait_for_input();
  // gru_out rows -> xa[k][f]; zero the K padding of both buffers (rows 128 .. 143: the FIR design contracts over 144)
  for (int e = tid; e < NWS_HIDDEN * NF; e += 256) {
    const int k = e / NF, f = e - k * NF;
    L.xa[e] = !have_input ? __builtin_nanf("") : f < T ? gru_out[((size_t)b * T + f) * NWS_HIDDEN + k] : 0.0f;
  }
  for (int e = tid; e < (kFewK - NWS_HIDDEN) * NF; e += 256) {
    L.xa[NWS_HIDDEN * NF + e] = 0.0f;
    L.xb[NWS_HIDDEN * NF + e] = 0.0f;
  }
  __syncthreads();
  tick();
  {
    // emb = proj(gru_out) -> xb (no LayerNorm)
    const int o = 32 * mt + row;
    NwsFewAcc<NF> acc;
    nws_few_dot<8, NF>(A, L.xa, half, acc);
    const float bo = b_proj;
    if (half == 0) {
      if (NF == 2) *reinterpret_cast<f32x2*>(L.xb + o * NF) = f32x2{acc.p[0].x + bo, acc.p[0].y + bo};
      else *reinterpret_cast<f32x4*>(L.xb + o * NF) = f32x4{acc.p[0].x + bo, acc.p[0].y + bo, acc.p[NF / 2 - 1].x + bo, acc.p[NF / 2 - 1].y + bo};
    }
    __syncthreads();
  }
  tick();
  const int id0 = path ? 5 : 1;
  // three hidden layers: xb -> xa -> xb -> xa, each with the next layer's fragments in flight
  A = An;
  nws_few_load<8>(An, F, nws_few_frag_base(id0 + 1), mt, li);
  nws_few_hidden<NF>(L, A, L.xb, L.xa, b_h[0], g_h[0], t_h[0], mt, row, half);
  tick();
  A = An;
  nws_few_load<8>(An, F, nws_few_frag_base(id0 + 2), mt, li);
  nws_few_hidden<NF>(L, A, L.xa, L.xb, b_h[1], g_h[1], t_h[1], mt, row, half);
  tick();
  A = An;
  nws_few_load<8>(An, F, nws_few_frag_base(id0 + 3), mt, li);                // output layer, M-tile mt
  nws_few_hidden<NF>(L, A, L.xb, L.xa, b_h[2], g_h[2], t_h[2], mt, row, half);
  tick();
  A = An;
  if (path == 0) {
    // FiLM rows: 256 channels = M-tiles mt and mt + 4 from xa
    nws_few_load<8>(An, F, nws_few_frag_base(4), mt + 4, li);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int o = 32 * (mt + 4 * pass) + row;
      NwsFewAcc<NF> acc;
      nws_few_dot<8, NF>(pass ? An : A, L.xa, half, acc);
      const float bo = pass ? b_out1 : b_out0;
      if (half == 0) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
          if (f < T) film_w[((size_t)b * out_T + out_off + f) * NWS_FILM_CH + o] = ((f & 1) ? acc.p[f >> 1].y : acc.p[f >> 1].x) + bo;
      }
    }
    tick();
    return;
  }
  // H (129 bands): M-tiles 0 .. 3 from xa -> xb rows 0 .. 127; wave 0 also M-tile 4 = row 128 (rows 129 .. 143 stay zero)
  NwsFewFrag<9> A9;
  nws_few_load<9>(A9, F, nws_few_frag_base(9), mt, li);                      // FIR design, M-tile mt
  if (mt == 0) nws_few_load<8>(An, F, nws_few_frag_base(8), 4, li);
  {
    const int o = 32 * mt + row;
    NwsFewAcc<NF> acc;
    nws_few_dot<8, NF>(A, L.xa, half, acc);
    const float bo = b_out0;
    if (half == 0) {
      if (NF == 2) *reinterpret_cast<f32x2*>(L.xb + o * NF) = f32x2{acc.p[0].x + bo, acc.p[0].y + bo};
      else *reinterpret_cast<f32x4*>(L.xb + o * NF) = f32x4{acc.p[0].x + bo, acc.p[0].y + bo, acc.p[NF / 2 - 1].x + bo, acc.p[NF / 2 - 1].y + bo};
    }
    if (mt == 0) {
      nws_few_dot<8, NF>(An, L.xa, half, acc);
      const float b128 = b_out1;
      if (row == 0 && half == 0) {
        if (NF == 2) *reinterpret_cast<f32x2*>(L.xb + 128 * NF) = f32x2{acc.p[0].x + b128, acc.p[0].y + b128};
        else *reinterpret_cast<f32x4*>(L.xb + 128 * NF) = f32x4{acc.p[0].x + b128, acc.p[0].y + b128, acc.p[NF / 2 - 1].x + b128, acc.p[NF / 2 - 1].y + b128};
      }
    }
  }
  __syncthreads();
  tick();
  {
    // fir = D[128 .. 255] H (upper half-taps), K = 144 padded
    const int o = 32 * mt + row;
    NwsFewAcc<NF> acc;
    nws_few_dot<9, NF>(A9, L.xb, half, acc);
    if (half == 0) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
        if (f < T) fir_w[((size_t)b * out_T + out_off + f) * NWS_FIR_HALF + o] = (f & 1) ? acc.p[f >> 1].y : acc.p[f >> 1].x;
    }
  }
  tick();
}
