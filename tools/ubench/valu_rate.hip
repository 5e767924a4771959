// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU forms the fused oscillator/NEWT kernel
// and the GRU are made of, at 1/2/4/8 waves per SIMD.  Decides what "fewer VALU instructions" has to mean on gfx950:
// is v_pk_fma_f32 one issue slot or two, what does a transcendental cost, do they overlap with plain VALU.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate.out && tools/ubench/valu_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters, float seed, const float* __restrict__ table) {
  extern __shared__ char lds[];
  float a[8];
  f32x2 p[8];
  double d[8];
  unsigned u[8];
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  typedef float f16v __attribute__((ext_vector_type(16)));
  h8 ha, hb;
  f16v m0 = {}, m1 = {}, m2 = {}, m3 = {};
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.001f * (threadIdx.x + i)); hb[i] = (_Float16)(0.002f * i); }
  const float s0 = seed + threadIdx.x * 1e-3f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = s0 + i;
    p[i] = f32x2{s0 + i, s0 - i};
    d[i] = (double)s0 + i;
    u[i] = threadIdx.x + i;
  }
  const float c1 = 0.999f, c2 = 1e-3f;
  const f32x2 q1 = {0.999f, 0.998f}, q2 = {1e-3f, 2e-3f};
  const double e1 = 0.999, e2 = 1e-3;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
        REP8(X)
#undef X
      } else if (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q1), "v"(q2));
        REP8(X)
#undef X
      } else if (OP == 2) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
        REP8(X)
#undef X
      } else if (OP == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q2));
        REP8(X)
#undef X
      } else if (OP == 4) {
#define X(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 5) {
#define X(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 6) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 7) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 8) {
#define X(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 9) {
#define X(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c2), "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 10) {
#define X(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 11) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 12) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 13) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 14) {
#define X(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e2));
        REP8(X)
#undef X
      } else if (OP == 15) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(e1), "v"(e2));
        REP8(X)
#undef X
      } else if (OP == 16) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 17) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1));
        REP8(X)
#undef X
      } else if (OP == 18) {  // 6 fma : 2 sin interleaved -> does the transcendental overlap with plain VALU?
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c1), "v"(c2));
        asm volatile("v_sin_f32 %0, %0" : "+v"(a[1]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(c1), "v"(c2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(c1), "v"(c2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(c1), "v"(c2));
        asm volatile("v_sin_f32 %0, %0" : "+v"(a[5]));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[6]) : "v"(c1), "v"(c2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[7]) : "v"(c1), "v"(c2));
      } else if (OP == 19) {  // 6 pk_fma : 2 sin
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(q1), "v"(q2));
        asm volatile("v_sin_f32 %0, %0" : "+v"(a[1]));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[2]) : "v"(q1), "v"(q2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[3]) : "v"(q1), "v"(q2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[4]) : "v"(q1), "v"(q2));
        asm volatile("v_sin_f32 %0, %0" : "+v"(a[5]));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[6]) : "v"(q1), "v"(q2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[7]) : "v"(q1), "v"(q2));
      } else if (OP == 20) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
        X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
      } else if (OP == 21) {  // broadcast ds_read_b128 (all lanes of a half read one address), like the GRU's h reads
        f32x2 lo, hi;
        typedef float f4 __attribute__((ext_vector_type(4)));
#define X(i) { f4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((threadIdx.x >> 5) * 256), "i"(i * 16)); \
               asm volatile("s_waitcnt lgkmcnt(0)\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(v.x), "v"(v.y)); }
        REP8(X)
#undef X
        (void)lo; (void)hi;
      } else if (OP == 22) {  // 8 ds_read_b128 in flight then 16 pk_fma (GRU step shape)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v[8];
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[i]) : "v"((threadIdx.x >> 5) * 256), "i"(i * 16));
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(f32x2{v[i].x, v[i].y}), "v"(q1)); \
             asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(f32x2{v[i].z, v[i].w}), "v"(q1));
        REP8(X)
#undef X
      } else if (OP == 23) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[i]) : "v"(u[i]));
        REP8(X)
#undef X
      } else if (OP == 30) {  // 4 independent v_mfma_f32_32x32x16_f16 x2
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m0) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m1) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m2) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m3) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m0) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m1) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m2) : "v"(ha), "v"(hb));
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m3) : "v"(ha), "v"(hb));
      } else if (OP == 31) {  // dependent chain: 8 MFMAs on ONE accumulator
#define X(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m0) : "v"(ha), "v"(hb));
        REP8(X)
#undef X
      } else if (OP == 32) {  // two chains of 3 dependent MFMAs (the mixer's K-step shape), 2 idle slots -> 8 "instructions"
#define X(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m0) : "v"(ha), "v"(hb));
        X(0) X(1) X(2)
#undef X
#define X(i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(m1) : "v"(ha), "v"(hb));
        X(0) X(1) X(2)
#undef X
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c1), "v"(c2));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(c1), "v"(c2));
      } else if (OP == 33) {  // per MFMA (independent accumulators): 4 v_fma_f32 fillers; count = 8 x (1 mfma + 4 fma) -> report per group of 5
#define G(M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(M) : "v"(ha), "v"(hb)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(c1), "v"(c2));
        G(m0) G(m1) G(m2) G(m3) G(m0) G(m1) G(m2) G(m3)
#undef G
      } else if (OP == 34) {  // per MFMA: 4 v_pk_fma_f32 fillers
#define G(M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(M) : "v"(ha), "v"(hb)); \
             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(q1), "v"(q2)); \
             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[1]) : "v"(q1), "v"(q2)); \
             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[2]) : "v"(q1), "v"(q2)); \
             asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[3]) : "v"(q1), "v"(q2));
        G(m0) G(m1) G(m2) G(m3) G(m0) G(m1) G(m2) G(m3)
#undef G
      } else if (OP == 35) {  // per MFMA: 8 v_fma_f32 fillers
#define G(M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(M) : "v"(ha), "v"(hb)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[2]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[5]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[6]) : "v"(c1), "v"(c2)); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[7]) : "v"(c1), "v"(c2));
        G(m0) G(m1) G(m2) G(m3) G(m0) G(m1) G(m2) G(m3)
#undef G
      } else if (OP == 36) {  // per MFMA: 2 v_sin_f32 + 2 v_fma fillers
#define G(M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(M) : "v"(ha), "v"(hb)); \
             asm volatile("v_sin_f32 %0, %0" : "+v"(a[0])); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[1]) : "v"(c1), "v"(c2)); \
             asm volatile("v_sin_f32 %0, %0" : "+v"(a[2])); \
             asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(c1), "v"(c2));
        G(m0) G(m1) G(m2) G(m3) G(m0) G(m1) G(m2) G(m3)
#undef G
      } else if (OP >= 50 && OP <= 57) {  // per MFMA: 4 fillers of one class (does the class overlap with the matrix pipe?)
#define F0(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define F1(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define F2(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
#define F3(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(a[i]));
#define F4(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i]), "v"(c1));
#define F5(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e2));
#define F6(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
#define F7(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
#define FF(i) if (OP == 50) { F0(i) } else if (OP == 51) { F1(i) } else if (OP == 52) { F2(i) } else if (OP == 53) { F3(i) } \
              else if (OP == 54) { F4(i) } else if (OP == 55) { F5(i) } else if (OP == 56) { F6(i) } else { F7(i) }
#define G(M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(M) : "v"(ha), "v"(hb)); FF(0) FF(1) FF(2) FF(3)
        G(m0) G(m1) G(m2) G(m3) G(m0) G(m1) G(m2) G(m3)
#undef G
      } else if (OP == 70 || OP == 71 || OP == 72) {
        // the oscillator K-step's instruction multiset: 6 MFMAs (two chains of 3) + 60 vector instructions (per sine pair:
        // 4 fma-class, 2 v_mul, 2 fract, 2 sin, 1 cvt_pk, 2 fma_mix + 2 more fma-class = 15; x4), either interleaved 1 : 10
        // (OP 70) or clustered like the compiler's schedule (OP 71: 60 vector then 6 MFMA); OP 72 = the 60 vector alone
#define V10(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c1)); \
               asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i + 1]) : "v"(c1), "v"(c2)); \
               asm volatile("v_fract_f32 %0, %0" : "+v"(a[i + 2])); \
               asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i + 3]) : "v"(c1), "v"(c2)); \
               asm volatile("v_sin_f32 %0, %0" : "+v"(a[i])); \
               asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i + 1]) : "v"(c2)); \
               asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[i]) : "v"(a[i + 2]), "v"(c1)); \
               asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u[i + 1]) : "v"(u[i + 2]), "v"(a[i + 3])); \
               asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i + 2]) : "v"(c1)); \
               asm volatile("v_sin_f32 %0, %0" : "+v"(a[i + 3]));
#define MF(M) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(M) : "v"(ha), "v"(hb));
        if (OP == 70) {
          MF(m0) V10(0) MF(m0) V10(4) MF(m0) V10(0) MF(m1) V10(4) MF(m1) V10(0) MF(m1) V10(4)
        } else if (OP == 71) {
          V10(0) V10(4) V10(0) V10(4) V10(0) V10(4) MF(m0) MF(m0) MF(m0) MF(m1) MF(m1) MF(m1)
        } else {
          V10(0) V10(4) V10(0) V10(4) V10(0) V10(4)
        }
#undef V10
#undef MF
      } else if (OP == 58) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 59) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        REP8(X)
#undef X
      } else if (OP == 60) {
#define X(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
        REP8(X)
#undef X
      } else if (OP == 61) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 62) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c2));
        REP8(X)
#undef X
      } else if (OP == 63) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) & 7]));
        REP8(X)
#undef X
      } else if (OP == 64) {
#define X(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        REP8(X)
#undef X
      } else if (OP == 65) {
#define X(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
        REP8(X)
#undef X
      } else if (OP == 37) {  // 8 x v_mfma_f32_32x32x8_f16 (the pre-gfx950 half-K form), independent accumulators
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const h4 a4 = {ha[0], ha[1], ha[2], ha[3]}, b4 = {hb[0], hb[1], hb[2], hb[3]};
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m0) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m1) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m2) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m3) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m0) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m1) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m2) : "v"(a4), "v"(b4));
        asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(m3) : "v"(a4), "v"(b4));
      } else if (OP == 38) {  // 8 divergent 8-byte gathers from a 2 MB table (L2-resident): the LUT tail's access pattern
        const char* tb = reinterpret_cast<const char*>(table);
#define X(i) { const unsigned off = ((u[i] * 2654435761u) >> 11) & 0x1ffff8u; \
               p[i] += *reinterpret_cast<const f32x2*>(tb + off); u[i] += (unsigned)it * 64u + 1u; }
        REP8(X)
#undef X
      } else if (OP == 39) {  // same count of gathers, 16 lanes per 128-B line
        const char* tb = reinterpret_cast<const char*>(table);
#define X(i) { const unsigned off = ((((u[i] >> 4) * 2654435761u) >> 11) & 0x1fff80u) + (threadIdx.x & 15) * 8; \
               p[i] += *reinterpret_cast<const f32x2*>(tb + off); u[i] += (unsigned)it * 64u + 16u; }
        REP8(X)
#undef X
      } else if (OP == 41) {  // 4-byte random gathers
        const char* tb = reinterpret_cast<const char*>(table);
#define X(i) { const unsigned off = ((u[i] * 2654435761u) >> 11) & 0x1ffffcu; \
               a[i] += *reinterpret_cast<const float*>(tb + off); u[i] += (unsigned)it * 64u + 1u; }
        REP8(X)
#undef X
      } else if (OP == 42) {  // random 8-byte gathers inside a 16 KB window (one shaper's hot table range: L1-resident)
        const char* tb = reinterpret_cast<const char*>(table);
#define X(i) { const unsigned off = ((u[i] * 2654435761u) >> 11) & 0x3ff8u; \
               p[i] += *reinterpret_cast<const f32x2*>(tb + off); u[i] += (unsigned)it * 64u + 1u; }
        REP8(X)
#undef X
      } else if (OP == 40) {  // 8 random ds_read_b64 gathers from 64 KB of LDS
#define X(i) { f32x2 v; unsigned off = ((u[i] * 2654435761u) >> 16) & 0xfff8u; \
               asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(off)); u[i] += (unsigned)it; p[i] = v; }
        REP8(X)
#undef X
        asm volatile("s_waitcnt lgkmcnt(0)");
      } else if (OP == 24) {  // v_pk_fma_f32 with one SGPR-pair operand (exact-bank shaper form)
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "s"(q1), "v"(q2));
        REP8(X)
#undef X
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc += a[i] + p[i].x + p[i].y + (float)d[i] + (float)u[i];
  for (int i = 0; i < 16; ++i) acc += m0[i] + m1[i] + m2[i] + m3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_sin_f32", "v_fract_f32",
                               "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_floor_f32", "v_med3_f32", "v_cvt_i32_f32",
                               "v_lshl_add_u32", "v_exp_f32", "v_rcp_f32", "v_add_f64", "v_fma_f64", "v_mov_b32_dpp",
                               "v_mul_f32", "6 fma + 2 sin", "6 pk_fma + 2 sin", "v_permlane32_swap",
                               "ds_read_b128 bcast + wait + fma", "8 ds_read_b128 + 16 pk_fma", "v_cvt_f32_f16",
                               "v_pk_fma_f32 (sgpr src)", "", "", "", "", "",
                               "mfma 32x32x16 f16 (4 indep acc)", "mfma 32x32x16 dependent chain", "2 chains of 3 dep mfma + 2 fma",
                               "8 x (mfma + 4 v_fma) [per instr]", "8 x (mfma + 4 v_pk_fma) [per instr]",
                               "8 x (mfma + 8 v_fma) [per instr]", "8 x (mfma + 2 sin + 2 fma) [per instr]",
                               "mfma 32x32x8 f16 (4 indep acc)", "8 random 8-B global gathers (2 MB)",
                               "8 gathers, 16 lanes per 128-B line", "8 random ds_read_b64 (64 KB)"};

static float* g_table = nullptr;
static const char* name_of(int op);
template <int OP>
void run(float* out, long long* cyc, hipStream_t st, const float* table = g_table) {
  const int iters = 512;
  const int group = (OP == 70 || OP == 71) ? 0 : OP == 72 ? -1 : (OP == 33 || OP == 34 || OP == 36 || (OP >= 50 && OP <= 57)) ? 5 : (OP == 35 ? 9 : 1);
  const int instr_per_wave = group ? (group > 0 ? iters * 4 * 8 * group : iters * 4 * 60) : iters * 4 * 66;
  printf("%-38s", name_of(OP));
  for (int k : {1, 2, 4, 8}) {
    // k workgroups of 4 waves per CU: k waves per SIMD (LDS reservation pins the residency)
    const size_t lds = k == 8 ? 16 * 1024 : (k == 1 && OP != 40 ? 159 * 1024 : (160 * 1024) / k - 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * k;
    rate_kernel<OP><<<blocks, 256, lds, st>>>(out, cyc, 16, 1.0f, table);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    rate_kernel<OP><<<blocks, 256, lds, st>>>(out, cyc, iters, 1.0f, table);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h) mean += (double)v;
    mean /= h.size();
    // issue cost per instruction per SIMD = wave-elapsed cycles / (instructions x co-resident waves)
    printf("  k=%d: %6.2f cyc/inst/SIMD (wave %7.0f cyc, %.3f ms)", k, mean / ((double)instr_per_wave * k), mean, ms);
    fflush(stdout);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  printf("\n");
}

static const char* name_of(int op) {
  switch (op) {
    case 41: return "8 random 4-B global gathers (2 MB)";
    case 42: return "8 random 8-B gathers in 16 KB";
    case 50: return "8 x (mfma + 4 v_floor) [per instr]";
    case 51: return "8 x (mfma + 4 v_lshl_add) [per instr]";
    case 52: return "8 x (mfma + 4 v_sin) [per instr]";
    case 53: return "8 x (mfma + 4 fma_mixlo) [per instr]";
    case 54: return "8 x (mfma + 4 cvt_pk_f16) [per instr]";
    case 55: return "8 x (mfma + 4 v_add_f64) [per instr]";
    case 56: return "8 x (mfma + 4 mov_dpp) [per instr]";
    case 57: return "8 x (mfma + 4 v_and) [per instr]";
    case 70: return "K-step mix interleaved 1 mfma : 10 v [per instr]";
    case 71: return "K-step mix clustered 60 v + 6 mfma [per instr]";
    case 72: return "K-step vector part alone (60 v) [per instr]";
    case 58: return "v_and_b32";
    case 59: return "v_add_f32";
    case 60: return "v_cvt_u32_f32";
    case 61: return "v_min_u32";
    case 62: return "v_max_f32";
    case 63: return "v_mov_b32";
    case 64: return "v_pk_fma_f16";
    case 65: return "v_dot2_f32_f16";
    default: return kNames[op];
  }
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, sizeof(float) * 256 * 8 * 256 < (4u << 20) ? (4u << 20) : sizeof(float) * 256 * 8 * 256);
  hipMalloc(&cyc, sizeof(long long) * 256 * 8 * 4);
  hipMalloc(&g_table, 4u << 20);
  hipMemset(g_table, 0, 4u << 20);
  hipStream_t st;
  hipStreamCreate(&st);
  printf("cycles are s_memtime/readcyclecounter ticks; cost = per-wave elapsed / (instructions * waves per SIMD)\n");
  if (getenv("KSTEP_ONLY")) {
    run<70>(out, cyc, st); run<71>(out, cyc, st); run<72>(out, cyc, st); run<30>(out, cyc, st);
    return 0;
  }
  run<0>(out, cyc, st); run<1>(out, cyc, st); run<2>(out, cyc, st); run<3>(out, cyc, st); run<17>(out, cyc, st);
  run<4>(out, cyc, st); run<5>(out, cyc, st); run<12>(out, cyc, st); run<13>(out, cyc, st);
  run<6>(out, cyc, st); run<7>(out, cyc, st); run<23>(out, cyc, st); run<8>(out, cyc, st); run<9>(out, cyc, st);
  run<10>(out, cyc, st); run<11>(out, cyc, st); run<14>(out, cyc, st); run<15>(out, cyc, st); run<16>(out, cyc, st);
  run<20>(out, cyc, st); run<18>(out, cyc, st); run<19>(out, cyc, st); run<24>(out, cyc, st);
  run<21>(out, cyc, st); run<22>(out, cyc, st);
  run<30>(out, cyc, st); run<31>(out, cyc, st); run<32>(out, cyc, st); run<37>(out, cyc, st);
  run<33>(out, cyc, st); run<34>(out, cyc, st); run<35>(out, cyc, st); run<36>(out, cyc, st);
  run<50>(out, cyc, st); run<51>(out, cyc, st); run<52>(out, cyc, st); run<53>(out, cyc, st); run<54>(out, cyc, st);
  run<55>(out, cyc, st); run<56>(out, cyc, st); run<57>(out, cyc, st);
  run<58>(out, cyc, st); run<59>(out, cyc, st); run<60>(out, cyc, st); run<61>(out, cyc, st); run<62>(out, cyc, st);
  run<63>(out, cyc, st); run<64>(out, cyc, st); run<65>(out, cyc, st);
  fflush(stdout);
  run<38>(out, cyc, st); run<39>(out, cyc, st); run<41>(out, cyc, st); run<42>(out, cyc, st); run<40>(out, cyc, st);
  return 0;
}
