// Diagnostic kernels for the MI355X co-execution hazard that shaped ForwardPipeline (LABBOOK.md "5.2").
//
// Observation (ROCm 7.2 image, torch 2.10+rocm7.0 runtime, MI355X): a packed fp32 VALU instruction
//   v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32  with op_sel[1] = 1
// (the LOW result lane reads the HIGH half of src1: a swapped or high-broadcast second operand) sporadically returns a
// wrong value while ANOTHER WAVE on the same compute unit - a different kernel on another stream, or other waves of the
// very same workgroup (nws_coexec_pk_probe_mixed) - issues the K=16/K=32 half-precision MFMAs
// v_mfma_f32_32x32x16_f16 / v_mfma_f32_16x16x32_f16.  Not affected: the same packed instructions without that swizzle
// (plain, src0 or src2 swizzles, op_sel_hi-only broadcasts, neg modifiers), scalar VALU, LDS contents, barriers; not a
// trigger: v_mfma_f32_32x32x8f16, v_mfma_f32_32x32x2f32, dense VALU.  The reverb's FFT butterflies ("times -i" is a swapped
// second operand) were the victims of the frame-MLP / noise kernels of the next batch.
//
//   nws_coexec_pk_probe    evaluates eight packed forms with explicit instructions and checks each against scalar
//                          arithmetic on the same operands; report[k] counts wrong results of form k.
//   nws_coexec_mfma_load   a bare MFMA loop of the chosen flavour to run beside it on another stream.
// tools/coexec_probe.py prints the matrix; tests/test_gpu_coexec.py keeps the probe honest and checks the product path.
#include "nws_common.h"
#include "../../include/nws_probe.h"

namespace {
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sfma(float a, float b, float c) {
  float d;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float smul(float a, float b) {
  float d;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float sadd(float a, float b) {
  float d;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

__device__ __forceinline__ void pk_probe_body(int iters, unsigned* __restrict__ report) {
  const int tid = threadIdx.x;
  unsigned bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float2 a = make_float2(0.37f + 0.001f * (float)tid, -1.21f + 0.002f * (float)(blockIdx.x & 63));
  float2 b = make_float2(1.0009f, 0.9991f), c = make_float2(0.125f, -0.375f);
  for (int it = 0; it < iters; ++it) {
    float2 d;
    // 0..3: forms that were never seen to fail
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    bad[0] += ((d.x != sfma(a.x, b.x, c.x)) || (d.y != sfma(a.y, b.y, c.y))) ? 1u : 0u;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(c));  // src0 swapped
    bad[1] += ((d.x != sadd(a.y, c.x)) || (d.y != sadd(a.x, c.y))) ? 1u : 0u;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(c));               // src1 low broadcast
    bad[2] += ((d.x != sadd(a.x, c.x)) || (d.y != sadd(a.y, c.x))) ? 1u : 0u;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0] neg_lo:[0,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    bad[3] += ((d.x != sfma(a.x, b.x, -c.y)) || (d.y != sfma(a.y, b.y, c.x))) ? 1u : 0u;                           // src2 swapped
    // 4..7: op_sel[1] = 1
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(c));  // src1 swapped
    bad[4] += ((d.x != sadd(a.x, c.y)) || (d.y != sadd(a.y, c.x))) ? 1u : 0u;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(c));                   // src1 high broadcast
    bad[5] += ((d.x != sadd(a.x, c.y)) || (d.y != sadd(a.y, c.y))) ? 1u : 0u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    bad[6] += ((d.x != smul(a.x, b.y)) || (d.y != smul(a.y, b.x))) ? 1u : 0u;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    bad[7] += ((d.x != sfma(a.x, b.y, c.x)) || (d.y != sfma(a.y, b.x, c.y))) ? 1u : 0u;
    // next operands (kept in a sane range)
    a = make_float2(sfma(a.x, 0.9993f, smul(0.0007f, d.y)), sfma(a.y, 0.9989f, smul(-0.0003f, d.x)));
    b = make_float2(sfma(b.x, 0.99f, 0.0101f), sfma(b.y, 0.99f, 0.0099f));
  }
  for (int k = 0; k < 8; ++k)
    if (bad[k]) atomicAdd(&report[k], bad[k]);
}

__global__ __launch_bounds__(256) void pk_probe_kernel(int iters, unsigned* __restrict__ report) { pk_probe_body(iters, report); }

// Same kernel, different waves: waves 0-1 of every workgroup run the probe, waves 2-3 a v_mfma_f32_16x16x32_f16 loop - is the
// hazard a matter of two KERNELS sharing a compute unit, or of any two waves?
__global__ __launch_bounds__(256) void pk_probe_mixed_kernel(int iters, int mfma_iters, unsigned* __restrict__ report,
                                                             float* __restrict__ sink) {
  if (threadIdx.x < 128) {
    pk_probe_body(iters, report);
  } else {
    typedef float f32x4m __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) {
      a[q] = (_Float16)(0.01f * (float)(lane + q));
      b[q] = (_Float16)(0.02f * (float)(q + 1));
    }
    f32x4m c4 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int it = 0; it < mfma_iters; ++it) {
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
    }
    if (c4[0] + c4[1] + c4[2] + c4[3] == 12345.678f) sink[threadIdx.x] = c4[0];
  }
}

// Second probe: the same question for the other VOP3P families and for a scalar-register second operand.  Every swizzled form
// is checked against the PLAIN form of the same instruction on operands that were swizzled beforehand with integer moves.
//  0 v_pk_add_f16 src1 halves swapped      1 v_pk_fma_f16 src1 halves swapped     2 v_pk_mul_f16 src1 high half broadcast
//  3 v_fma_mix_f32 src1 = high fp16 half    4 v_fma_mixlo_f16 src1 = high fp16 half
//  5 v_pk_add_f32 src1 = SGPR pair, high broadcast    6 v_pk_mul_f32 src1 = SGPR pair, swapped    7 v_pk_add_f16 plain (control)
//  8 v_add_f64   9 v_mul_f64   10 v_fma_f64   (64-bit register pairs in src1; each checked against the same instruction with
//  src0 and src1 exchanged - the operations commute bit for bit)
__global__ __launch_bounds__(256) void pk_probe2_kernel(int iters, unsigned* __restrict__ report) {
  const int tid = threadIdx.x;
  unsigned bad[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double da = 1.0 + 1e-3 * (double)tid, db = 0.75 - 1e-4 * (double)(blockIdx.x & 63), dc = 0.125;
  // packed fp16 operands as raw 32-bit words: {lo, hi}
  unsigned a = 0x3c003800u + (unsigned)tid * 0x00010003u;       // ~{0.5.., 1.0..}
  unsigned b = 0x34003a00u + (unsigned)(blockIdx.x & 63) * 0x00030001u;
  unsigned c = 0x2e663266u;
  float2 fa = make_float2(0.37f + 0.001f * (float)tid, -1.21f + 0.002f * (float)(blockIdx.x & 63));
  float fc = 0.125f;
  for (int it = 0; it < iters; ++it) {
    const unsigned b_sw = (b >> 16) | (b << 16);      // halves swapped
    const unsigned b_hh = (b >> 16) | (b & 0xffff0000u);  // high half in both
    unsigned d, r;
    asm volatile("v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b_sw));
    bad[0] += d != r;
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b_sw), "v"(c));
    bad[1] += d != r;
    asm volatile("v_pk_mul_f16 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b_hh));
    bad[2] += d != r;
    float df, rf;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(df) : "v"(a), "v"(b), "v"(fc));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(rf) : "v"(a), "v"(b_sw), "v"(fc));
    bad[3] += __float_as_uint(df) != __float_as_uint(rf);
    d = 0u; r = 0u;
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "+v"(d) : "v"(a), "v"(b), "v"(fc));
    asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "+v"(r) : "v"(a), "v"(b_sw), "v"(fc));
    bad[4] += d != r;
    // scalar-register pair as second operand (wave-uniform values)
    const float s_lo = 0.75f + 0.001f * (float)(it & 31), s_hi = -0.3125f + 0.002f * (float)(it & 15);
    const unsigned long long sp = ((unsigned long long)__float_as_uint(s_hi) << 32) | __float_as_uint(s_lo);
    const unsigned long long spu = __builtin_amdgcn_readfirstlane((unsigned)sp) |
                                   ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sp >> 32)) << 32);
    float2 d2;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d2) : "v"(fa), "s"(spu));
    bad[5] += ((d2.x != sadd(fa.x, s_hi)) || (d2.y != sadd(fa.y, s_hi))) ? 1u : 0u;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d2) : "v"(fa), "s"(spu));
    bad[6] += ((d2.x != smul(fa.x, s_hi)) || (d2.y != smul(fa.y, s_lo))) ? 1u : 0u;
    // control: the plain packed f16 add against two scalar halves
    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    {
      unsigned lo, hi;
      asm volatile("v_add_f16 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
      asm volatile("v_add_f16 %0, %1, %2" : "=v"(hi) : "v"(a >> 16), "v"(b >> 16));
      bad[7] += d != ((lo & 0xffffu) | (hi << 16));
    }
    {
      double x, y;
      asm volatile("v_add_f64 %0, %1, %2" : "=v"(x) : "v"(da), "v"(db));
      asm volatile("v_add_f64 %0, %1, %2" : "=v"(y) : "v"(db), "v"(da));
      bad[8] += __double_as_longlong(x) != __double_as_longlong(y);
      asm volatile("v_mul_f64 %0, %1, %2" : "=v"(x) : "v"(da), "v"(db));
      asm volatile("v_mul_f64 %0, %1, %2" : "=v"(y) : "v"(db), "v"(da));
      bad[9] += __double_as_longlong(x) != __double_as_longlong(y);
      asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(x) : "v"(da), "v"(db), "v"(dc));
      asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(y) : "v"(db), "v"(da), "v"(dc));
      bad[10] += __double_as_longlong(x) != __double_as_longlong(y);
      da = da * 0.999 + 0.001 * y;
      db = db * 0.998 + 0.0021;
    }
    // next operands: small integer steps keep the fp16 fields finite and varied
    a = (a & 0x7bff7bffu) ^ ((d & 0x000f000fu) | 0x30003000u);
    b = (b & 0x7bff7bffu) ^ ((unsigned)it * 0x00050003u & 0x00ff00ffu);
    fa = make_float2(sfma(fa.x, 0.9993f, 0.0007f), sfma(fa.y, 0.9989f, -0.0003f));
  }
  for (int k = 0; k < 11; ++k)
    if (bad[k]) atomicAdd(&report[k], bad[k]);
}

// KIND 0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_16x16x32_f16   2: v_mfma_f32_32x32x8f16   3: v_mfma_f32_32x32x2f32
template <int KIND>
__global__ __launch_bounds__(256, 4) void mfma_load_kernel(int iters, float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  f32x16 acc = {};
  f16x8 a, b;
  for (int q = 0; q < 8; ++q) {
    a[q] = (_Float16)(0.01f * (float)(lane + q));
    b[q] = (_Float16)(0.02f * (float)(q + 1));
  }
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    } else if (KIND == 1) {
      f32x4 c4 = {acc[0], acc[1], acc[2], acc[3]};
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
      c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
      acc[0] = c4[0]; acc[1] = c4[1]; acc[2] = c4[2]; acc[3] = c4[3];
    } else if (KIND == 2) {
      const f16x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
      acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc, 0, 0, 0);
    }
  }
  float t = 0.0f;
  for (int r = 0; r < 16; ++r) t += acc[r];
  if (t == 12345.678f) sink[threadIdx.x] = t;  // never true: keeps the loop alive
}
}  // namespace

extern "C" {

int nws_coexec_pk_probe(int blocks, int iters, unsigned* report, void* stream) {
  if (blocks <= 0 || iters <= 0 || !report) return NWS_ERR_BAD_ARG;
  pk_probe_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, report);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_coexec_pk_probe_mixed(int blocks, int iters, int mfma_iters, unsigned* report, float* sink, void* stream) {
  if (blocks <= 0 || iters <= 0 || mfma_iters <= 0 || !report || !sink) return NWS_ERR_BAD_ARG;
  pk_probe_mixed_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, mfma_iters, report, sink);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_coexec_pk_probe2(int blocks, int iters, unsigned* report, void* stream) {
  if (blocks <= 0 || iters <= 0 || !report) return NWS_ERR_BAD_ARG;
  pk_probe2_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(iters, report);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

int nws_coexec_mfma_load(int kind, int blocks, int iters, float* sink, void* stream) {
  if (kind < 0 || kind > 3 || blocks <= 0 || iters <= 0 || !sink) return NWS_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (kind == 0) mfma_load_kernel<0><<<blocks, 256, 0, st>>>(iters, sink);
  else if (kind == 1) mfma_load_kernel<1><<<blocks, 256, 0, st>>>(iters, sink);
  else if (kind == 2) mfma_load_kernel<2><<<blocks, 256, 0, st>>>(iters, sink);
  else mfma_load_kernel<3><<<blocks, 256, 0, st>>>(iters, sink);
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
