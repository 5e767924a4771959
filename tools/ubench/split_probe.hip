// (hi, lo) fp16 splits of fp32 values: round-then-residual (v_cvt_pk + v_fma_mix, as the oscillator kernel) against mask-and-subtract
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// BAD: hipcc 7.2 reads element 0 for __builtin_bit_cast(unsigned, v.y) of an ext_vector (hi.y / lo.y come out relative to v.x)
__device__ __forceinline__ void split_mask2_bad(f32x2 v, f16x2& hi, f16x2& lo) {
  const float hx = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.x) & 0xFFFFE000u);
  const float hy = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.y) & 0xFFFFE000u);
  hi = __builtin_convertvector(f32x2{hx, hy}, f16x2);
  lo = __builtin_convertvector(f32x2{v.x - hx, v.y - hy}, f16x2);
}
__device__ __forceinline__ float mask_hi11(float x) { return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xFFFFE000u); }
__device__ __forceinline__ void split_mask2(f32x2 v, f16x2& hi, f16x2& lo) {
  const float vx = v.x, vy = v.y;
  const float hx = mask_hi11(vx), hy = mask_hi11(vy);
  hi = __builtin_convertvector(f32x2{hx, hy}, f16x2);
  lo = __builtin_convertvector(f32x2{vx - hx, vy - hy}, f16x2);
}
__global__ void k(const float* x, float* o, int bad) {
  const int i = threadIdx.x;
  f32x2 v = {x[2 * i], x[2 * i + 1]};
  f16x2 hi, lo;
  if (bad) split_mask2_bad(v, hi, lo); else split_mask2(v, hi, lo);
  o[4 * i] = (float)hi.x; o[4 * i + 1] = (float)lo.x; o[4 * i + 2] = (float)hi.y; o[4 * i + 3] = (float)lo.y;
}
int main() {
  float hx[128], ho[256];
  for (int i = 0; i < 128; ++i) hx[i] = sinf(0.37f * i + 0.1f) * (i % 7 == 0 ? 1e-3f : 1.0f);
  float *x, *o; hipMalloc(&x, 512); hipMalloc(&o, 1024);
  hipMemcpy(x, hx, 512, hipMemcpyHostToDevice);
  for (int bad = 0; bad < 2; ++bad) {
    k<<<1, 64>>>(x, o, bad);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    double worst = 0, worst_hi = 0;
    for (int i = 0; i < 128; ++i) {
      const double e = fabs((double)ho[2 * i] + (double)ho[2 * i + 1] - (double)hx[i]);
      if (e > worst) worst = e;
      const double eh = fabs((double)ho[2 * i] - (double)hx[i]) / fmax(fabs(hx[i]), 1e-30);
      if (eh > worst_hi) worst_hi = eh;
    }
    printf("%s form: worst |hi + lo - v| %.3e, worst |hi - v| / |v| %.3e (a correct hi is within 2^-10)\n", bad ? "bit_cast(v.y)" : "scalar-copy", worst, worst_hi);
  }
  return 0;
}
