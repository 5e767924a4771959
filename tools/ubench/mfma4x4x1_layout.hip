// Layout probe for v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4x1 blocks per wave): which lane supplies which operand element
// and where the results land.  hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma4x4x1_layout.hip -o /tmp/m && /tmp/m
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}
int main() {
  float ha[64], hb[64], hd[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1.0f + l; hb[l] = 100.0f * (1 + l); }
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, b, d);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost);
  // hypothesis: block blk = l / 4; D[i][j] of block blk = A(lane 4 blk + i) * B(lane 4 blk + j), held by lane 4 blk + j, register i
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int blk = l / 4, j = l % 4;
      const float want = ha[4 * blk + i] * hb[4 * blk + j];
      if (hd[l * 4 + i] != want) ++bad;
    }
  printf("layout hypothesis (lane = 4 blk + j holds column j; register i = row i; A from lane 4 blk + i): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  if (bad) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hd[4*l], hd[4*l+1], hd[4*l+2], hd[4*l+3]);
  return bad != 0;
}
