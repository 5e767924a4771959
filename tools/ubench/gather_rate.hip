// Micro-benchmark: what a 64-lane 8-byte global gather from an L2-resident 2 MB table costs as a function of how many
// distinct 128-byte lines (and how many distinct addresses) the wave touches - the access pattern of the FastNEWT tail
// (exciter_newt.hip: 32 gathers per lane and 32-sample wave, lanes = 32 consecutive samples x 2 shapers).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/gather_rate.hip -o /tmp/gather_rate.out && /tmp/gather_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// PAT: lanes per distinct line L in {64, 32, 16, 8, 4, 2, 1} (64 = one line for the wave ... 1 = every lane its own line);
// SAME = 1: lanes that share a line read the SAME 8 bytes (a smooth signal: neighbouring samples hit one table cell)
template <int L, int SAME, int LDS>
__global__ __launch_bounds__(512) void gather_kernel(const char* __restrict__ table, float* out, int iters) {
  extern __shared__ char lds[];
  const unsigned lane = threadIdx.x & 63;
  const unsigned grp = lane / L;                    // which line of the instruction
  const unsigned within = SAME ? 0u : (lane % L) % 16u * 8u;
  unsigned u = blockIdx.x * 977u + (threadIdx.x >> 6) * 131u;
  f32x2 acc[8] = {};
  if (LDS) {
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      // wave-uniform pseudo-random base line, then the pattern
      const unsigned base = ((u + i * 2654435761u) * 2246822519u) >> 7;
      unsigned off;
      if (LDS) off = (((base + grp * 37u) * 128u) & 0xff80u) + within;
      else off = (((base + grp * 37u) * 128u) & 0x1fff80u) + within;
      if (LDS) acc[i] += *reinterpret_cast<const f32x2*>(lds + off);
      else acc[i] += *reinterpret_cast<const f32x2*>(table + off);
    }
    u += 7919u;
  }
  float a = 0.f;
  for (int i = 0; i < 8; ++i) a += acc[i].x + acc[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = a;
}

template <int L, int SAME, int LDS>
void run(const char* table, float* out) {
  const int iters = 256, blocks = 256 * 3 * 4;
  const size_t lds = LDS ? 65536 : 40000;   // 40 KB: three 8-wave workgroups per CU, like the oscillator kernel (64 KB: two)
  hipFuncSetAttribute(reinterpret_cast<const void*>(gather_kernel<L, SAME, LDS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  gather_kernel<L, SAME, LDS><<<blocks, 512, lds>>>(table, out, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  gather_kernel<L, SAME, LDS><<<blocks, 512, lds>>>(table, out, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_cu = (double)blocks * 8 * iters * 8 / 256.0;
  printf("%-6s lines/instr %2d  %s : %.3f ms  -> %.2f ns per wave-instruction per CU (%.1f clk at 2.1 GHz)\n", LDS ? "LDS" : "global",
         64 / L, SAME ? "same cell per line " : "16 cells per line  ", ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
}

int main() {
  char* table; float* out;
  hipMalloc(&table, 4u << 20); hipMemset(table, 0, 4u << 20);
  hipMalloc(&out, sizeof(float) * 512 * 256 * 12);
  run<64, 1, 0>(table, out); run<32, 1, 0>(table, out); run<16, 1, 0>(table, out); run<8, 1, 0>(table, out);
  run<4, 1, 0>(table, out); run<2, 1, 0>(table, out); run<1, 1, 0>(table, out);
  run<64, 0, 0>(table, out); run<32, 0, 0>(table, out); run<16, 0, 0>(table, out); run<8, 0, 0>(table, out); run<4, 0, 0>(table, out);
  run<64, 1, 1>(table, out); run<16, 0, 1>(table, out); run<4, 0, 1>(table, out); run<1, 1, 1>(table, out);
  return 0;
}
