// Which pipe of the command processor serves a HIP stream's hardware queue?  (pipeline.placed_streams)
//
// The reference has no counterpart (it enqueues everything on torch's current stream, models/neural_waveshaping.py:74-90); this
// file exists because the throughput mode of that forward runs on five streams (ForwardPipeline) and, on MI355X, WHERE their
// hardware queues sit decides up to 35 % of the step (profiles/r05/queue_placement.txt): a pipe dispatches one kernel at a
// time and stays on a kernel for as long as workgroups of its grid are waiting for a slot.
//
// The measurement: `hold` = a grid of one-wave workgroups that each take 64 000 B of LDS (two per CU: the grid is handed out
// in rounds, the wave slots stay free) and spin on the constant-rate wall clock; every workgroup folds its start time into
// (first start, last start).  Right behind it on ANOTHER stream, `touch` = one wave that stores the wall clock.  A touch
// stamp behind the hold grid's last start means the second queue was not served while the first one was dispatching: same
// pipe (or the same hardware queue - HIP shares queues between streams beyond GPU_MAX_HW_QUEUES).
#include "nws_common.h"

namespace {

__global__ __launch_bounds__(64) void queue_hold_kernel(unsigned long long* stamps, unsigned spin_ticks) {
  __shared__ float pad[16000];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) {
    atomicMin(&stamps[0], t0);
    atomicMax(&stamps[1], t0);
  }
  pad[threadIdx.x] = (float)(t0 & 0xffff);
  while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(16);
  if (pad[(threadIdx.x + 1) & 63] == -1.0f) stamps[3] = 0;      // never true: keeps the LDS allocation alive
}

__global__ __launch_bounds__(64) void queue_touch_kernel(unsigned long long* stamps) {
  if (threadIdx.x == 0) stamps[2] = wall_clock64();
}

// a resident load for the CU-pressure table (tools/cu_pressure.py): `groups` 256-thread workgroups that keep the vector pipe of
// their CU busy with dependent FMAs for `spin_ticks` of the wall clock - what a collective's ring kernels take from the
// oscillator kernel while a step's rows arrive
__global__ __launch_bounds__(256) void queue_busy_kernel(float* sink, unsigned spin_ticks) {
  const unsigned long long t0 = wall_clock64();
  float a = (float)threadIdx.x, b = 1.0001f;
  do {
#pragma unroll
    for (int i = 0; i < 64; ++i) a = fmaf(a, b, 0.5f);
  } while (wall_clock64() - t0 < spin_ticks);
  if (a == 12345.678f) sink[threadIdx.x] = a;                  // never true: keeps the loop alive
}

int wall_clock_khz(int* khz) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  e = hipDeviceGetAttribute(khz, hipDeviceAttributeWallClockRate, dev);
  if (e != hipSuccess) return (int)e;
  if (*khz <= 0) *khz = 100000;     // gfx9: the constant 100 MHz counter
  return NWS_OK;
}

}  // namespace

extern "C" {

int nws_queue_probe(void* stream_hold, void* stream_touch, int groups, int spin_us, unsigned long long* scratch, float* frac_out) {
  if (groups <= 0 || spin_us <= 0 || spin_us > 100000 || !scratch || !frac_out || stream_hold == stream_touch) return NWS_ERR_BAD_ARG;
  int khz = 0;
  int rc = wall_clock_khz(&khz);
  if (rc != NWS_OK) return rc;
  const hipStream_t sh = (hipStream_t)stream_hold, st = (hipStream_t)stream_touch;
  const unsigned long long init[4] = {~0ull, 0ull, 0ull, 0ull};
  hipError_t e = hipMemcpyAsync(scratch, init, sizeof(init), hipMemcpyHostToDevice, sh);
  if (e != hipSuccess) return (int)e;
  if ((e = hipStreamSynchronize(sh)) != hipSuccess) return (int)e;
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return (int)e;
  const unsigned ticks = (unsigned)((long long)spin_us * khz / 1000);
  queue_hold_kernel<<<groups, 64, 0, sh>>>(scratch, ticks);
  NWS_CHECK_LAUNCH();
  queue_touch_kernel<<<1, 64, 0, st>>>(scratch);
  NWS_CHECK_LAUNCH();
  if ((e = hipStreamSynchronize(st)) != hipSuccess) return (int)e;
  if ((e = hipStreamSynchronize(sh)) != hipSuccess) return (int)e;
  unsigned long long got[4];
  if ((e = hipMemcpy(got, scratch, sizeof(got), hipMemcpyDeviceToHost)) != hipSuccess) return (int)e;
  if (got[1] <= got[0]) {           // the whole grid started at once: nothing was held (fewer groups than slots)
    *frac_out = -1.0f;
    return NWS_OK;
  }
  *frac_out = (float)((double)((long long)(got[2] - got[0])) / (double)(got[1] - got[0]));
  return NWS_OK;
}

int nws_debug_queue_busy(int groups, int spin_us, float* sink, void* stream) {
  if (groups <= 0 || spin_us <= 0 || spin_us > 100000 || !sink) return NWS_ERR_BAD_ARG;
  int khz = 0;
  int rc = wall_clock_khz(&khz);
  if (rc != NWS_OK) return rc;
  queue_busy_kernel<<<groups, 256, 0, (hipStream_t)stream>>>(sink, (unsigned)((long long)spin_us * khz / 1000));
  NWS_CHECK_LAUNCH();
  return NWS_OK;
}

}  // extern "C"
