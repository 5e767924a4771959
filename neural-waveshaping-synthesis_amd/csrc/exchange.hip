// Host side of the waveform exchange of the multi-GPU path (the reference has no collective code, its only trace of
// more than one GPU is gin/train/train_newt.gin:13).  parallel.PeerCopyAllGather pushes every step's (b, N) shard into each
// peer's gather buffer with device-to-device copies, one per peer, each on that peer's own copy stream (copy engines over the
// point-to-point xGMI links).  Issued from Python, the seven copies + seven event records of an 8-rank job cost the helper thread
// ~0.13 ms per step of a 0.40 ms step (measured with seven local destinations, profiles/r06/fake_peers_ab.txt); these two calls
// do the same in one trip through the binding, without the interpreter lock.
#include "nws_common.h"

extern "C" {

int nws_peer_push(int n, void* const* dst, const void* src, size_t bytes, void* const* streams, void* const* events) {
  if (n < 0 || (n > 0 && (!dst || !streams)) || !src || bytes == 0) return NWS_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    if (!dst[i]) return NWS_ERR_BAD_ARG;
    hipError_t e = hipMemcpyAsync(dst[i], src, bytes, hipMemcpyDefault, (hipStream_t)streams[i]);
    if (e != hipSuccess) return (int)e;
    if (events && events[i]) {
      e = hipEventRecord((hipEvent_t)events[i], (hipStream_t)streams[i]);
      if (e != hipSuccess) return (int)e;
    }
  }
  return NWS_OK;
}

int nws_events_wait(int n, void* const* events) {
  if (n < 0 || (n > 0 && !events)) return NWS_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    if (!events[i]) continue;
    const hipError_t e = hipEventSynchronize((hipEvent_t)events[i]);
    if (e != hipSuccess) return (int)e;
  }
  return NWS_OK;
}

int nws_streams_wait(int n, void* const* streams) {
  if (n < 0 || (n > 0 && !streams)) return NWS_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) {
    const hipError_t e = hipStreamSynchronize((hipStream_t)streams[i]);
    if (e != hipSuccess) return (int)e;
  }
  return NWS_OK;
}

}  // extern "C"
