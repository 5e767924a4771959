// Whole NeuralWaveshaping.forward (models/neural_waveshaping.py:74-90) as one enqueue on one stream:
//   phase carries -> GRU -> frame MLPs -> fused exciter+NEWT -> FIR noise (+ branch sum) -> reverb.
// Scratch comes from the caller's workspace; nothing is allocated or synchronised here, so the whole
// call can be captured into a hipGraph (streaming mode, scripts/time_buffer_sizes.py counterpart).
#include "nws_common.h"

namespace {

struct Carve {
  char* p;
  size_t left;
  bool ok = true;
  void* take(size_t bytes) {
    bytes = (bytes + 255) & ~size_t(255);
    if (bytes > left) {
      ok = false;
      return nullptr;
    }
    void* r = p;
    p += bytes;
    left -= bytes;
    return r;
  }
};

size_t aligned(size_t b) { return (b + 255) & ~size_t(255); }

// ---- optional live profiling: hipEvents recorded on the launch stream around each stage ----
constexpr int kStages = 6;  // carry, gru, mlps, exciter_newt, fir_noise, reverb
struct Profile {
  int slots = 0;      // forward calls that can be recorded
  int used = 0;
  unsigned mask = 0;  // bit s: bracket stage s
  hipEvent_t* ev = nullptr;  // [slots][kStages][2]
} g_prof;

inline hipEvent_t* prof_events(int stage) {
  if (g_prof.ev == nullptr || g_prof.used >= g_prof.slots || !(g_prof.mask & (1u << stage))) return nullptr;
  return g_prof.ev + ((size_t)g_prof.used * kStages + stage) * 2;
}

}  // namespace

extern "C" {

size_t nws_forward_workspace_bytes(const NwsReverbPlan* plan, int B, int T) {
  if (!plan || B <= 0 || T <= 0) return 0;
  const size_t N = (size_t)T * NWS_HOP;
  size_t total = 0;
  total += aligned((size_t)B * (N / 32) * sizeof(double));          // carries
  total += aligned((size_t)B * T * NWS_HIDDEN * sizeof(float));     // gru_out
  total += aligned((size_t)B * T * NWS_FILM_CH * sizeof(float));    // film
  total += aligned((size_t)B * T * NWS_FIR_LEN * sizeof(float));    // fir
  total += aligned((size_t)B * N * sizeof(float));                  // newt_out
  total += aligned((size_t)B * N * sizeof(float));                  // pre-reverb
  total += aligned(nws_reverb_workspace_bytes(plan, B));
  return total;
}

int nws_forward(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, const float* control, int B, int C,
                int T, float sample_rate, const float* phase_u, const float* rand_phase, const float* noise, float* out,
                void* workspace,
                size_t workspace_bytes, void* stream) {
  if (!w || !aux || !aux->plan || !aux->fir_design || !aux->reverb_tables || !aux->reverb_spectrum) return NWS_ERR_BAD_ARG;
  if (!f0 || !control || !phase_u || !rand_phase || !noise || !out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2 || C < 2) return NWS_ERR_BAD_ARG;
  const size_t N = (size_t)T * NWS_HOP;
  if ((long long)N > aux->plan->L) return NWS_ERR_BAD_ARG;
  Carve cv{static_cast<char*>(workspace), workspace_bytes};
  double* carry = static_cast<double*>(cv.take((size_t)B * (N / 32) * sizeof(double)));
  float* gru_out = static_cast<float*>(cv.take((size_t)B * T * NWS_HIDDEN * sizeof(float)));
  float* film = static_cast<float*>(cv.take((size_t)B * T * NWS_FILM_CH * sizeof(float)));
  float* fir = static_cast<float*>(cv.take((size_t)B * T * NWS_FIR_LEN * sizeof(float)));
  float* newt_out = static_cast<float*>(cv.take((size_t)B * N * sizeof(float)));
  float* pre = static_cast<float*>(cv.take((size_t)B * N * sizeof(float)));
  const size_t rv_bytes = nws_reverb_workspace_bytes(aux->plan, B);
  void* rv_ws = cv.take(rv_bytes);
  if (!cv.ok) return NWS_ERR_WORKSPACE;

  hipStream_t st = (hipStream_t)stream;
  int rc = NWS_OK;
#define NWS_STAGE(idx, call)                                   \
  do {                                                         \
    hipEvent_t* e__ = prof_events(idx);                        \
    if (e__) (void)hipEventRecord(e__[0], st);                 \
    rc = (call);                                               \
    if (e__) (void)hipEventRecord(e__[1], st);                 \
    if (rc != NWS_OK) return rc;                               \
  } while (0)
  NWS_STAGE(0, nws_phase_carry(f0, nullptr, B, T, carry, stream));
  NWS_STAGE(1, nws_control_gru(w, control, B, C, T, gru_out, stream));
  NWS_STAGE(2, nws_frame_mlps(w, gru_out, aux->fir_design, B, T, nullptr, film, nullptr, fir, stream));
  NWS_STAGE(3, nws_exciter_newt(w, f0, nullptr, carry, phase_u, rand_phase, film, B, T, sample_rate, nullptr, newt_out, stream));
  NWS_STAGE(4, nws_fir_noise(fir, noise, newt_out, B, T, pre, stream));
  NWS_STAGE(5, nws_reverb(aux->plan, aux->reverb_tables, aux->reverb_spectrum, pre, B, (int)N, out, rv_ws, rv_bytes, stream));
#undef NWS_STAGE
  if (g_prof.ev != nullptr && g_prof.used < g_prof.slots) ++g_prof.used;
  return NWS_OK;
}

int nws_profile_begin(int slots, unsigned stage_mask) {
  nws_profile_end();
  if (slots <= 0) return NWS_ERR_BAD_ARG;
  g_prof.ev = new hipEvent_t[(size_t)slots * kStages * 2];
  for (size_t i = 0; i < (size_t)slots * kStages * 2; ++i) {
    hipError_t e = hipEventCreate(&g_prof.ev[i]);
    if (e != hipSuccess) return (int)e;
  }
  g_prof.slots = slots;
  g_prof.used = 0;
  g_prof.mask = stage_mask;
  return NWS_OK;
}

int nws_profile_collect(float* ms_out, int* n_out) {
  if (!ms_out || !n_out) return NWS_ERR_BAD_ARG;
  const int n = g_prof.used;
  for (int i = 0; i < n; ++i)
    for (int s = 0; s < kStages; ++s) {
      float ms = -1.0f;
      if (g_prof.mask & (1u << s)) {
        hipEvent_t* e = g_prof.ev + ((size_t)i * kStages + s) * 2;
        hipError_t err = hipEventSynchronize(e[1]);
        if (err != hipSuccess) return (int)err;
        err = hipEventElapsedTime(&ms, e[0], e[1]);
        if (err != hipSuccess) return (int)err;
      }
      ms_out[(size_t)i * kStages + s] = ms;
    }
  *n_out = n;
  g_prof.used = g_prof.slots;  // stop recording until the next begin
  return NWS_OK;
}

int nws_profile_end(void) {
  if (g_prof.ev != nullptr) {
    for (size_t i = 0; i < (size_t)g_prof.slots * kStages * 2; ++i) (void)hipEventDestroy(g_prof.ev[i]);
    delete[] g_prof.ev;
  }
  g_prof = Profile();
  return NWS_OK;
}

}  // extern "C"
