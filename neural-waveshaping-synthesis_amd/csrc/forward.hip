// Whole NeuralWaveshaping.forward (models/neural_waveshaping.py:74-90) as one enqueue on one stream:
//   phase carries -> GRU -> frame MLPs -> fused exciter+NEWT -> FIR noise (+ branch sum) -> reverb,
// or as two halves (control: carries + GRU; audio: the rest) so that a throughput pipeline can run the latency-bound
// recurrence of the next batch on a side stream under the all-CU kernels of the current one.
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
  total += aligned((size_t)B * T * NWS_FIR_HALF * sizeof(float));   // fir (upper half-taps)
  total += aligned((size_t)B * N * sizeof(float));                  // noise branch (B, N)
  total += aligned((size_t)B * N * sizeof(float));                  // pre-reverb
  total += aligned(nws_reverb_workspace_bytes(plan, B));
  return total;
}

}  // extern "C"

namespace {

struct Arena {
  double* carry;
  float *gru_out, *film, *fir, *noise_out, *pre;
  void* rv_ws;
  size_t rv_bytes;
  bool ok;
};

// One layout for nws_forward, nws_forward_control (writes the head: carries, gru_out) and nws_forward_audio (reads the
// head, uses the rest).
Arena carve_arena(const NwsReverbPlan* plan, void* workspace, size_t bytes, int B, int T, bool head_only) {
  const size_t N = (size_t)T * NWS_HOP;
  Carve cv{static_cast<char*>(workspace), bytes};
  Arena a{};
  a.carry = static_cast<double*>(cv.take((size_t)B * (N / 32) * sizeof(double)));
  a.gru_out = static_cast<float*>(cv.take((size_t)B * T * NWS_HIDDEN * sizeof(float)));
  if (!head_only) {
    a.film = static_cast<float*>(cv.take((size_t)B * T * NWS_FILM_CH * sizeof(float)));
    a.fir = static_cast<float*>(cv.take((size_t)B * T * NWS_FIR_HALF * sizeof(float)));
    a.noise_out = static_cast<float*>(cv.take((size_t)B * N * sizeof(float)));
    a.pre = static_cast<float*>(cv.take((size_t)B * N * sizeof(float)));
    a.rv_bytes = nws_reverb_workspace_bytes(plan, B);
    a.rv_ws = cv.take(a.rv_bytes);
  }
  a.ok = cv.ok;
  return a;
}

#define NWS_STAGE(idx, call)                                   \
  do {                                                         \
    hipEvent_t* e__ = prof_events(idx);                        \
    if (e__) (void)hipEventRecord(e__[0], st);                 \
    rc = (call);                                               \
    if (e__) (void)hipEventRecord(e__[1], st);                 \
    if (rc != NWS_OK) return rc;                               \
  } while (0)

int control_half(const NwsWeights* w, const float* f0, const float* control, int B, int C, int T, int batched_gru,
                 const Arena& a, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  int rc = NWS_OK;
  if (batched_gru) {
    NWS_STAGE(0, nws_phase_carry(f0, nullptr, B, T, a.carry, stream));
    NWS_STAGE(1, nws_control_gru_batched(w, control, B, C, T, nullptr, a.gru_out, nullptr, stream));
  } else {
    // one launch: every GRU workgroup first leaves its utterance's phase carries (stage 0 has no launch of its own: its
    // profiling bracket is recorded empty)
    NWS_STAGE(0, NWS_OK);
    NWS_STAGE(1, nws_control_gru_carry(w, control, f0, B, C, T, a.gru_out, a.carry, stream));
  }
  return NWS_OK;
}

int audio_half(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
               const float* phase_u, const float* rand_phase, const float* noise, float* out, const Arena& a, void* stream,
               void* wait_before_exciter = nullptr, void* record_after_exciter = nullptr, bool with_reverb = true) {
  hipStream_t st = (hipStream_t)stream;
  int rc = NWS_OK;
  const size_t N = (size_t)T * NWS_HOP;
  NWS_STAGE(2, nws_frame_mlps(w, a.gru_out, aux->fir_design, B, T, nullptr, a.film, nullptr, a.fir, stream));
  // the noise branch first (its output starts the NEWT sum), then the oscillator kernel adds its own sum on top and leaves
  // pre = newt + noise (models/neural_waveshaping.py:77-81; a + b in either order is the same fp32 sum): the FIR-noise kernel
  // neither reads the other branch nor waits for the chain of exciter events
  NWS_STAGE(4, nws_fir_noise(a.fir, noise, nullptr, B, T, a.noise_out, stream));
  if (wait_before_exciter != nullptr) {
    const hipError_t e = hipStreamWaitEvent(st, (hipEvent_t)wait_before_exciter, 0);
    if (e != hipSuccess) return (int)e;
  }
  NWS_STAGE(3, nws_exciter_newt_add(w, f0, nullptr, a.carry, phase_u, rand_phase, a.film, a.noise_out, B, T, sample_rate, nullptr, a.pre, stream));
  if (record_after_exciter != nullptr) {
    const hipError_t e = hipEventRecord((hipEvent_t)record_after_exciter, st);
    if (e != hipSuccess) return (int)e;
  }
  if (with_reverb)
    NWS_STAGE(5, nws_reverb(aux->plan, aux->reverb_tables, aux->reverb_spectrum, a.pre, B, (int)N, out, a.rv_ws, a.rv_bytes, stream));
  if (g_prof.ev != nullptr && g_prof.used < g_prof.slots) ++g_prof.used;
  return NWS_OK;
}
#undef NWS_STAGE

bool aux_ok(const NwsForwardAux* aux) {
  return aux && aux->plan && aux->fir_design && aux->reverb_tables && aux->reverb_spectrum;
}

}  // namespace

extern "C" {

size_t nws_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(NwsWeights);
    case 1: return sizeof(NwsReverbPlan);
    case 2: return sizeof(NwsForwardAux);
    case 3: return sizeof(NwsShaperDesc);
    case 4: return sizeof(NwsGenericModel);
    default: return 0;
  }
}

size_t nws_forward_control_bytes(int B, int T) {
  if (B <= 0 || T <= 0) return 0;
  const size_t N = (size_t)T * NWS_HOP;
  return aligned((size_t)B * (N / 32) * sizeof(double)) + aligned((size_t)B * T * NWS_HIDDEN * sizeof(float));
}

int nws_forward(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, const float* control, int B, int C,
                int T, float sample_rate, const float* phase_u, const float* rand_phase, const float* noise, float* out,
                void* workspace, size_t workspace_bytes, void* stream) {
  if (!w || !aux_ok(aux)) return NWS_ERR_BAD_ARG;
  if (!f0 || !control || !phase_u || !rand_phase || !noise || !out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2 || C < 2) return NWS_ERR_BAD_ARG;
  if (T > (1 << 23) || !nws_reverb_plan_serves(aux->plan, T * NWS_HOP, 0)) return NWS_ERR_BAD_ARG;
  const Arena a = carve_arena(aux->plan, workspace, workspace_bytes, B, T, false);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  // one stream, one call: the per-utterance recurrence (64 CUs for 64 utterances) has the shorter latency
  const int rc = control_half(w, f0, control, B, C, T, 0, a, stream);
  if (rc != NWS_OK) return rc;
  return audio_half(w, aux, f0, B, T, sample_rate, phase_u, rand_phase, noise, out, a, stream);
}

int nws_forward_control(const NwsWeights* w, const float* f0, const float* control, int B, int C, int T, int batched_gru,
                        void* workspace, size_t workspace_bytes, void* stream) {
  if (!w || !f0 || !control || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2 || C < 2) return NWS_ERR_BAD_ARG;
  const Arena a = carve_arena(nullptr, workspace, workspace_bytes, B, T, true);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  return control_half(w, f0, control, B, C, T, batched_gru, a, stream);
}

int nws_forward_audio(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                      const float* phase_u, const float* rand_phase, const float* noise, float* out, void* workspace,
                      size_t workspace_bytes, void* stream) {
  if (!w || !aux_ok(aux)) return NWS_ERR_BAD_ARG;
  if (!f0 || !phase_u || !rand_phase || !noise || !out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2) return NWS_ERR_BAD_ARG;
  if (T > (1 << 23) || !nws_reverb_plan_serves(aux->plan, T * NWS_HOP, 0)) return NWS_ERR_BAD_ARG;
  const Arena a = carve_arena(aux->plan, workspace, workspace_bytes, B, T, false);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  return audio_half(w, aux, f0, B, T, sample_rate, phase_u, rand_phase, noise, out, a, stream);
}

int nws_forward_audio_ev(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                         const float* phase_u, const float* rand_phase, const float* noise, float* out, void* workspace,
                         size_t workspace_bytes, void* stream, void* wait_before_exciter, void* record_after_exciter) {
  if (!w || !aux_ok(aux)) return NWS_ERR_BAD_ARG;
  if (!f0 || !phase_u || !rand_phase || !noise || !out || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2) return NWS_ERR_BAD_ARG;
  if (T > (1 << 23) || !nws_reverb_plan_serves(aux->plan, T * NWS_HOP, 0)) return NWS_ERR_BAD_ARG;
  const Arena a = carve_arena(aux->plan, workspace, workspace_bytes, B, T, false);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  return audio_half(w, aux, f0, B, T, sample_rate, phase_u, rand_phase, noise, out, a, stream, wait_before_exciter,
                    record_after_exciter);
}

// The audio half in two parts, so that a multi-GPU caller can push sub-batches of finished waveforms to its peers while the
// reverb of the next sub-batch still runs ("gather sub-batches as they finish"): everything up to the reverb
// input for the whole batch, then the reverb of rows [row0, row0 + nrows) into the same rows of `out`.
int nws_forward_audio_pre(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                          const float* phase_u, const float* rand_phase, const float* noise, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (!w || !aux_ok(aux)) return NWS_ERR_BAD_ARG;
  if (!f0 || !phase_u || !rand_phase || !noise || !workspace) return NWS_ERR_BAD_ARG;
  if (B <= 0 || T < 2) return NWS_ERR_BAD_ARG;
  if (T > (1 << 23) || !nws_reverb_plan_serves(aux->plan, T * NWS_HOP, 0)) return NWS_ERR_BAD_ARG;
  const Arena a = carve_arena(aux->plan, workspace, workspace_bytes, B, T, false);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  return audio_half(w, aux, f0, B, T, sample_rate, phase_u, rand_phase, noise, nullptr, a, stream, nullptr, nullptr, false);
}

int nws_forward_reverb_rows(const NwsForwardAux* aux, int B, int T, int row0, int nrows, float* out /* (B, N): rows written */,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!aux_ok(aux) || !out || !workspace || B <= 0 || T < 2 || row0 < 0 || nrows <= 0 || row0 + nrows > B) return NWS_ERR_BAD_ARG;
  const Arena a = carve_arena(aux->plan, workspace, workspace_bytes, B, T, false);
  if (!a.ok) return NWS_ERR_WORKSPACE;
  const size_t N = (size_t)T * NWS_HOP;
  // the transform scratch of the whole batch is partitioned by rows (two utterances share one complex transform), so the
  // sub-batches of one forward may be in flight together as long as their rows are disjoint and row0 is even
  if (row0 & 1) return NWS_ERR_BAD_ARG;
  const size_t per_pair = 2 * (size_t)aux->plan->nblk * (size_t)aux->plan->L * sizeof(float);
  char* rv = static_cast<char*>(a.rv_ws) + (size_t)(row0 / 2) * per_pair;
  const size_t left = a.rv_bytes - (size_t)(row0 / 2) * per_pair;
  return nws_reverb(aux->plan, aux->reverb_tables, aux->reverb_spectrum, a.pre + (size_t)row0 * N, nrows, (int)N, out + (size_t)row0 * N, rv,
                    left, stream);
}

int nws_forward_audio_blocks(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                             const float* phase_u, const float* rand_phase, const float* noise, float* out, void* workspace,
                             size_t workspace_bytes, void* stream, const int32_t* row0, const int32_t* nrows, void* const* events,
                             int nblocks) {
  if (!row0 || !nrows || nblocks <= 0 || !out) return NWS_ERR_BAD_ARG;
  int next = 0;
  for (int q = 0; q < nblocks; ++q) {      // in order, covering [0, B), even sizes except the last (two utterances share a transform)
    if (row0[q] != next || nrows[q] <= 0 || ((nrows[q] & 1) && q != nblocks - 1)) return NWS_ERR_BAD_ARG;
    next += nrows[q];
  }
  if (next != B) return NWS_ERR_BAD_ARG;
  int rc = nws_forward_audio_pre(w, aux, f0, B, T, sample_rate, phase_u, rand_phase, noise, workspace, workspace_bytes, stream);
  if (rc != NWS_OK) return rc;
  for (int q = 0; q < nblocks; ++q) {
    rc = nws_forward_reverb_rows(aux, B, T, row0[q], nrows[q], out, workspace, workspace_bytes, stream);
    if (rc != NWS_OK) return rc;
    if (events != nullptr && events[q] != nullptr) {
      const hipError_t e = hipEventRecord((hipEvent_t)events[q], (hipStream_t)stream);
      if (e != hipSuccess) return (int)e;
    }
  }
  return NWS_OK;
}

int nws_profile_begin(int slots, unsigned stage_mask) {
  nws_profile_end();
  if (slots <= 0) return NWS_ERR_BAD_ARG;
  g_prof.ev = new hipEvent_t[(size_t)slots * kStages * 2];
  for (size_t i = 0; i < (size_t)slots * kStages * 2; ++i) {
    // timing only: no system-scope fence (cache writeback + invalidation) when the event is recorded - it would sit right in
    // front of and behind the kernel being timed, inside the step that is being timed
    hipError_t e = hipEventCreateWithFlags(&g_prof.ev[i], hipEventDisableSystemFence);
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
