// torch.ops.newt_hip.* - the PyTorch-ROCm custom-op layer over the C-ABI of include/nws_hip.h (last row).
//
// The reference's hot path is a chain of ATen ops dispatched from NeuralWaveshaping.forward (models/neural_waveshaping.py:74-90);
// this layer gives the HIP replacement the same standing: dispatcher-visible operators that take contiguous fp32 tensors,
// check shapes / dtypes / devices with TORCH_CHECK (-> RuntimeError, the reference's error convention), enqueue the
// hand-written kernels on torch's CURRENT stream of the tensors' device and return fresh tensors.  Nothing is computed
// here: every op is one call of an extern "C" launcher of libnws_hip.so.
//
// Weights travel as `wdesc`: a CPU uint8 tensor holding one NwsWeights struct (device pointers into the module's own
// parameters and derived tables; built once per weights version by engine.py, which keeps those tensors alive).
// Host-only translation unit: compiled with g++ against the torch headers, linked to libnws_hip.so (build.py).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/nws_hip.h"
#include "../../include/nws_hip_debug.h"

namespace {

using at::Tensor;
using OptTensor = c10::optional<at::Tensor>;

const NwsWeights* weights_of(const Tensor& wdesc) {
  TORCH_CHECK(wdesc.device().is_cpu() && wdesc.scalar_type() == at::kByte && wdesc.is_contiguous() &&
                  (size_t)wdesc.numel() == sizeof(NwsWeights),
              "wdesc: expected a contiguous CPU uint8 tensor of ", sizeof(NwsWeights), " bytes (one NwsWeights struct), got ",
              wdesc.sizes(), " ", wdesc.scalar_type(), " on ", wdesc.device());
  return reinterpret_cast<const NwsWeights*>(wdesc.data_ptr());
}

void check_dev(const Tensor& t, const char* name, at::ScalarType st = at::kFloat) {
  TORCH_CHECK(t.is_cuda(), name, " lives on ", t.device(),
              ": the NEWT forward path only runs as HIP kernels on an AMD GPU (there is no CPU fallback)");
  TORCH_CHECK(t.scalar_type() == st, name, ": expected ", st, ", got ", t.scalar_type());
  TORCH_CHECK(t.is_contiguous(), name, ": expected a contiguous tensor");
}

void check_same_device(const Tensor& a, const char* an, const Tensor& b, const char* bn) {
  TORCH_CHECK(a.device() == b.device(), an, " is on ", a.device(), " but ", bn, " is on ", b.device(),
              ": all tensors of one call must live on the same GPU");
}

void nws_check(int rc, const char* what) {
  TORCH_CHECK(rc == NWS_OK, what, " failed (", rc, "): ", nws_error_string(rc));
}

const float* fptr(const OptTensor& t) { return t.has_value() ? t->data_ptr<float>() : nullptr; }

// device guard + torch's current stream on that device: kernels go where the ATen ops they replace would have gone
struct Launch {
  c10::hip::HIPGuardMasqueradingAsCUDA guard;
  void* stream;
  explicit Launch(const Tensor& t)
      : guard(t.device()), stream(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream()) {}
};

NwsReverbPlan plan_of(const Tensor& plan) {
  TORCH_CHECK(plan.device().is_cpu() && plan.scalar_type() == at::kInt && plan.is_contiguous() && plan.numel() == 8,
              "plan: expected a CPU int32 tensor [L, N1, N2, Lc, hist, nblk, 0, 0] (the NwsReverbPlan of nws_reverb_plan)");
  const int32_t* p = plan.data_ptr<int32_t>();
  return NwsReverbPlan{p[0], p[1], p[2], p[3], p[4], p[5], {p[6], p[7]}};
}

// the spectrum / table tensors must be the ones built for THIS plan: the kernels index them by plan.L / N1 / N2 (a tensor made
// for another L, or for an older layout of the spectrum buffer, would be read out of bounds)
void check_reverb_buffers(const NwsReverbPlan& plan, const Tensor& tables, const Tensor& spectrum) {
  TORCH_CHECK(plan.L > 0 && plan.N1 > 0 && plan.N2 > 0 && (int64_t)plan.N1 * plan.N2 == plan.L, "plan: inconsistent [L, N1, N2] = [",
              plan.L, ", ", plan.N1, ", ", plan.N2, "]");
  TORCH_CHECK((size_t)tables.numel() * tables.element_size() == nws_reverb_table_bytes(&plan), "reverb_tables: ",
              (size_t)tables.numel() * tables.element_size(), " bytes, the plan (L = ", plan.L, ") needs ", nws_reverb_table_bytes(&plan));
  TORCH_CHECK((size_t)spectrum.numel() * spectrum.element_size() == nws_reverb_spectrum_bytes(&plan), "reverb_spectrum: ",
              (size_t)spectrum.numel() * spectrum.element_size(), " bytes, the plan (L = ", plan.L, ") needs ",
              nws_reverb_spectrum_bytes(&plan), " (Sre | Sim | ir_)");
}

struct Aux {
  NwsReverbPlan plan;
  NwsForwardAux aux;
  Aux(const Tensor& fir_design, const Tensor& plan_t, const Tensor& tables, const Tensor& spectrum) : plan(plan_of(plan_t)) {
    check_dev(fir_design, "fir_design");
    check_dev(tables, "reverb_tables");
    check_dev(spectrum, "reverb_spectrum");
    TORCH_CHECK(fir_design.numel() == NWS_FIR_LEN * 132, "fir_design: expected (256, 132)");
    check_reverb_buffers(plan, tables, spectrum);
    aux.fir_design = fir_design.data_ptr<float>();
    aux.plan = &plan;
    aux.reverb_tables = tables.data_ptr();
    aux.reverb_spectrum = spectrum.data_ptr();
  }
};

void check_inputs(const Tensor& f0, const Tensor& control, int64_t& B, int64_t& C, int64_t& T) {
  check_dev(f0, "f0");
  check_dev(control, "control");
  check_same_device(f0, "f0", control, "control");
  TORCH_CHECK(f0.dim() == 3 && f0.size(1) == 1, "f0: expected (B, 1, T), got ", f0.sizes());
  TORCH_CHECK(control.dim() == 3 && control.size(1) >= 2, "control: expected (B, C>=2, T), got ", control.sizes());
  B = f0.size(0);
  T = f0.size(2);
  C = control.size(1);
  TORCH_CHECK(control.size(0) == B && control.size(2) == T, "f0 ", f0.sizes(), " and control ", control.sizes(),
              " disagree on batch / frames");
  TORCH_CHECK(T >= 2, "need at least 2 control frames (reflect padding of the noise STFT, generators.py:31)");
}

void check_draws(const Tensor& phase_u, const Tensor& rand_phase, const Tensor& noise, int64_t T, const Tensor& like) {
  check_dev(phase_u, "phase_u");
  check_dev(rand_phase, "rand_phase");
  check_dev(noise, "noise");
  check_same_device(like, "f0", phase_u, "phase_u");
  check_same_device(like, "f0", noise, "noise");
  check_same_device(like, "f0", rand_phase, "osc.rand_phase");
  TORCH_CHECK(phase_u.numel() == NWS_N_HARMONICS, "phase_u: expected 101 elements, got ", phase_u.sizes());
  TORCH_CHECK(rand_phase.numel() == NWS_N_HARMONICS, "rand_phase: expected 101 elements, got ", rand_phase.sizes());
  TORCH_CHECK(noise.numel() == NWS_HOP * T - 1, "noise: expected ", NWS_HOP * T - 1, " elements, got ", noise.sizes());
}

// ---- whole forward: models/neural_waveshaping.py:74-90 ---------------------------------------------------------------
Tensor forward(const Tensor& wdesc, const Tensor& f0, const Tensor& control, const Tensor& phase_u, const Tensor& rand_phase,
               const Tensor& noise, const Tensor& fir_design, const Tensor& plan, const Tensor& reverb_tables,
               const Tensor& reverb_spectrum, Tensor& workspace, double sample_rate) {
  const NwsWeights* w = weights_of(wdesc);
  int64_t B, C, T;
  check_inputs(f0, control, B, C, T);
  check_draws(phase_u, rand_phase, noise, T, f0);
  check_dev(workspace, "workspace", at::kByte);
  check_same_device(f0, "f0", workspace, "workspace");
  Aux a(fir_design, plan, reverb_tables, reverb_spectrum);
  check_same_device(f0, "f0", fir_design, "the model's tables");
  Launch L(f0);
  Tensor out = at::empty({B, T * NWS_HOP}, f0.options());
  nws_check(nws_forward(w, &a.aux, f0.data_ptr<float>(), control.data_ptr<float>(), (int)B, (int)C, (int)T, (float)sample_rate,
                        phase_u.data_ptr<float>(), rand_phase.data_ptr<float>(), noise.data_ptr<float>(), out.data_ptr<float>(),
                        workspace.data_ptr(), (size_t)workspace.numel(), L.stream),
            "nws_forward");
  return out;
}

void forward_control(const Tensor& wdesc, const Tensor& f0, const Tensor& control, Tensor& workspace, bool batched_gru) {
  const NwsWeights* w = weights_of(wdesc);
  int64_t B, C, T;
  check_inputs(f0, control, B, C, T);
  check_dev(workspace, "workspace", at::kByte);
  check_same_device(f0, "f0", workspace, "workspace");
  Launch L(f0);
  nws_check(nws_forward_control(w, f0.data_ptr<float>(), control.data_ptr<float>(), (int)B, (int)C, (int)T, batched_gru ? 1 : 0,
                                workspace.data_ptr(), (size_t)workspace.numel(), L.stream),
            "nws_forward_control");
}

Tensor forward_audio(const Tensor& wdesc, const Tensor& f0, const Tensor& phase_u, const Tensor& rand_phase, const Tensor& noise,
                     const Tensor& fir_design, const Tensor& plan, const Tensor& reverb_tables, const Tensor& reverb_spectrum,
                     Tensor& workspace, double sample_rate, const OptTensor& out_opt, int64_t wait_event, int64_t record_event) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(f0, "f0");
  TORCH_CHECK(f0.dim() == 3 && f0.size(1) == 1 && f0.size(2) >= 2, "f0: expected (B, 1, T>=2), got ", f0.sizes());
  const int64_t B = f0.size(0), T = f0.size(2);
  check_draws(phase_u, rand_phase, noise, T, f0);
  check_dev(workspace, "workspace", at::kByte);
  check_same_device(f0, "f0", workspace, "workspace");
  Aux a(fir_design, plan, reverb_tables, reverb_spectrum);
  Launch L(f0);
  Tensor out;
  if (out_opt.has_value()) {
    out = *out_opt;
    check_dev(out, "out");
    check_same_device(f0, "f0", out, "out");
    TORCH_CHECK(out.dim() == 2 && out.size(0) == B && out.size(1) == T * NWS_HOP, "out: expected (", B, ", ", T * NWS_HOP, ")");
  } else {
    out = at::empty({B, T * NWS_HOP}, f0.options());
  }
  // wait_event / record_event: raw hipEvent_t handles (torch.cuda.Event.cuda_event) or 0, see nws_forward_audio_ev
  nws_check(nws_forward_audio_ev(w, &a.aux, f0.data_ptr<float>(), (int)B, (int)T, (float)sample_rate, phase_u.data_ptr<float>(),
                                 rand_phase.data_ptr<float>(), noise.data_ptr<float>(), out.data_ptr<float>(), workspace.data_ptr(),
                                 (size_t)workspace.numel(), L.stream, reinterpret_cast<void*>(wait_event),
                                 reinterpret_cast<void*>(record_event)),
            "nws_forward_audio");
  return out;
}

// the audio half in two parts (sub-batch gathers): up to the reverb input; then the reverb of a block of rows
void forward_audio_pre(const Tensor& wdesc, const Tensor& f0, const Tensor& phase_u, const Tensor& rand_phase, const Tensor& noise,
                       const Tensor& fir_design, const Tensor& plan, const Tensor& reverb_tables, const Tensor& reverb_spectrum,
                       Tensor& workspace, double sample_rate) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(f0, "f0");
  TORCH_CHECK(f0.dim() == 3 && f0.size(1) == 1 && f0.size(2) >= 2, "f0: expected (B, 1, T>=2), got ", f0.sizes());
  const int64_t B = f0.size(0), T = f0.size(2);
  check_draws(phase_u, rand_phase, noise, T, f0);
  check_dev(workspace, "workspace", at::kByte);
  check_same_device(f0, "f0", workspace, "workspace");
  Aux a(fir_design, plan, reverb_tables, reverb_spectrum);
  Launch L(f0);
  nws_check(nws_forward_audio_pre(w, &a.aux, f0.data_ptr<float>(), (int)B, (int)T, (float)sample_rate, phase_u.data_ptr<float>(),
                                  rand_phase.data_ptr<float>(), noise.data_ptr<float>(), workspace.data_ptr(),
                                  (size_t)workspace.numel(), L.stream), "nws_forward_audio_pre");
}

void forward_reverb_rows(const Tensor& fir_design, const Tensor& plan, const Tensor& reverb_tables, const Tensor& reverb_spectrum,
                         Tensor& workspace, int64_t T, int64_t row0, int64_t nrows, Tensor& out) {
  check_dev(workspace, "workspace", at::kByte);
  check_dev(out, "out");
  check_same_device(out, "out", workspace, "workspace");
  TORCH_CHECK(out.dim() == 2 && out.size(1) == T * NWS_HOP, "out: expected (B, ", T * NWS_HOP, "), got ", out.sizes());
  TORCH_CHECK(row0 >= 0 && nrows > 0 && row0 + nrows <= out.size(0) && (row0 & 1) == 0, "forward_reverb_rows: bad row block");
  Aux a(fir_design, plan, reverb_tables, reverb_spectrum);
  Launch L(out);
  nws_check(nws_forward_reverb_rows(&a.aux, (int)out.size(0), (int)T, (int)row0, (int)nrows, out.data_ptr<float>(), workspace.data_ptr(),
                                    (size_t)workspace.numel(), L.stream), "nws_forward_reverb_rows");
}

// the same two in one call for a fixed list of row blocks, an event recorded behind each block (sub-batch exchange: one op per step)
void forward_audio_blocks(const Tensor& wdesc, const Tensor& f0, const Tensor& phase_u, const Tensor& rand_phase, const Tensor& noise,
                          const Tensor& fir_design, const Tensor& plan, const Tensor& reverb_tables, const Tensor& reverb_spectrum,
                          Tensor& workspace, double sample_rate, Tensor& out, at::IntArrayRef row0, at::IntArrayRef nrows,
                          at::IntArrayRef events) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(f0, "f0");
  TORCH_CHECK(f0.dim() == 3 && f0.size(1) == 1 && f0.size(2) >= 2, "f0: expected (B, 1, T>=2), got ", f0.sizes());
  const int64_t B = f0.size(0), T = f0.size(2);
  check_draws(phase_u, rand_phase, noise, T, f0);
  check_dev(workspace, "workspace", at::kByte);
  check_dev(out, "out");
  check_same_device(f0, "f0", workspace, "workspace");
  check_same_device(f0, "f0", out, "out");
  TORCH_CHECK(out.dim() == 2 && out.size(0) == B && out.size(1) == T * NWS_HOP && out.is_contiguous(), "out: expected a contiguous (", B, ", ",
              T * NWS_HOP, "), got ", out.sizes());
  TORCH_CHECK(!row0.empty() && row0.size() == nrows.size() && (events.empty() || events.size() == row0.size()),
              "forward_audio_blocks: row0 / nrows / events must have one entry per block");
  std::vector<int32_t> r0(row0.begin(), row0.end()), nr(nrows.begin(), nrows.end());
  std::vector<void*> ev(events.size());
  for (size_t q = 0; q < events.size(); ++q) ev[q] = reinterpret_cast<void*>(static_cast<uintptr_t>(events[q]));
  Aux a(fir_design, plan, reverb_tables, reverb_spectrum);
  Launch L(f0);
  nws_check(nws_forward_audio_blocks(w, &a.aux, f0.data_ptr<float>(), (int)B, (int)T, (float)sample_rate, phase_u.data_ptr<float>(),
                                     rand_phase.data_ptr<float>(), noise.data_ptr<float>(), out.data_ptr<float>(), workspace.data_ptr(),
                                     (size_t)workspace.numel(), L.stream, r0.data(), nr.data(), ev.empty() ? nullptr : ev.data(), (int)r0.size()),
            "nws_forward_audio_blocks");
}

// ---- stages ------------------------------------------------------------------------------------------------------------
// exclusive fp64 prefix sums at 32-sample granularity (torch.cumsum of generators.py:59): f0 (B, T) frames or f0_up (B, 128 T)
Tensor phase_carry(const OptTensor& f0, const OptTensor& f0_up) {
  TORCH_CHECK(f0.has_value() != f0_up.has_value(), "phase_carry: give exactly one of f0 (B, T) and f0_up (B, 128 T)");
  const Tensor& src = f0.has_value() ? *f0 : *f0_up;
  check_dev(src, f0.has_value() ? "f0" : "f0_up");
  TORCH_CHECK(src.dim() == 2, "phase_carry: expected a 2-D tensor, got ", src.sizes());
  const int64_t B = src.size(0);
  TORCH_CHECK(f0.has_value() || src.size(1) % NWS_HOP == 0, "f0_up: the sample count must be a multiple of ", NWS_HOP);
  const int64_t T = f0.has_value() ? src.size(1) : src.size(1) / NWS_HOP;
  Launch L(src);
  Tensor carry = at::empty({B, T * NWS_HOP / 32}, src.options().dtype(at::kDouble));
  nws_check(nws_phase_carry(fptr(f0), fptr(f0_up), (int)B, (int)T, carry.data_ptr<double>(), L.stream), "nws_phase_carry");
  return carry;
}

std::tuple<Tensor, Tensor> exciter_newt(const Tensor& wdesc, const OptTensor& f0, const OptTensor& f0_up, const Tensor& carry,
                                        const Tensor& phase_u, const Tensor& rand_phase, const OptTensor& film,
                                        double sample_rate, bool want_exciter, bool want_newt) {
  const NwsWeights* w = weights_of(wdesc);
  TORCH_CHECK(f0.has_value() != f0_up.has_value(), "exciter_newt: give exactly one of f0 (B, T) and f0_up (B, 128 T)");
  TORCH_CHECK(want_exciter || want_newt, "exciter_newt: nothing to compute");
  const Tensor& src = f0.has_value() ? *f0 : *f0_up;
  check_dev(src, "f0");
  check_dev(carry, "carry", at::kDouble);
  check_dev(phase_u, "phase_u");
  check_dev(rand_phase, "rand_phase");
  check_same_device(src, "f0", carry, "carry");
  check_same_device(src, "f0", phase_u, "phase_u");
  TORCH_CHECK(src.dim() == 2, "exciter_newt: f0 / f0_up must be 2-D, got ", src.sizes());
  const int64_t B = src.size(0);
  const int64_t T = f0.has_value() ? src.size(1) : src.size(1) / NWS_HOP;
  const int64_t N = T * NWS_HOP;
  TORCH_CHECK(f0.has_value() || src.size(1) == N, "f0_up: the sample count must be a multiple of ", NWS_HOP);
  TORCH_CHECK(carry.numel() == B * (N / 32), "carry: expected (", B, ", ", N / 32, "), got ", carry.sizes());
  TORCH_CHECK(phase_u.numel() == NWS_N_HARMONICS && rand_phase.numel() == NWS_N_HARMONICS, "phase_u / rand_phase: 101 elements each");
  if (want_newt) {
    TORCH_CHECK(film.has_value(), "exciter_newt: the waveshaper bank needs the FiLM parameters (film)");
    check_dev(*film, "film");
    check_same_device(src, "f0", *film, "film");
    TORCH_CHECK(film->numel() == B * T * NWS_FILM_CH, "film: expected (", B, ", ", T, ", 256), got ", film->sizes());
  }
  Launch L(src);
  Tensor exc = want_exciter ? at::empty({B, NWS_N_SHAPERS, N}, src.options()) : at::empty({0}, src.options());
  Tensor out = want_newt ? at::empty({B, N}, src.options()) : at::empty({0}, src.options());
  nws_check(nws_exciter_newt(w, fptr(f0), fptr(f0_up), carry.data_ptr<double>(), phase_u.data_ptr<float>(),
                             rand_phase.data_ptr<float>(), fptr(film), (int)B, (int)T, (float)sample_rate,
                             want_exciter ? exc.data_ptr<float>() : nullptr, want_newt ? out.data_ptr<float>() : nullptr, L.stream),
            "nws_exciter_newt");
  return {exc, out};
}

// HarmonicOscillator.forward (generators.py:58-66): f0_up (B, N) -> (B, 101, N)
Tensor oscillator(const Tensor& f0_up, const Tensor& phase_u, const Tensor& rand_phase, double sample_rate) {
  check_dev(f0_up, "f0");
  check_dev(phase_u, "phase_u");
  check_dev(rand_phase, "rand_phase");
  check_same_device(f0_up, "f0", phase_u, "phase_u");
  check_same_device(f0_up, "f0", rand_phase, "rand_phase");
  TORCH_CHECK(f0_up.dim() == 2, "HarmonicOscillator: expected f0 of shape (B, N), got ", f0_up.sizes());
  TORCH_CHECK(f0_up.size(1) % NWS_HOP == 0 && f0_up.size(1) > 0,
              "HarmonicOscillator on the HIP path: the sample count must be a multiple of ", NWS_HOP, ", got ", f0_up.size(1));
  TORCH_CHECK(phase_u.numel() == NWS_N_HARMONICS && rand_phase.numel() == NWS_N_HARMONICS,
              "kernels are specialised for 101 harmonics (gin/models/newt.gin)");
  const int64_t B = f0_up.size(0), N = f0_up.size(1);
  Launch L(f0_up);
  Tensor carry = at::empty({B, N / 32}, f0_up.options().dtype(at::kDouble));
  nws_check(nws_phase_carry(nullptr, f0_up.data_ptr<float>(), (int)B, (int)(N / NWS_HOP), carry.data_ptr<double>(), L.stream),
            "nws_phase_carry");
  Tensor out = at::empty({B, NWS_N_HARMONICS, N}, f0_up.options());
  nws_check(nws_oscillator(f0_up.data_ptr<float>(), carry.data_ptr<double>(), phase_u.data_ptr<float>(),
                           rand_phase.data_ptr<float>(), (int)B, (int)N, (float)sample_rate, out.data_ptr<float>(), L.stream),
            "nws_oscillator");
  return out;
}

// GRU(2 -> 128) over T frames of control[:, 0:2] (models/neural_waveshaping.py:24-25): -> (gru_out (B, T, 128), hT (B, 128))
std::tuple<Tensor, Tensor> control_gru(const Tensor& wdesc, const Tensor& control, const OptTensor& h0, bool batched) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(control, "control");
  TORCH_CHECK(control.dim() == 3 && control.size(1) >= 2, "control: expected (B, C>=2, T), got ", control.sizes());
  const int64_t B = control.size(0), C = control.size(1), T = control.size(2);
  TORCH_CHECK(T >= 1, "control: need at least one frame");
  if (h0.has_value()) {
    check_dev(*h0, "h0");
    check_same_device(control, "control", *h0, "h0");
    TORCH_CHECK(h0->numel() == B * NWS_HIDDEN, "h0: expected (", B, ", 128), got ", h0->sizes());
  }
  Launch L(control);
  Tensor out = at::empty({B, T, NWS_HIDDEN}, control.options());
  Tensor hT = at::empty({B, NWS_HIDDEN}, control.options());
  if (batched)
    nws_check(nws_control_gru_batched(w, control.data_ptr<float>(), (int)B, (int)C, (int)T, fptr(h0), out.data_ptr<float>(),
                                      hT.data_ptr<float>(), L.stream), "nws_control_gru_batched");
  else
    nws_check(nws_control_gru_state(w, control.data_ptr<float>(), (int)B, (int)C, (int)T, fptr(h0), out.data_ptr<float>(),
                                    hT.data_ptr<float>(), L.stream), "nws_control_gru_state");
  return {out, hT};
}

// proj + newt.mlp + h_generator + FIR design in one kernel: -> (emb (B,128,T), film (B,T,256), H (B,T,129), fir (B,T,128) upper half-taps)
std::tuple<Tensor, Tensor, Tensor, Tensor> frame_mlps(const Tensor& wdesc, const Tensor& gru_out, const Tensor& fir_design,
                                                       bool want_emb, bool want_H) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(gru_out, "gru_out");
  check_dev(fir_design, "fir_design");
  check_same_device(gru_out, "gru_out", fir_design, "fir_design");
  TORCH_CHECK(gru_out.dim() == 3 && gru_out.size(2) == NWS_HIDDEN, "gru_out: expected (B, T, 128), got ", gru_out.sizes());
  TORCH_CHECK(fir_design.numel() == NWS_FIR_LEN * 132, "fir_design: expected (256, 132)");
  const int64_t B = gru_out.size(0), T = gru_out.size(1);
  Launch L(gru_out);
  const auto o = gru_out.options();
  Tensor emb = want_emb ? at::empty({B, NWS_HIDDEN, T}, o) : at::empty({0}, o);
  Tensor film = at::empty({B, T, NWS_FILM_CH}, o);
  Tensor H = want_H ? at::empty({B, T, NWS_N_BANDS}, o) : at::empty({0}, o);
  Tensor fir = at::empty({B, T, NWS_FIR_HALF}, o);
  nws_check(nws_frame_mlps(w, gru_out.data_ptr<float>(), fir_design.data_ptr<float>(), (int)B, (int)T,
                           want_emb ? emb.data_ptr<float>() : nullptr, film.data_ptr<float>(),
                           want_H ? H.data_ptr<float>() : nullptr, fir.data_ptr<float>(), L.stream), "nws_frame_mlps");
  return {emb, film, H, fir};
}

// time-varying FIR noise (generators.py:30-35) + optional branch sum; origin / noise_len < 0: the reference's own framing
Tensor fir_noise(const Tensor& fir, const Tensor& noise, const OptTensor& add_in, int64_t origin) {
  check_dev(fir, "fir");
  check_dev(noise, "noise");
  check_same_device(fir, "fir", noise, "noise");
  TORCH_CHECK(fir.dim() == 3 && fir.size(2) == NWS_FIR_HALF, "fir: expected (B, T, 128) upper half-taps (nws_frame_mlps), got ", fir.sizes());
  const int64_t B = fir.size(0), T = fir.size(1);
  if (add_in.has_value()) {
    check_dev(*add_in, "add_in");
    check_same_device(fir, "fir", *add_in, "add_in");
    TORCH_CHECK(add_in->numel() == B * T * NWS_HOP, "add_in: expected (", B, ", ", T * NWS_HOP, "), got ", add_in->sizes());
  }
  Launch L(fir);
  Tensor out = at::empty({B, T * NWS_HOP}, fir.options());
  if (origin < 0) {
    TORCH_CHECK(noise.numel() == T * NWS_HOP - 1, "noise: expected ", T * NWS_HOP - 1, " samples, got ", noise.sizes());
    nws_check(nws_fir_noise(fir.data_ptr<float>(), noise.data_ptr<float>(), fptr(add_in), (int)B, (int)T, out.data_ptr<float>(),
                            L.stream), "nws_fir_noise");
  } else {
    nws_check(nws_fir_noise_window(fir.data_ptr<float>(), noise.data_ptr<float>(), (int)noise.numel(), (int)origin, fptr(add_in),
                                   (int)B, (int)T, out.data_ptr<float>(), L.stream), "nws_fir_noise_window");
  }
  return out;
}

// FIRNoiseSynth's zero-phase FIR design from H (B, 129, T) (generators.py:22-28) -> upper half-taps (B, T, 128)
Tensor fir_from_h(const Tensor& H, const Tensor& fir_design) {
  check_dev(H, "H");
  check_dev(fir_design, "fir_design");
  check_same_device(H, "H", fir_design, "fir_design");
  TORCH_CHECK(H.dim() == 3 && H.size(1) == NWS_N_BANDS, "FIRNoiseSynth: expected H of shape (B, 129, T), got ", H.sizes());
  TORCH_CHECK(fir_design.numel() == NWS_FIR_LEN * 132, "fir_design: expected (256, 132)");
  Launch L(H);
  Tensor fir = at::empty({H.size(0), H.size(2), NWS_FIR_HALF}, H.options());
  nws_check(nws_fir_from_h(H.data_ptr<float>(), fir_design.data_ptr<float>(), (int)H.size(0), (int)H.size(2), fir.data_ptr<float>(),
                           L.stream), "nws_fir_from_h");
  return fir;
}

// Reverb.forward (shaping.py:161-173): x (B, N) -> x + circconv_L(x, [0, ir])[:N]
Tensor reverb(const Tensor& plan_t, const Tensor& tables, const Tensor& spectrum, const Tensor& x) {
  NwsReverbPlan plan = plan_of(plan_t);
  check_dev(x, "x");
  check_dev(tables, "reverb_tables");
  check_dev(spectrum, "reverb_spectrum");
  check_same_device(x, "x", tables, "reverb tables");
  check_same_device(x, "x", spectrum, "reverb.ir spectrum");
  TORCH_CHECK(x.dim() == 2, "Reverb: expected (B, N), got ", x.sizes());
  check_reverb_buffers(plan, tables, spectrum);
  const int64_t B = x.size(0), N = x.size(1);
  TORCH_CHECK(nws_reverb_plan_serves(&plan, (int)N, 0), "Reverb: the plan [L ", plan.L, ", Lc ", plan.Lc, ", hist ", plan.hist, ", nblk ", plan.nblk,
              "] was not made for ", N, " samples");
  Launch L(x);
  const size_t nbytes = nws_reverb_workspace_bytes(&plan, (int)B);
  Tensor ws = at::empty({(int64_t)nbytes}, x.options().dtype(at::kByte));
  Tensor y = at::empty_like(x);
  nws_check(nws_reverb(&plan, tables.data_ptr(), spectrum.data_ptr(), x.data_ptr<float>(), (int)B, (int)N, y.data_ptr<float>(),
                       ws.data_ptr(), nbytes, L.stream), "nws_reverb");
  return y;
}

// streaming (linear) reverb chunk: -> (y (B, M), tail_out (B, tail_len))
std::tuple<Tensor, Tensor> reverb_linear_chunk(const Tensor& plan_t, const Tensor& tables, const Tensor& spectrum, const Tensor& x,
                                               const Tensor& tail_in) {
  NwsReverbPlan plan = plan_of(plan_t);
  check_dev(x, "x");
  check_dev(tail_in, "tail_in");
  check_dev(tables, "reverb_tables");
  check_dev(spectrum, "reverb_spectrum");
  check_same_device(x, "x", tail_in, "tail_in");
  check_same_device(x, "x", tables, "reverb tables");
  TORCH_CHECK(x.dim() == 2 && tail_in.dim() == 2 && tail_in.size(0) == x.size(0), "reverb_linear_chunk: x (B, M), tail_in (B, tail)");
  const int64_t B = x.size(0), M = x.size(1), tail_len = tail_in.size(1);
  check_reverb_buffers(plan, tables, spectrum);
  check_same_device(x, "x", spectrum, "reverb.ir spectrum");
  TORCH_CHECK(plan.Lc == 0, "reverb_linear_chunk: needs a direct plan (one transform of M + tail_len samples)");
  TORCH_CHECK(M + tail_len <= plan.L, "reverb_linear_chunk: chunk of ", M, " + tail of ", tail_len, " samples wraps around L = ", plan.L);
  Launch L(x);
  const size_t nbytes = (size_t)(2 * ((B + 1) / 2) * plan.L + B * plan.L) * sizeof(float);
  Tensor ws = at::empty({(int64_t)nbytes}, x.options().dtype(at::kByte));
  Tensor y = at::empty_like(x);
  Tensor tail_out = at::empty_like(tail_in);
  nws_check(nws_reverb_linear_chunk(&plan, tables.data_ptr(), spectrum.data_ptr(), x.data_ptr<float>(), (int)B, (int)M,
                                    tail_in.data_ptr<float>(), tail_out.data_ptr<float>(), (int)tail_len, y.data_ptr<float>(),
                                    ws.data_ptr(), nbytes, L.stream), "nws_reverb_linear_chunk");
  return {y, tail_out};
}

// TrainableNonlinearity.forward / FastNEWT.shaping_fn on (B, 64, N) (shaping.py:36-37, :136-151)
Tensor shaper_apply(const Tensor& wdesc, const Tensor& x) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(x, "x");
  TORCH_CHECK(x.dim() == 3 && x.size(1) == NWS_N_SHAPERS, "expected (B, 64, N), got ", x.sizes());
  Launch L(x);
  Tensor y = at::empty_like(x);
  nws_check(nws_shaper_apply(w, x.data_ptr<float>(), x.size(0), x.size(2), y.data_ptr<float>(), L.stream), "nws_shaper_apply");
  return y;
}

// FastNEWT._init_lookup_table (shaping.py:107-119)
Tensor shaper_table(const Tensor& wdesc, const Tensor& like, int64_t size, double tmin, double tmax) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(like, "shaping_fn.input_scale");
  TORCH_CHECK(size >= 2 && tmax > tmin, "FastNEWT: need table_size >= 2 and table_max > table_min");
  Launch L(like);
  Tensor table = at::empty({NWS_N_SHAPERS, size}, like.options());
  nws_check(nws_shaper_table(w, (int)size, (float)tmin, (float)tmax, table.data_ptr<float>(), L.stream), "nws_shaper_table");
  return table;
}

// NEWT.forward / FastNEWT.forward on a materialised exciter (shaping.py:67-79): film (B, 256, T) channel-major
Tensor newt_apply(const Tensor& wdesc, const Tensor& exciter, const Tensor& film) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(exciter, "exciter");
  check_dev(film, "film_params");
  check_same_device(exciter, "exciter", film, "control embedding");
  TORCH_CHECK(exciter.dim() == 3 && exciter.size(1) == NWS_N_SHAPERS, "NEWT: expected an exciter of shape (B, 64, N), got ", exciter.sizes());
  TORCH_CHECK(film.dim() == 3 && film.size(1) == NWS_FILM_CH && film.size(0) == exciter.size(0),
              "NEWT: expected FiLM parameters of shape (B, 256, T), got ", film.sizes());
  const int64_t B = exciter.size(0), T = film.size(2);
  TORCH_CHECK(exciter.size(2) == T * NWS_HOP, "NEWT: exciter has ", exciter.size(2), " samples, the control embedding ", T,
              " frames (x128 = ", T * NWS_HOP, ")");
  Launch L(exciter);
  Tensor out = at::empty({B, 1, T * NWS_HOP}, exciter.options());
  nws_check(nws_newt_apply(w, exciter.data_ptr<float>(), film.data_ptr<float>(), (int)B, (int)T, out.data_ptr<float>(), L.stream),
            "nws_newt_apply");
  return out;
}

// TimeDistributedMLP.forward (dynamic.py:20-40): x (B, in, T); weights[i] (rows_i, cols_i[, 1]), biases[i]; ln_w / ln_b for
// every layer but the last
Tensor td_mlp(const Tensor& x, at::TensorList weights, at::TensorList biases, at::TensorList ln_w, at::TensorList ln_b, double eps,
              double slope) {
  check_dev(x, "x");
  TORCH_CHECK(x.dim() == 3, "TimeDistributedMLP: expected (B, C, T), got ", x.sizes());
  const int depth = (int)weights.size();
  TORCH_CHECK(depth >= 1 && depth <= 8 && (int)biases.size() == depth && (int)ln_w.size() == depth - 1 &&
                  (int)ln_b.size() == depth - 1, "TimeDistributedMLP: inconsistent layer lists");
  const int64_t B = x.size(0), in_size = x.size(1), T = x.size(2);
  const int64_t hidden = weights[0].size(0), out_size = weights[depth - 1].size(0);
  const float* wp[8];
  const float* bp[8];
  const float* gp[8] = {nullptr};
  const float* lp[8] = {nullptr};
  for (int i = 0; i < depth; ++i) {
    check_dev(weights[i], "net weight");
    check_dev(biases[i], "net bias");
    check_same_device(x, "x", weights[i], "the MLP's parameters");
    const int64_t rows = i == depth - 1 ? out_size : hidden, cols = i == 0 ? in_size : hidden;
    TORCH_CHECK(weights[i].numel() == rows * cols && weights[i].size(0) == rows, "TimeDistributedMLP layer ", i, ": weight ",
                weights[i].sizes(), " does not match (", rows, ", ", cols, ") - input has ", in_size, " channels");
    TORCH_CHECK(biases[i].numel() == rows, "TimeDistributedMLP layer ", i, ": bias ", biases[i].sizes());
    wp[i] = weights[i].data_ptr<float>();
    bp[i] = biases[i].data_ptr<float>();
    if (i < depth - 1) {
      check_dev(ln_w[i], "layer_norm weight");
      check_dev(ln_b[i], "layer_norm bias");
      TORCH_CHECK(ln_w[i].numel() == hidden && ln_b[i].numel() == hidden, "LayerNorm ", i, ": expected ", hidden, " elements");
      gp[i] = ln_w[i].data_ptr<float>();
      lp[i] = ln_b[i].data_ptr<float>();
    }
  }
  Launch L(x);
  Tensor y = at::empty({B, out_size, T}, x.options());
  nws_check(nws_td_mlp(x.data_ptr<float>(), (int)B, (int)in_size, (int)hidden, (int)out_size, depth, (int)T, wp, bp, gp, lp,
                       (float)eps, (float)slope, y.data_ptr<float>(), L.stream), "nws_td_mlp");
  return y;
}

Tensor td_layer_norm(const Tensor& x, const Tensor& weight, const Tensor& bias, double eps) {
  check_dev(x, "x");
  check_dev(weight, "layer_norm.weight");
  check_dev(bias, "layer_norm.bias");
  check_same_device(x, "x", weight, "layer_norm.weight");
  TORCH_CHECK(x.dim() == 3, "TimeDistributedLayerNorm: expected (B, C, T), got ", x.sizes());
  TORCH_CHECK(weight.numel() == x.size(1) && bias.numel() == x.size(1), "TimeDistributedLayerNorm: input has ", x.size(1),
              " channels, the LayerNorm ", weight.numel());
  Launch L(x);
  Tensor y = at::empty_like(x);
  nws_check(nws_td_layer_norm(x.data_ptr<float>(), weight.data_ptr<float>(), bias.data_ptr<float>(), (int)x.size(0), (int)x.size(1),
                              (int)x.size(2), (float)eps, y.data_ptr<float>(), L.stream), "nws_td_layer_norm");
  return y;
}

Tensor film(const Tensor& x, const Tensor& gamma, const Tensor& beta) {
  check_dev(x, "x");
  check_dev(gamma, "gamma");
  check_dev(beta, "beta");
  check_same_device(x, "x", gamma, "gamma");
  check_same_device(x, "x", beta, "beta");
  TORCH_CHECK(x.sizes() == gamma.sizes() && x.sizes() == beta.sizes(), "FiLM: x ", x.sizes(), ", gamma ", gamma.sizes(), ", beta ",
              beta.sizes(), " must have the same shape (expand and make contiguous first)");
  Launch L(x);
  Tensor y = at::empty_like(x);
  nws_check(nws_film(x.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(), x.numel(), y.data_ptr<float>(), L.stream),
            "nws_film");
  return y;
}

Tensor sine(const Tensor& x) {
  check_dev(x, "x");
  Launch L(x);
  Tensor y = at::empty_like(x);
  nws_check(nws_sin(x.data_ptr<float>(), y.data_ptr<float>(), x.numel(), L.stream), "nws_sin");
  return y;
}

// extract_perceptual_loudness (data/utils/loudness_extraction.py:41-67): audio (B, N) -> (B, 1 + N / hop)
Tensor loudness(const Tensor& audio, const Tensor& dft, int64_t n_fft, int64_t hop, double amin, double top_db, bool normalise) {
  check_dev(audio, "audio");
  check_dev(dft, "dft");
  check_same_device(audio, "audio", dft, "dft");
  TORCH_CHECK(audio.dim() == 2, "loudness: expected (B, N), got ", audio.sizes());
  const int64_t B = audio.size(0), N = audio.size(1);
  const size_t nbytes = nws_loudness_workspace_bytes((int)B, (int)N, (int)n_fft, (int)hop);
  TORCH_CHECK(nbytes > 0, "loudness: unsupported n_fft / hop_length (", n_fft, ", ", hop,
              "): n_fft must be a power of two in [64, 2048], 1 <= hop <= n_fft and 31 hop + n_fft samples must fit 160 KB of LDS");
  TORCH_CHECK((size_t)dft.numel() * sizeof(float) == nws_loudness_dft_bytes((int)n_fft), "loudness: dft does not belong to n_fft = ", n_fft);
  Launch L(audio);
  Tensor ws = at::empty({(int64_t)nbytes}, audio.options().dtype(at::kByte));
  Tensor out = at::empty({B, nws_loudness_frames((int)N, (int)hop)}, audio.options());
  nws_check(nws_loudness(audio.data_ptr<float>(), (int)B, (int)N, (int)n_fft, (int)hop, dft.data_ptr<float>(), (float)amin,
                         (float)top_db, normalise ? 1 : 0, out.data_ptr<float>(), ws.data_ptr(), nbytes, L.stream), "nws_loudness");
  return out;
}

// ---- runtime-size path (csrc/generic.hip): any gin configuration of the reference --------------------------------------
template <class T>
const T* struct_of(const Tensor& desc, const char* name) {
  TORCH_CHECK(desc.device().is_cpu() && desc.scalar_type() == at::kByte && desc.is_contiguous() && (size_t)desc.numel() == sizeof(T),
              name, ": expected a contiguous CPU uint8 tensor of ", sizeof(T), " bytes, got ", desc.sizes());
  return reinterpret_cast<const T*>(desc.data_ptr());
}

Tensor forward_generic(const Tensor& gdesc, const Tensor& f0, const Tensor& control, const Tensor& phase_u, const Tensor& rand_phase,
                       const Tensor& noise, const OptTensor& plan_t, const OptTensor& tables, const OptTensor& spectrum,
                       Tensor& reverb_workspace, Tensor& workspace, double sample_rate) {
  const NwsGenericModel* m = struct_of<NwsGenericModel>(gdesc, "gdesc");
  int64_t B, C, T;
  check_inputs(f0, control, B, C, T);
  check_dev(phase_u, "phase_u");
  check_dev(rand_phase, "rand_phase");
  check_dev(noise, "noise");
  check_same_device(f0, "f0", phase_u, "phase_u");
  check_same_device(f0, "f0", noise, "noise");
  check_same_device(f0, "f0", rand_phase, "osc.rand_phase");
  check_dev(workspace, "workspace", at::kByte);
  check_dev(reverb_workspace, "reverb_workspace", at::kByte);
  check_same_device(f0, "f0", workspace, "workspace");
  TORCH_CHECK(m->hop > 0 && m->n_harmonics > 0, "gdesc: not a NwsGenericModel");
  TORCH_CHECK(phase_u.numel() == m->n_harmonics && rand_phase.numel() == m->n_harmonics, "phase_u / rand_phase: expected ",
              m->n_harmonics, " elements");
  const int64_t N = T * m->hop;
  TORCH_CHECK(noise.numel() == N - 1, "noise: expected ", N - 1, " elements, got ", noise.sizes());
  TORCH_CHECK((size_t)workspace.numel() >= nws_forward_generic_workspace_bytes(m, (int)B, (int)T), "workspace too small");
  NwsReverbPlan plan{};
  const bool fft = plan_t.has_value();
  if (fft) {
    TORCH_CHECK(tables.has_value() && spectrum.has_value(), "forward_generic: plan without tables / spectrum");
    plan = plan_of(*plan_t);
    check_dev(*tables, "reverb_tables");
    check_dev(*spectrum, "reverb_spectrum");
    check_reverb_buffers(plan, *tables, *spectrum);
    TORCH_CHECK(nws_reverb_plan_serves(&plan, (int)N, m->ir_len + 1), "forward_generic: the reverb plan was not made for ", N, " samples");
    TORCH_CHECK((size_t)reverb_workspace.numel() >= nws_reverb_workspace_bytes(&plan, (int)B), "reverb workspace too small");
  }
  Launch L(f0);
  Tensor out = at::empty({B, N}, f0.options());
  nws_check(nws_forward_generic(m, f0.data_ptr<float>(), control.data_ptr<float>(), (int)B, (int)C, (int)T, (float)sample_rate,
                                phase_u.data_ptr<float>(), rand_phase.data_ptr<float>(), noise.data_ptr<float>(),
                                fft ? &plan : nullptr, fft ? tables->data_ptr() : nullptr, fft ? spectrum->data_ptr() : nullptr,
                                fft ? reverb_workspace.data_ptr() : nullptr, fft ? (size_t)reverb_workspace.numel() : 0,
                                out.data_ptr<float>(), workspace.data_ptr(), (size_t)workspace.numel(), L.stream),
            "nws_forward_generic");
  return out;
}

// nn.GRU(C_in -> H, batch_first) over control[:, 0:C_in] of (B, C_total, T): -> (out (B, T, H), hT (B, H))
std::tuple<Tensor, Tensor> g_gru(const Tensor& w_ih, const Tensor& w_hh, const Tensor& b_ih, const Tensor& b_hh, const Tensor& control,
                                 const OptTensor& h0) {
  check_dev(control, "control");
  for (const Tensor* t : {&w_ih, &w_hh, &b_ih, &b_hh}) {
    check_dev(*t, "gru parameter");
    check_same_device(control, "control", *t, "the GRU's parameters");
  }
  TORCH_CHECK(control.dim() == 3 && w_hh.dim() == 2 && w_ih.dim() == 2, "g_gru: control (B, C, T), weight_ih (3H, C_in), weight_hh (3H, H)");
  const int64_t B = control.size(0), Ct = control.size(1), T = control.size(2), H = w_hh.size(1), Cin = w_ih.size(1);
  TORCH_CHECK(w_hh.size(0) == 3 * H && w_ih.size(0) == 3 * H && b_ih.numel() == 3 * H && b_hh.numel() == 3 * H, "g_gru: inconsistent GRU parameter shapes");
  TORCH_CHECK(Ct >= Cin && T >= 1, "g_gru: control has ", Ct, " channels, the GRU takes ", Cin);
  if (h0.has_value()) {
    check_dev(*h0, "h0");
    TORCH_CHECK(h0->numel() == B * H, "h0: expected (", B, ", ", H, ")");
  }
  Launch L(control);
  Tensor out = at::empty({B, T, H}, control.options());
  Tensor hT = at::empty({B, H}, control.options());
  const size_t nb = nws_g_gru_workspace_bytes((int)H);
  Tensor ws = at::empty({(int64_t)nb}, control.options().dtype(at::kByte));
  nws_check(nws_g_gru(w_ih.data_ptr<float>(), w_hh.data_ptr<float>(), b_ih.data_ptr<float>(), b_hh.data_ptr<float>(),
                      control.data_ptr<float>(), (int)B, (int)Ct, (int)Cin, (int)H, (int)T, fptr(h0), out.data_ptr<float>(),
                      hT.data_ptr<float>(), ws.data_ptr(), nb, L.stream), "nws_g_gru");
  return {out, hT};
}

// HarmonicOscillator.forward for any n_harmonics / length: f0_up (B, N) -> (B, K, N)
Tensor g_oscillator(const Tensor& f0_up, const Tensor& phase_u, const Tensor& rand_phase, double sample_rate) {
  check_dev(f0_up, "f0");
  check_dev(phase_u, "phase_u");
  check_dev(rand_phase, "rand_phase");
  check_same_device(f0_up, "f0", phase_u, "phase_u");
  check_same_device(f0_up, "f0", rand_phase, "rand_phase");
  TORCH_CHECK(f0_up.dim() == 2 && f0_up.size(1) > 0, "HarmonicOscillator: expected f0 of shape (B, N), got ", f0_up.sizes());
  const int64_t B = f0_up.size(0), N = f0_up.size(1), K = phase_u.numel();
  TORCH_CHECK(rand_phase.numel() == K && K > 0, "phase_u / rand_phase disagree");
  Launch L(f0_up);
  Tensor phase = at::empty_like(f0_up);
  nws_check(nws_g_phase(nullptr, f0_up.data_ptr<float>(), (int)B, (int)N, 1, (float)sample_rate, nullptr, phase.data_ptr<float>(),
                        L.stream), "nws_g_phase");
  Tensor out = at::empty({B, K, N}, f0_up.options());
  nws_check(nws_g_oscillator(f0_up.data_ptr<float>(), phase.data_ptr<float>(), phase_u.data_ptr<float>(), rand_phase.data_ptr<float>(),
                             (int)K, (int)B, (int)N, (float)sample_rate, out.data_ptr<float>(), L.stream), "nws_g_oscillator");
  return out;
}

// F.upsample(x, T * hop, mode="linear") on the last axis of (..., T)
Tensor g_upsample(const Tensor& x, int64_t hop) {
  check_dev(x, "x");
  TORCH_CHECK(x.dim() >= 1 && x.size(-1) >= 1 && hop >= 1, "upsample: expected (..., T) and hop >= 1");
  const int64_t T = x.size(-1), rows = x.numel() / T;
  Launch L(x);
  auto shape = x.sizes().vec();
  shape.back() = T * hop;
  Tensor y = at::empty(shape, x.options());
  nws_check(nws_g_upsample(x.data_ptr<float>(), rows, (int)T, (int)hop, y.data_ptr<float>(), L.stream), "nws_g_upsample");
  return y;
}

Tensor g_conv1x1(const Tensor& x, const Tensor& w, const OptTensor& bias) {
  check_dev(x, "x");
  check_dev(w, "weight");
  check_same_device(x, "x", w, "weight");
  TORCH_CHECK(x.dim() == 3 && w.dim() >= 2 && w.size(1) == x.size(1) && w.numel() == w.size(0) * w.size(1),
              "conv1x1: x (B, Cin, N), weight (Cout, Cin[, 1]); got ", x.sizes(), " / ", w.sizes());
  if (bias.has_value()) {
    check_dev(*bias, "bias");
    TORCH_CHECK(bias->numel() == w.size(0), "conv1x1: bias ", bias->sizes());
  }
  Launch L(x);
  Tensor y = at::empty({x.size(0), w.size(0), x.size(2)}, x.options());
  nws_check(nws_g_conv1x1(x.data_ptr<float>(), w.data_ptr<float>(), fptr(bias), (int)x.size(0), (int)x.size(1), (int)w.size(0),
                          (int)x.size(2), y.data_ptr<float>(), L.stream), "nws_g_conv1x1");
  return y;
}

Tensor g_shaper_apply(const Tensor& sdesc, const Tensor& x) {
  const NwsShaperDesc* d = struct_of<NwsShaperDesc>(sdesc, "sdesc");
  check_dev(x, "x");
  TORCH_CHECK(x.dim() == 3 && x.size(1) == d->n_shapers, "expected (B, ", d->n_shapers, ", N), got ", x.sizes());
  Launch L(x);
  Tensor y = at::empty_like(x);
  nws_check(nws_g_shaper_apply(d, x.data_ptr<float>(), x.size(0) * x.size(1), x.size(2), y.data_ptr<float>(), L.stream),
            "nws_g_shaper_apply");
  return y;
}

Tensor g_shaper_table(const Tensor& sdesc, const Tensor& like, int64_t size, double tmin, double tmax) {
  const NwsShaperDesc* d = struct_of<NwsShaperDesc>(sdesc, "sdesc");
  check_dev(like, "shaping_fn.input_scale");
  TORCH_CHECK(size >= 2 && tmax > tmin, "FastNEWT: need table_size >= 2 and table_max > table_min");
  Launch L(like);
  Tensor table = at::empty({(int64_t)d->n_shapers, size}, like.options());
  nws_check(nws_g_shaper_table(d, (int)size, (float)tmin, (float)tmax, table.data_ptr<float>(), L.stream), "nws_g_shaper_table");
  return table;
}

// NEWT.forward / FastNEWT.forward for any sizes: exciter (B, S, N), film (B, 4S, T) -> (B, O, N)
Tensor g_newt_apply(const Tensor& sdesc, const Tensor& exciter, const Tensor& film, const Tensor& mix_w, const Tensor& mix_b) {
  const NwsShaperDesc* d = struct_of<NwsShaperDesc>(sdesc, "sdesc");
  check_dev(exciter, "exciter");
  check_dev(film, "film_params");
  check_dev(mix_w, "mixer weight");
  check_dev(mix_b, "mixer bias");
  check_same_device(exciter, "exciter", film, "control embedding");
  check_same_device(exciter, "exciter", mix_w, "newt.mixer");
  const int64_t S = d->n_shapers;
  TORCH_CHECK(exciter.dim() == 3 && exciter.size(1) == S, "NEWT: expected an exciter of shape (B, ", S, ", N), got ", exciter.sizes());
  TORCH_CHECK(film.dim() == 3 && film.size(1) == 4 * S && film.size(0) == exciter.size(0), "NEWT: expected FiLM parameters of shape (B, ",
              4 * S, ", T), got ", film.sizes());
  const int64_t B = exciter.size(0), N = exciter.size(2), T = film.size(2);
  TORCH_CHECK(T > 0 && N % T == 0, "NEWT: ", N, " samples are not a whole multiple of ", T, " control frames");
  TORCH_CHECK(mix_w.numel() % S == 0 && mix_b.numel() == mix_w.numel() / S, "NEWT: mixer weight ", mix_w.sizes(), " / bias ", mix_b.sizes());
  const int64_t O = mix_w.numel() / S;
  Launch L(exciter);
  Tensor shaped = at::empty_like(exciter);
  nws_check(nws_g_film_shaper(d, exciter.data_ptr<float>(), film.data_ptr<float>(), (int)B, (int)T, (int)(N / T),
                              shaped.data_ptr<float>(), L.stream), "nws_g_film_shaper");
  Tensor out = at::empty({B, O, N}, exciter.options());
  nws_check(nws_g_conv1x1(shaped.data_ptr<float>(), mix_w.data_ptr<float>(), mix_b.data_ptr<float>(), (int)B, (int)S, (int)O, (int)N,
                          out.data_ptr<float>(), L.stream), "nws_g_conv1x1");
  return out;
}

// FIRNoiseSynth.forward for any even ir_length >= hop: H (B, L/2+1, T) -> (B, hop * T)
Tensor g_fir_noise(const Tensor& H, const Tensor& window, const Tensor& noise, int64_t hop) {
  check_dev(H, "H");
  check_dev(window, "window");
  check_dev(noise, "noise");
  check_same_device(H, "H", window, "noise_synth.window");
  check_same_device(H, "H", noise, "noise");
  const int64_t Lf = window.numel();
  TORCH_CHECK(H.dim() == 3 && H.size(1) == Lf / 2 + 1, "FIRNoiseSynth: expected H of shape (B, ", Lf / 2 + 1, ", T), got ", H.sizes());
  const int64_t B = H.size(0), T = H.size(2);
  TORCH_CHECK(noise.numel() == hop * T - 1, "noise: expected ", hop * T - 1, " samples, got ", noise.sizes());
  Launch L(H);
  Tensor fir = at::empty({B, T, Lf}, H.options());
  nws_check(nws_g_fir_design(H.data_ptr<float>(), window.data_ptr<float>(), (int)Lf, (int)B, (int)T, fir.data_ptr<float>(), L.stream),
            "nws_g_fir_design");
  Tensor out = at::empty({B, hop * T}, H.options());
  nws_check(nws_g_fir_noise(fir.data_ptr<float>(), noise.data_ptr<float>(), (int)Lf, (int)hop, (int)B, (int)T, nullptr, 0,
                            out.data_ptr<float>(), L.stream), "nws_g_fir_noise");
  return out;
}

Tensor g_reverb_direct(const Tensor& x, const Tensor& ir) {
  check_dev(x, "x");
  check_dev(ir, "reverb.ir");
  check_same_device(x, "x", ir, "reverb.ir");
  TORCH_CHECK(x.dim() == 2, "Reverb: expected (B, N), got ", x.sizes());
  Launch L(x);
  Tensor y = at::empty_like(x);
  nws_check(nws_g_reverb_direct(x.data_ptr<float>(), ir.data_ptr<float>(), (int)ir.numel(), (int)x.size(0), (int)x.size(1),
                                y.data_ptr<float>(), L.stream), "nws_g_reverb_direct");
  return y;
}

// ---- stateful streaming step (csrc/stream.hip): K new frames of B streams -> out (B, M); all state in `state` ----------
void stream_step(const Tensor& wdesc, const Tensor& fir_design, const OptTensor& plan_t, const OptTensor& tables,
                 const OptTensor& spectrum, Tensor& state, int64_t max_frames, const Tensor& f0, const Tensor& control, bool first,
                 bool final, int64_t frames_seen, int64_t nz_prev_start, double sample_rate, const Tensor& phase_u,
                 const Tensor& rand_phase, const OptTensor& noise_new, const OptTensor& noise_all, const Tensor& ir, Tensor& out,
                 const OptTensor& pre_out) {
  const NwsWeights* w = weights_of(wdesc);
  check_dev(f0, "f0");
  check_dev(control, "control");
  check_dev(fir_design, "fir_design");
  check_dev(state, "state", at::kByte);
  check_dev(phase_u, "phase_u");
  check_dev(rand_phase, "rand_phase");
  check_dev(ir, "reverb.ir");
  check_dev(out, "out");
  check_same_device(f0, "f0", control, "control");
  check_same_device(f0, "f0", state, "state");
  check_same_device(f0, "f0", out, "out");
  check_same_device(f0, "f0", ir, "reverb.ir");
  TORCH_CHECK(f0.dim() == 2 && control.dim() == 3 && control.size(0) == f0.size(0) && control.size(2) == f0.size(1) && control.size(1) >= 2,
              "stream_step: f0 (B, K), control (B, C>=2, K); got ", f0.sizes(), " / ", control.sizes());
  const int64_t B = f0.size(0), K = f0.size(1), C = control.size(1);
  TORCH_CHECK(K >= 1 && K <= max_frames, "stream_step: chunk of ", K, " frames, the stream was sized for ", max_frames);
  TORCH_CHECK(phase_u.numel() == NWS_N_HARMONICS && rand_phase.numel() == NWS_N_HARMONICS, "phase_u / rand_phase: 101 elements each");
  TORCH_CHECK(noise_new.has_value() != noise_all.has_value(), "stream_step: give exactly one of noise_new and noise_all");
  const int M = nws_stream_out_samples((int)K, first, final);
  TORCH_CHECK(out.numel() == B * M, "out: expected (", B, ", ", M, "), got ", out.sizes());
  if (pre_out.has_value()) {
    check_dev(*pre_out, "pre_out");
    TORCH_CHECK(pre_out->numel() == B * M, "pre_out: expected (", B, ", ", M, ")");
  }
  if (noise_new.has_value()) {
    check_dev(*noise_new, "noise_new");
    TORCH_CHECK(noise_new->numel() >= nws_stream_noise_draws((int)K, first, frames_seen), "noise_new: expected ",
                nws_stream_noise_draws((int)K, first, frames_seen), " fresh samples");
  } else {
    check_dev(*noise_all, "noise_all");
  }
  NwsReverbPlan plan{};
  const bool fft = plan_t.has_value();
  if (fft) {
    TORCH_CHECK(tables.has_value() && spectrum.has_value(), "stream_step: plan without tables / spectrum");
    plan = plan_of(*plan_t);
    check_dev(*tables, "reverb_tables");
    check_dev(*spectrum, "reverb_spectrum");
    check_reverb_buffers(plan, *tables, *spectrum);
  }
  TORCH_CHECK((size_t)state.numel() >= nws_stream_state_bytes((int)B, (int)max_frames, (int)ir.numel(), fft ? &plan : nullptr),
              "stream_step: state blob too small");
  Launch L(f0);
  nws_check(nws_stream_step(w, fir_design.data_ptr<float>(), fft ? &plan : nullptr, fft ? tables->data_ptr() : nullptr,
                            fft ? spectrum->data_ptr() : nullptr, state.data_ptr(), (size_t)state.numel(), (int)B, (int)max_frames,
                            f0.data_ptr<float>(), control.data_ptr<float>(), (int)C, (int)K, first ? 1 : 0, final ? 1 : 0,
                            (long long)frames_seen, (long long)nz_prev_start, (float)sample_rate, phase_u.data_ptr<float>(),
                            rand_phase.data_ptr<float>(), fptr(noise_new), fptr(noise_all),
                            noise_all.has_value() ? (int)noise_all->numel() : 0, ir.data_ptr<float>(), (int)ir.numel(),
                            out.data_ptr<float>(), pre_out.has_value() ? pre_out->data_ptr<float>() : nullptr, L.stream),
            "nws_stream_step");
}

int64_t abi_version() { return nws_abi_version(); }

}  // namespace

TORCH_LIBRARY(newt_hip, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("forward(Tensor wdesc, Tensor f0, Tensor control, Tensor phase_u, Tensor rand_phase, Tensor noise, Tensor fir_design, "
        "Tensor plan, Tensor reverb_tables, Tensor reverb_spectrum, Tensor(a!) workspace, float sample_rate) -> Tensor", &forward);
  m.def("forward_control(Tensor wdesc, Tensor f0, Tensor control, Tensor(a!) workspace, bool batched_gru) -> ()", &forward_control);
  m.def("forward_audio(Tensor wdesc, Tensor f0, Tensor phase_u, Tensor rand_phase, Tensor noise, Tensor fir_design, Tensor plan, "
        "Tensor reverb_tables, Tensor reverb_spectrum, Tensor(a!) workspace, float sample_rate, Tensor? out, int wait_event, "
        "int record_event) -> Tensor", &forward_audio);
  m.def("forward_audio_pre(Tensor wdesc, Tensor f0, Tensor phase_u, Tensor rand_phase, Tensor noise, Tensor fir_design, Tensor plan, "
        "Tensor reverb_tables, Tensor reverb_spectrum, Tensor(a!) workspace, float sample_rate) -> ()", &forward_audio_pre);
  m.def("forward_audio_blocks(Tensor wdesc, Tensor f0, Tensor phase_u, Tensor rand_phase, Tensor noise, Tensor fir_design, Tensor plan, "
        "Tensor reverb_tables, Tensor reverb_spectrum, Tensor(a!) workspace, float sample_rate, Tensor(b!) out, int[] row0, int[] nrows, "
        "int[] events) -> ()", &forward_audio_blocks);
  m.def("forward_reverb_rows(Tensor fir_design, Tensor plan, Tensor reverb_tables, Tensor reverb_spectrum, Tensor(a!) workspace, int T, "
        "int row0, int nrows, Tensor(b!) out) -> ()", &forward_reverb_rows);
  m.def("phase_carry(Tensor? f0, Tensor? f0_up) -> Tensor", &phase_carry);
  m.def("exciter_newt(Tensor wdesc, Tensor? f0, Tensor? f0_up, Tensor carry, Tensor phase_u, Tensor rand_phase, Tensor? film, "
        "float sample_rate, bool want_exciter, bool want_newt) -> (Tensor, Tensor)", &exciter_newt);
  m.def("oscillator(Tensor f0_up, Tensor phase_u, Tensor rand_phase, float sample_rate) -> Tensor", &oscillator);
  m.def("control_gru(Tensor wdesc, Tensor control, Tensor? h0, bool batched) -> (Tensor, Tensor)", &control_gru);
  m.def("frame_mlps(Tensor wdesc, Tensor gru_out, Tensor fir_design, bool want_emb, bool want_H) -> (Tensor, Tensor, Tensor, Tensor)",
        &frame_mlps);
  m.def("fir_noise(Tensor fir, Tensor noise, Tensor? add_in, int origin) -> Tensor", &fir_noise);
  m.def("fir_from_h(Tensor H, Tensor fir_design) -> Tensor", &fir_from_h);
  m.def("reverb(Tensor plan, Tensor tables, Tensor spectrum, Tensor x) -> Tensor", &reverb);
  m.def("reverb_linear_chunk(Tensor plan, Tensor tables, Tensor spectrum, Tensor x, Tensor tail_in) -> (Tensor, Tensor)",
        &reverb_linear_chunk);
  m.def("shaper_apply(Tensor wdesc, Tensor x) -> Tensor", &shaper_apply);
  m.def("shaper_table(Tensor wdesc, Tensor like, int size, float tmin, float tmax) -> Tensor", &shaper_table);
  m.def("newt_apply(Tensor wdesc, Tensor exciter, Tensor film) -> Tensor", &newt_apply);
  m.def("td_mlp(Tensor x, Tensor[] weights, Tensor[] biases, Tensor[] ln_w, Tensor[] ln_b, float eps, float slope) -> Tensor", &td_mlp);
  m.def("td_layer_norm(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor", &td_layer_norm);
  m.def("film(Tensor x, Tensor gamma, Tensor beta) -> Tensor", &film);
  m.def("sine(Tensor x) -> Tensor", &sine);
  m.def("forward_generic(Tensor gdesc, Tensor f0, Tensor control, Tensor phase_u, Tensor rand_phase, Tensor noise, Tensor? plan, "
        "Tensor? reverb_tables, Tensor? reverb_spectrum, Tensor(a!) reverb_workspace, Tensor(b!) workspace, float sample_rate) -> Tensor",
        &forward_generic);
  m.def("g_gru(Tensor w_ih, Tensor w_hh, Tensor b_ih, Tensor b_hh, Tensor control, Tensor? h0) -> (Tensor, Tensor)", &g_gru);
  m.def("g_oscillator(Tensor f0_up, Tensor phase_u, Tensor rand_phase, float sample_rate) -> Tensor", &g_oscillator);
  m.def("g_conv1x1(Tensor x, Tensor weight, Tensor? bias) -> Tensor", &g_conv1x1);
  m.def("g_upsample(Tensor x, int hop) -> Tensor", &g_upsample);
  m.def("g_shaper_apply(Tensor sdesc, Tensor x) -> Tensor", &g_shaper_apply);
  m.def("g_shaper_table(Tensor sdesc, Tensor like, int size, float tmin, float tmax) -> Tensor", &g_shaper_table);
  m.def("g_newt_apply(Tensor sdesc, Tensor exciter, Tensor film, Tensor mix_w, Tensor mix_b) -> Tensor", &g_newt_apply);
  m.def("g_fir_noise(Tensor H, Tensor window, Tensor noise, int hop) -> Tensor", &g_fir_noise);
  m.def("g_reverb_direct(Tensor x, Tensor ir) -> Tensor", &g_reverb_direct);
  m.def("stream_step(Tensor wdesc, Tensor fir_design, Tensor? plan, Tensor? reverb_tables, Tensor? reverb_spectrum, Tensor(a!) state, "
        "int max_frames, Tensor f0, Tensor control, bool first, bool final, int frames_seen, int nz_prev_start, float sample_rate, "
        "Tensor phase_u, Tensor rand_phase, Tensor? noise_new, Tensor? noise_all, Tensor ir, Tensor(b!) out, Tensor(c!)? pre_out) -> ()",
        &stream_step);
  m.def("loudness(Tensor audio, Tensor dft, int n_fft, int hop, float amin, float top_db, bool normalise) -> Tensor", &loudness);
}
