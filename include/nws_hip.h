/*
 * nws_hip.h — C-ABI of the MI355X (gfx950) NEWT synthesis engine.
 *
 * The reference (ben-hayes/neural-waveshaping-synthesis) has no FFI: its hot path sits behind
 * a Python nn.Module.  This header is the drop-in boundary *below* that
 * surface: every entry point replaces one block of ATen calls made by the reference's
 * forward().  Citations are relative to /root/reference/neural_waveshaping_synthesis/.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers to contiguous fp32 unless
 *     stated otherwise; `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - every launcher returns 0 on success, a positive hipError_t if the HIP runtime failed, or
 *     one of the negative NWS_ERR_* codes below.  Nothing is synchronised: work is enqueued on
 *     `stream` exactly like the ATen ops it replaces.
 *   - kernels are specialised for the architecture of gin/models/newt.gin (the only one the
 *     reference ships): 101 harmonics, 64 waveshapers, hidden/embedding 128, shaper MLP width 8
 *     depth 4, control hop 128, FIR length 256.  Other sizes return NWS_ERR_UNSUPPORTED.
 *   - B = batch, T = control frames, N = 128*T samples.
 */
#ifndef NWS_HIP_H
#define NWS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NWS_ABI_VERSION 6

#define NWS_N_HARMONICS 101
#define NWS_N_SHAPERS 64
#define NWS_HIDDEN 128
#define NWS_HOP 128
#define NWS_FIR_LEN 256
#define NWS_N_BANDS 129      /* FIR_LEN/2 + 1 */
#define NWS_FIR_HALF 128     /* taps handed from the frame MLPs to the noise kernel per frame: h[128 .. 255] (see nws_frame_mlps) */
#define NWS_SHAPER_WIDTH 8
#define NWS_FILM_CH 256      /* 4 * N_SHAPERS */
#define NWS_FILM_REC_BYTES 1536 /* per-frame FiLM fragment record: 3 parameter types x 64 shapers x {bf16 t0, t1, t2, 0} (csrc/exciter_newt.hip) */

enum {
  NWS_OK = 0,
  NWS_ERR_UNSUPPORTED = -1, /* sizes outside the compiled specialisation */
  NWS_ERR_BAD_ARG = -2,     /* NULL / non-positive / misaligned argument */
  NWS_ERR_WORKSPACE = -3    /* workspace too small */
};

/* Device pointers to the parameters, in the reference's own state_dict layouts. */
typedef struct NwsWeights {
  /* embedding = ControlModule (models/neural_waveshaping.py:17-26) */
  const float* gru_w_ih;  /* (384, 2)   rows [r; z; n] */
  const float* gru_w_hh;  /* (384, 128) */
  const float* gru_b_ih;  /* (384) */
  const float* gru_b_hh;  /* (384) */
  const float* proj_w;    /* (128, 128) Conv1d k=1 */
  const float* proj_b;    /* (128) */
  /* harmonic_mixer = Conv1d(101, 64, 1) (models/neural_waveshaping.py:54) */
  const float* mixer_w;   /* (64, 101) */
  const float* mixer_b;   /* (64) */
  const void* mixer_frags; /* optional 28672 B from nws_mixer_frags(): [mixer_b | mixer_w] as two fp16 terms in MFMA
                              fragment order; NULL -> every workgroup splits the weights itself (slower prologue) */
  /* newt.mlp = TimeDistributedMLP(128,128,256,depth=4) (models/modules/shaping.py:53-55) */
  const float* newt_mlp_w[4]; /* (128,128) x3, (256,128) */
  const float* newt_mlp_b[4];
  const float* newt_ln_g[3];  /* LayerNorm weight (128) */
  const float* newt_ln_b[3];
  /* h_generator = TimeDistributedMLP(128,128,129,depth=4) (models/neural_waveshaping.py:58-59) */
  const float* hgen_w[4];     /* (128,128) x3, (129,128) */
  const float* hgen_b[4];
  const float* hgen_ln_g[3];
  const float* hgen_ln_b[3];
  /* optional NWS_MLP_FRAGS_BYTES from nws_mlp_frags(): proj / newt.mlp / h_generator / FIR-design weights as two fp16 terms in MFMA
   * fragment order -> frame MLPs on the fp16 matrix pipe; NULL -> exact-fp32 MFMA kernel */
  const void* mlp_frags;
  /* newt.shaping_fn = TrainableNonlinearity(64, 8, depth=4) (models/modules/shaping.py:15-37) */
  const float* shaper_in_scale; /* (64) */
  const float* shaper_w0; /* (512)    net.0.weight (512,1,1) */
  const float* shaper_b0; /* (512) */
  const float* shaper_w2; /* (512, 8) net.2.weight */
  const float* shaper_b2; /* (512) */
  const float* shaper_w4; /* (512, 8) net.4.weight */
  const float* shaper_b4; /* (512) */
  const float* shaper_w6; /* (64, 8)  net.6.weight */
  const float* shaper_b6; /* (64) */
  const float* shaper_turns; /* optional (64, NWS_SHAPER_TURNS_ROW) from nws_shaper_turns(): all four layers of each shaper
                                times 1/(2 pi), hidden layers transposed, read by scalar loads; NULL -> weights staged in
                                LDS per workgroup (slower exact mode) */
  /* FastNEWT.lookup_table (models/modules/shaping.py:103-105); NULL selects the exact shapers */
  const float* lut;       /* (64, lut_size) */
  const float* lut_pairs; /* optional (64, lut_size, 2) from nws_lut_pairs(): {T[i], T[i+1]-T[i]}; NULL -> two gathers */
  int32_t lut_size;       /* 4096 */
  float lut_min;          /* -3 */
  float lut_max;          /* +3 */
  /* newt.mixer = Conv1d(64, 1, 1) (models/modules/shaping.py:63-65) */
  const float* newt_out_w; /* (64) */
  const float* newt_out_b; /* (1) */
  /* noise_synth.window (models/modules/generators.py:20), periodic Hann(256) */
  const float* noise_window; /* (256) */
  /* options of the fused oscillator + waveshaper kernel's FastNEWT path (0 = default: FiLM interpolation on the matrix
   * pipe, sines as two fp16 terms = 22-bit products) */
  int32_t exciter_opts;
  /* optional (64) from nws_exciter_bound(): X[s] >= |harmonic_mixer output of shaper s| for ANY phases (sum_k |W[s][k]| + |b[s]|,
   * rounded up).  With it the fused kernel proves per workgroup, from the FiLM rows it stages, which groups of eight shapers keep
   * their table index inside the table whatever the oscillator does, and runs those lookups without floor / clamp (bit-identical
   * there; everything else takes the reference's clamped form, shaping.py:136-151).  NULL -> every lookup clamped. */
  const float* exciter_bound;
} NwsWeights;
#define NWS_EXCITER_VALU_FILM 1 /* round-1 form: FiLM parameters interpolated on the VALU (kept for A/B timing) */
#define NWS_EXCITER_ONE_TERM 2  /* sines as ONE fp16 term: 2 instead of 3 MFMAs per product and no residual split (-15 %
                                   kernel time); mixer inputs carry 11 bits: 4e-6 .. 1.1e-5 RMS end to end on the shipped
                                   checkpoints instead of ~3e-7 */
#define NWS_EXCITER_HYBRID 4    /* two fp16 terms for the mixer bias + harmonics 1..15 (K-step 0: 61-85 % of the mixer's
                                   weight energy in the shipped checkpoints), one term for harmonics 16..101 */
#define NWS_EXCITER_HYBRID_W 8  /* NWS_EXCITER_HYBRID plus: the mixer WEIGHTS of harmonics 16..101 as one fp16 term as well
                                   (one MFMA per product there instead of two) */
#define NWS_EXCITER_BANK_NOFRACT 16 /* exact sin-MLP shapers (shaping.py:36-37): the hidden and output layers' pre-activations are
                                   provably inside v_sin_f32's own +-256-turn domain (sum |W| + |b| per row, checked by the caller
                                   from the weights), so their sines skip the v_fract: 17 of an evaluation's 25 (the 8 first-layer sines keep it: their argument is unbounded) */

int nws_abi_version(void);
/* sizeof of the C structs above as this library was compiled (0 NwsWeights, 1 NwsReverbPlan, 2 NwsForwardAux, 3 NwsShaperDesc, 4 NwsGenericModel): lets a
 * foreign-language binding verify its own struct declarations at load time */
size_t nws_sizeof(int which);
const char* nws_error_string(int code);

/* ---- hardware self-test: MFMA fragment layout the kernels rely on (returns #mismatches in *bad) ---- */
int nws_selftest_mfma(int32_t* bad_out /* device int32[1] */, void* stream);

/* ---- accurate sinf used by the oscillator / shapers, exposed for testing: y[i] = sin(x[i]) ---- */
int nws_sin(const float* x, float* y, int64_t n, void* stream);

/*
 * Exciter phase carries.  Replaces the double-accumulated torch.cumsum of
 * models/modules/generators.py:59 at 32-sample granularity:
 *   carry[b][c] = sum_{n < 32c} f0_up[b][n]   (float64, exclusive)
 * f0 (B,T) Hz frames are linearly upsampled x128 on the fly exactly like
 * F.upsample(mode="linear") (models/neural_waveshaping.py:75); if f0_up != NULL the (B,N)
 * already-upsampled F0 is used instead (public render_exciter(), :64-67).
 */
int nws_phase_carry(const float* f0, const float* f0_up, int B, int T, double* carry /* (B, N/32) */, void* stream);

/*
 * Fused harmonic exciter + waveshaper bank:
 *   HarmonicOscillator.forward (models/modules/generators.py:58-66)
 *   harmonic_mixer Conv1d 101->64 (models/neural_waveshaping.py:66)
 *   NEWT.forward / FastNEWT.shaping_fn: FiLM -> shaper (LUT or sin-MLP) -> FiLM -> Conv1d 64->1
 *   (models/modules/shaping.py:67-79, :136-151)
 * film: (B, T, 256) frame-major FiLM parameters [g_idx | b_idx | g_norm | b_norm] (output of
 *       nws_frame_mlps), linearly upsampled on the fly (shaping.py:69).
 * phase_u: (101) the U[0,1) draws of generators.py:55; rand_phase: (101) the osc.rand_phase buffer (= tau);
 *          the kernel forms shift = fl(fl(u * rand_phase) - pi) exactly like _create_phase_shift.
 * exciter_out: optional (B, 64, N) materialised exciter (render_exciter()); newt_out: optional (B, N).
 */
int nws_exciter_newt(const NwsWeights* w, const float* f0, const float* f0_up, const double* carry,
                     const float* phase_u, const float* rand_phase, const float* film, int B, int T,
                     float sample_rate, float* exciter_out, float* newt_out, void* stream);
/* same with newt_out = add_in + NEWT sum; add_in (B, N) or NULL (the noise branch of neural_waveshaping.py:81 when the
 * FIR-noise kernel ran first); add_in may not alias newt_out */
int nws_exciter_newt_add(const NwsWeights* w, const float* f0, const float* f0_up, const double* carry,
                         const float* phase_u, const float* rand_phase, const float* film, const float* add_in, int B,
                         int T, float sample_rate, float* exciter_out, float* newt_out, void* stream);

/*
 * Control encoder: GRU(2->128, h0=0) over T frames of control[:, 0:2]
 * (models/neural_waveshaping.py:69-72, :24-25).  control: (B, C, T) with C >= 2.
 * gru_out: (B, T, 128).
 */
int nws_control_gru(const NwsWeights* w, const float* control, int B, int C, int T, float* gru_out, void* stream);
/* same with an explicit initial state h0 (B,128) [NULL = zeros] and the final state written to hT (B,128) [NULL = dropped]:
 * stateful streaming; the reference's forward is the h0 = 0 case */
int nws_control_gru_state(const NwsWeights* w, const float* control, int B, int C, int T, const float* h0, float* gru_out,
                          float* hT, void* stream);
/* nws_control_gru + nws_phase_carry in ONE launch (grid 2B: B workgroups run the recurrences, B more compute the carries
 * beside them): the control-rate half of a forward; f0 (B,T) Hz, carry_out (B, 4T) doubles */
int nws_control_gru_carry(const NwsWeights* w, const float* control, const float* f0, int B, int C, int T, float* gru_out,
                          double* carry_out, void* stream);
/* the same recurrence, 16 utterances per workgroup on the matrix cores (fp16 two-term split, fp32 accumulate): ~2x the
 * latency of the per-utterance kernel above but ~1/10 of its VALU work per utterance and 1/16 of its workgroups -- the
 * form to run beside throughput kernels of other streams (nws_forward_control, batched_gru = 1).  Any B >= 1. */
int nws_control_gru_batched(const NwsWeights* w, const float* control, int B, int C, int T, const float* h0, float* gru_out,
                            float* hT, void* stream);

/*
 * Frame-rate MLPs on the GRU output (one kernel):
 *   emb  = proj(gru_out)                         models/neural_waveshaping.py:26
 *   film = newt.mlp(emb)                         models/modules/shaping.py:68, dynamic.py:20-40
 *   H    = h_generator(emb)                      models/neural_waveshaping.py:82
 *   fir  = window * roll(irfft(H), 128)          models/modules/generators.py:22-27 (zero-phase FIR design)
 * emb_out (B,128,T) [optional], film_out (B,T,256), H_out (B,T,129) [optional], fir_out (B,T,128).
 * fir_design: (256, 132) constant matrix from nws_fir_design_matrix().
 * fir_out holds the UPPER HALF of every frame's taps, u[d] = h[128 + d], d = 0..127: H is real, so irfft(H) is even and with
 * a window symmetric about tap 128 whose tap 0 is zero (the reference's periodic Hann, generators.py:20) the rolled,
 * windowed response satisfies h[128 - d] = h[128 + d], h[0] = 0.  Half the bytes between the two kernels; the noise
 * kernels mirror the row while staging it.  The caller checks the window (engine.py: any other window takes the
 * runtime-size path, whose nws_g_fir_design emits full rows).
 */
int nws_frame_mlps(const NwsWeights* w, const float* gru_out, const float* fir_design, int B, int T,
                   float* emb_out, float* film_out, float* H_out, float* fir_out, void* stream);

/* pre-split weight fragments for the fp16 two-term frame-MLP kernel (valid while |layer inputs| stay inside fp16 range:
 * the caller checks the weight-norm bounds, see engine.py) */
#define NWS_MLP_FRAGS_BYTES 1474560
int nws_mlp_frags(const NwsWeights* w, const float* fir_design /* (256,132) */, void* frags_out, void* stream);

/* D[n][k]: fir[n] = sum_k D[n][k] H[k]  (irfft + roll(128) + window folded), (256, 132) fp32, cols 129..131 = 0. */
int nws_fir_design_matrix(const float* window /* (256) */, float* D_out, void* stream);

/*
 * Time-varying FIR noise (models/modules/generators.py:30-35): rectangular-window STFT of the
 * shared noise vector (N-1 samples, reflect-padded by 128), per-frame 256-point CIRCULAR
 * convolution with fir[b][t], overlap-add / overlap count.   out = add_in + noise_branch
 * (the cat+sum of models/neural_waveshaping.py:85-86); add_in may be NULL.
 */
int nws_fir_noise(const float* fir /* (B,T,128): upper half-taps, see nws_frame_mlps */, const float* noise /* (N-1) */, const float* add_in /* (B,N) */,
                  int B, int T, float* out /* (B,N) */, void* stream);
/* general form: STFT frame t covers noise[128 t - origin, 128 t - origin + 256), reflected about 0 and noise_len-1 like
 * torch.stft's reflect padding (nws_fir_noise == origin 128, noise_len N-1); streaming windows use origin 0 */
int nws_fir_noise_window(const float* fir, const float* noise, int noise_len, int origin, const float* add_in, int B, int T,
                         float* out, void* stream);

/* ---- learned reverb (models/modules/shaping.py:161-173): y = x + circconv_Lc(x, [0, ir])[:N], Lc = max(N, ir_len+1) ----
 * A plan exists for EVERY even circular length (the reference takes any N; for an odd Lc its rfft / irfft pair is not a
 * circular convolution at all - runtime-size path, nws_g_reverb_direct).  Two forms:
 *   direct (Lc == 0): the L-point four-step transform IS the circular convolution, L = max(N, ir_len+1) = N1 * N2 with
 *     N1 = 125 (radix-5 column pass) or N1 <= 128 (DFT-matrix column pass on the matrix cores);
 *   overlap-save (Lc > 0): every other length.  The signal is read as Lc-periodic (zeros between N and Lc); block j of
 *     L = 125 * 2^k points starts `hist` = ir_len samples before output j * (L - hist), its last L - hist points are
 *     final samples of the circular convolution - the wrap-around is in the loads, nothing is folded afterwards.
 *     nblk = ceil(N / (L - hist)) transforms per utterance pair instead of one. */
typedef struct NwsReverbPlan {
  int32_t L;     /* transform length */
  int32_t N1;    /* column-DFT size  (L = N1 * N2) */
  int32_t N2;    /* row-FFT size, power of two, 32 .. 4096 */
  int32_t Lc;    /* 0: direct; else the reference's circular length, served by overlap-save blocks of L points */
  int32_t hist;  /* overlap-save: samples of history in front of every block (= ir_len); 0 when direct */
  int32_t nblk;  /* transforms per utterance pair (1 when direct) */
  int32_t reserved[2];
} NwsReverbPlan;

int nws_reverb_plan(int N, int ir_len_plus1, NwsReverbPlan* plan /* host */);
/* 1 when `plan` renders Reverb.forward for N samples (ir_len_plus1 > 0: and for that impulse-response length), else 0 */
int nws_reverb_plan_serves(const NwsReverbPlan* plan, int N, int ir_len_plus1);
/* bytes of the constant tables (DFT matrices + twiddles) and of the IR spectrum for a plan */
size_t nws_reverb_table_bytes(const NwsReverbPlan* plan);
size_t nws_reverb_spectrum_bytes(const NwsReverbPlan* plan);
size_t nws_reverb_workspace_bytes(const NwsReverbPlan* plan, int B);
int nws_reverb_build_tables(const NwsReverbPlan* plan, void* tables, void* stream);
/* spectrum of ir_ = [0, ir] zero-padded to L, in the engine's own (k1,k2) order; the buffer (nws_reverb_spectrum_bytes =
 * 3 L floats) holds Sre | Sim | ir_ itself, which nws_reverb uses for buffers of <= 1024 samples (time-domain form) */
int nws_reverb_ir_spectrum(const NwsReverbPlan* plan, const void* tables, const float* ir, int ir_len,
                           void* spectrum, void* workspace, size_t workspace_bytes, void* stream);
int nws_reverb(const NwsReverbPlan* plan, const void* tables, const void* spectrum, const float* x /* (B,N) */,
               int B, int N, float* y /* (B,N) */, void* workspace, size_t workspace_bytes, void* stream);

/* streaming (LINEAR, non-wrapping) variant, one chunk of M samples: y = x + wet[0:M] + tail_in[0:M]; tail_out = shifted
 * tail_in + rest of wet.  plan->L >= M + tail_len (tail_len = len(ir)+1 = 32000); workspace (2*ceil(B/2)*L + B*L) floats. */
int nws_reverb_linear_chunk(const NwsReverbPlan* plan, const void* tables, const void* spectrum, const float* x, int B, int M,
                            const float* tail_in, float* tail_out, int tail_len, float* y, void* workspace,
                            size_t workspace_bytes, void* stream);

/*
 * ---- Stateful streaming (csrc/stream.hip): the reference's buffer benchmark (scripts/time_buffer_sizes.py:50-72)
 * restarts the GRU, the oscillator phase and the reverb with every buffer; here B parallel streams carry them across chunks.
 * The concatenated output equals the one-shot forward up to the reverb input for ANY chunking, and the learned reverb
 * (models/modules/shaping.py:161-173) is applied as a linear convolution of the stream.  All state lives in one caller-owned
 * device blob behind fixed pointers: a steady-state hop is eight launches that can be captured into a hipGraph.
 * `plan` (the L = 64 000 plan of nws_reverb_plan(64000, ir_len + 1)) is only needed for chunks beyond 2048 samples and for
 * nws_stream_reverb_tail; pass NULL to nws_stream_state_bytes when the stream never takes more than 16 frames at a time.
 */
size_t nws_stream_state_bytes(int B, int max_frames, int ir_len, const NwsReverbPlan* plan);
int nws_stream_reset(void* state, size_t state_bytes, void* stream);
int nws_stream_out_samples(int K, int first, int final);   /* 128 K (-64 for the first chunk, +64 for the final one) */
long long nws_stream_noise_start(int first, long long frames_seen);
int nws_stream_noise_draws(int K, int first, long long frames_seen);
int nws_stream_step(const NwsWeights* w, const float* fir_design, const NwsReverbPlan* plan, const void* reverb_tables,
                    const void* reverb_spectrum, void* state, size_t state_bytes, int B, int max_frames, const float* f0 /* (B,K) */,
                    const float* control /* (B,C,K) */, int C, int K, int first, int final, long long frames_seen,
                    long long nz_prev_start, float sample_rate, const float* phase_u, const float* rand_phase,
                    const float* noise_new, const float* noise_all, int noise_all_len, const float* ir, int ir_len,
                    float* out /* (B, M) */, float* pre_out /* optional (B, M) */, void* stream);
int nws_stream_reverb_tail(const NwsReverbPlan* plan, const void* reverb_tables, const void* reverb_spectrum, void* state,
                           size_t state_bytes, int B, int max_frames, int ir_len, float* tail_out /* (B, ir_len + 1) */,
                           void* workspace, size_t workspace_bytes, void* stream);

/*
 * Stand-alone forms of the reference's sub-modules ("the seven module classes" are public interface).  Inside
 * nws_forward the same arithmetic is fused; these entry points serve callers that invoke a sub-module on its own
 * (model.osc(f0), model.newt(exciter, emb), model.h_generator(emb), model.noise_synth(H), ...).
 */
/* HarmonicOscillator.forward (models/modules/generators.py:58-66): f0_up (B, N) Hz at sample rate, N a multiple of 128;
 * carry from nws_phase_carry(NULL, f0_up, B, N/128, .); phase_u / rand_phase as in nws_exciter_newt; out (B, 101, N) */
int nws_oscillator(const float* f0_up, const double* carry, const float* phase_u, const float* rand_phase, int B, int N,
                   float sample_rate, float* out, void* stream);
/* NEWT.forward / FastNEWT.forward on a materialised exciter (models/modules/shaping.py:67-79): exciter (B, 64, N),
 * film (B, 256, T) = newt.mlp(control_embedding) channel-major as the reference's Conv1d stack returns it; out (B, N).
 * w: shaper_* or lut fields, newt_out_w / newt_out_b. */
int nws_newt_apply(const NwsWeights* w, const float* exciter, const float* film, int B, int T, float* out, void* stream);
/* TimeDistributedMLP.forward (models/modules/dynamic.py:20-40), any sizes: x (B, in, T) -> y (B, out, T);
 * depth Conv1d(k=1) layers (HOST arrays of depth device pointers: w[i] (rows_i, cols_i), b[i]), LayerNorm over channels
 * (ln_g[i], ln_b[i], i < depth-1, biased variance, eps) + LeakyReLU(slope) after every layer but the last.
 * 1 <= depth <= 8 (depth 1 = a bare Conv1d(k=1), e.g. ControlModule.proj), max(in, hidden) <= ~600 (LDS). */
int nws_td_mlp(const float* x, int B, int in_size, int hidden, int out_size, int depth, int T, const float* const* w,
               const float* const* b, const float* const* ln_g, const float* const* ln_b, float ln_eps, float leaky_slope,
               float* y, void* stream);
/* TimeDistributedLayerNorm.forward (dynamic.py:11-17): LayerNorm over the channel axis of (B, C, T) */
int nws_td_layer_norm(const float* x, const float* gain, const float* bias, int B, int C, int T, float eps, float* y,
                      void* stream);
/* FiLM.forward (dynamic.py:6-8): y = gamma * x + beta on n equally laid out elements */
int nws_film(const float* x, const float* gamma, const float* beta, int64_t n, float* y, void* stream);
/* zero-phase FIR design of FIRNoiseSynth.forward (generators.py:22-28): H (B, 129, T) -> fir (B, T, 128) upper half-taps
 * (window * roll(irfft(H), 128))[128:256]; feed nws_fir_noise with them */
int nws_fir_from_h(const float* H, const float* fir_design /* (256,132) */, int B, int T, float* fir_out, void* stream);

/* ---- FastNEWT table (models/modules/shaping.py:107-119): table[s][i] = shaper_s(linspace(min,max,size)[i]) ---- */
int nws_shaper_table(const NwsWeights* w, int table_size, float table_min, float table_max, float* table_out, void* stream);

/* mixer_b (64) as K slot 0 and mixer_w (64,101) as K slots 1..101 (slots 102..111 zero) -> W_hi | W_lo fp16 fragments
   (7 K-steps x 2 M-tiles x 2 halves x 32 lanes x 8 halfs, twice) */
#define NWS_MIXER_FRAGS_BYTES 28672
int nws_mixer_frags(const float* mixer_w, const float* mixer_b, void* frags_out, void* stream);
/* X[s] = (sum_k |mixer_w[s][k]| + |mixer_b[s]|) * (1 + 2^-10): the worst-case magnitude of the exciter of shaper s
 * (NwsWeights.exciter_bound) */
int nws_exciter_bound(const float* mixer_w, const float* mixer_b, float* bound_out /* device, 64 floats */, void* stream);
/* Exact-mode companion of nws_lut_pairs: the TrainableNonlinearity weights (models/modules/shaping.py:15-37) of `w`
 * (shaper_* fields) as the (64, NWS_SHAPER_TURNS_ROW) fp32 table the fused kernel's shaper bank reads. */
#define NWS_SHAPER_TURNS_ROW 176
int nws_shaper_turns(const NwsWeights* w, float* table_out /* device, 64 * NWS_SHAPER_TURNS_ROW floats */, void* stream);

/* derived gather-friendly form of a FastNEWT table: pairs[s][i] = {table[s][i], fl(table[s][min(i+1,size-1)] - table[s][i])} */
int nws_lut_pairs(const float* table /* (64, size) */, int table_size, float* pairs_out /* (64, size, 2) */, void* stream);

/* exact shapers on an arbitrary (B,64,N) tensor (TrainableNonlinearity.forward, shaping.py:36-37) / LUT lookup (:136-151) */
int nws_shaper_apply(const NwsWeights* w, const float* x, int64_t B, int64_t N, float* y, void* stream);

/*
 * Whole forward (models/neural_waveshaping.py:74-90) as one enqueue: phase carries -> GRU -> frame MLPs ->
 * fused exciter+NEWT -> FIR noise (+sum) -> reverb.  All scratch lives in `workspace`
 * (nws_forward_workspace_bytes).  phase_u (101) and noise (N-1) are the two RNG draws of forward().
 */
typedef struct NwsForwardAux {
  const float* fir_design;       /* (256,132) */
  const NwsReverbPlan* plan;     /* host */
  const void* reverb_tables;
  const void* reverb_spectrum;
} NwsForwardAux;

size_t nws_forward_workspace_bytes(const NwsReverbPlan* plan, int B, int T);
int nws_forward(const NwsWeights* w, const NwsForwardAux* aux, const float* f0 /* (B,T) */, const float* control /* (B,C,T) */,
                int B, int C, int T, float sample_rate, const float* phase_u, const float* rand_phase,
                const float* noise, float* out /* (B,N) */, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same forward in two halves for throughput pipelines: `control` = phase carries + GRU (writes the head of the
 * workspace, nws_forward_control_bytes), `audio` = frame MLPs .. reverb (reads the head, uses the rest).  Same workspace
 * layout as nws_forward; one workspace per batch in flight.  The control half of batch i+1 may run on another stream while
 * the audio half of batch i runs (order them with events).  batched_gru != 0 selects nws_control_gru_batched.
 */
size_t nws_forward_control_bytes(int B, int T);
int nws_forward_control(const NwsWeights* w, const float* f0, const float* control, int B, int C, int T, int batched_gru,
                        void* workspace, size_t workspace_bytes, void* stream);
int nws_forward_audio(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                      const float* phase_u, const float* rand_phase, const float* noise, float* out, void* workspace,
                      size_t workspace_bytes, void* stream);

/* nws_forward_audio with two optional hipEvent_t hooks around the oscillator + waveshaper kernel (NULL = none): the stream
 * waits for `wait_before_exciter` right before that kernel and records `record_after_exciter` right after it.  A pipeline
 * that alternates batches over several streams chains them so that the VALU-saturated kernels of neighbouring batches run
 * one after the other while their matrix / memory kernels (frame MLPs, noise, reverb) overlap it (pipeline.py). */
int nws_forward_audio_ev(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                         const float* phase_u, const float* rand_phase, const float* noise, float* out, void* workspace,
                         size_t workspace_bytes, void* stream, void* wait_before_exciter, void* record_after_exciter);

/* The audio half in two parts for multi-GPU callers that push sub-batches of finished waveforms to their peers while the
 * reverb of the next sub-batch still runs: nws_forward_audio_pre = everything up to the reverb input of the
 * whole batch (kept in the workspace), nws_forward_reverb_rows = the reverb of rows [row0, row0 + nrows) (row0 even) into the
 * same rows of out (B, N).  pre + reverb_rows over all rows == nws_forward_audio, bit for bit. */
int nws_forward_audio_pre(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                          const float* phase_u, const float* rand_phase, const float* noise, void* workspace,
                          size_t workspace_bytes, void* stream);
int nws_forward_reverb_rows(const NwsForwardAux* aux, int B, int T, int row0, int nrows, float* out, void* workspace,
                            size_t workspace_bytes, void* stream);
/* The two above in ONE call for a fixed list of row blocks: audio_pre, then for q = 0 .. nblocks - 1 the reverb of rows
 * [row0[q], row0[q] + nrows[q]) followed by hipEventRecord(events[q], stream) when events and events[q] are non-NULL - a multi-GPU
 * caller's helper thread waits on those events (host side) and pushes each sub-batch as it completes (bench.py --gather-chunks:
 * one op call per step instead of 1 + nblocks, and no event record through the host language per block).  The blocks must tile
 * [0, B) in order, even sizes except the last.  Same bits as nws_forward_audio. */
int nws_forward_audio_blocks(const NwsWeights* w, const NwsForwardAux* aux, const float* f0, int B, int T, float sample_rate,
                             const float* phase_u, const float* rand_phase, const float* noise, float* out, void* workspace,
                             size_t workspace_bytes, void* stream, const int32_t* row0, const int32_t* nrows,
                             void* const* events /* hipEvent_t[nblocks] or NULL */, int nblocks);

/*
 * Perceptual-loudness feature, the step before the synthesis path:
 * neural_waveshaping_synthesis/data/utils/loudness_extraction.py:10-67 (extract_perceptual_loudness) with the shipped
 * configuration gin/data/urmp_4second_crepe.gin:11-14 = mean over bins of
 * amplitude_to_db(|stft(audio, n_fft, hop, hann, center/reflect)|, ref=max, amin, top_db), optionally (L + 80) / 80.
 * audio (B, N) fp32 -> out (B, 1 + N / hop).  `dft` is the constant windowed-DFT operand for n_fft (build once).
 * n_fft: power of two in [64, 2048]; 1 <= hop <= n_fft; N > n_fft / 2.  The reference's optional interpolate_fn is a
 * host callable on the returned frames and stays on the Python side.
 */
size_t nws_loudness_dft_bytes(int n_fft);
int nws_loudness_dft_matrix(int n_fft, float* dft_out, void* stream);
int nws_loudness_frames(int N, int hop);
size_t nws_loudness_workspace_bytes(int B, int N, int n_fft, int hop);
int nws_loudness(const float* audio, int B, int N, int n_fft, int hop, const float* dft, float amin, float top_db,
                 int normalise, float* out, void* workspace, size_t workspace_bytes, void* stream);

/*
 * ---- Runtime-size path (csrc/generic.hip): every gin-configurable size of the reference ------------------------------
 * The fused kernels above are compiled for gin/models/newt.gin.  These entry points take the sizes as arguments and run
 * one plain-fp32 stage kernel each (correct first, stage boundaries materialised); nws_forward_generic chains them into
 * NeuralWaveshaping.forward (models/neural_waveshaping.py:74-90) for ANY configuration:
 *   HarmonicOscillator.n_harmonics (generators.py:40-48), NeuralWaveshaping.n_waveshapers / control_hop (:31-62),
 *   NEWT.shaping_fn_size / out_channels (shaping.py:41-65), TrainableNonlinearity.depth (:15-34),
 *   ControlModule.hidden_size / embedding_size (neural_waveshaping.py:17-23), the h_generator's depth / hidden_size,
 *   FIRNoiseSynth.ir_length / hop_length (generators.py:13-19), Reverb.sr * length_in_seconds (shaping.py:156-158),
 *   FastNEWT table_size / table_min / table_max (shaping.py:83-105).
 */
/* TrainableNonlinearity(channels = n_shapers, width, depth) (shaping.py:15-37) or a FastNEWT table (lut != NULL) */
typedef struct NwsShaperDesc {
  int32_t n_shapers, width, depth, lut_size;
  float lut_min, lut_max;
  const float* in_scale; /* (n_shapers) */
  const float* w[8];     /* net.{2i}.weight: depth 1: (S); else layer 0: (S*width); hidden: (S*width, width); last: (S, width) */
  const float* b[8];     /* net.{2i}.bias */
  const float* lut;      /* (n_shapers, lut_size) or NULL */
} NwsShaperDesc;

typedef struct NwsGenericModel {
  int32_t control_size;   /* GRU input channels consumed (the reference's get_embedding always feeds 2, :69-72) */
  int32_t gru_hidden, embedding, n_harmonics, n_shapers;
  int32_t hop;            /* control_hop == FIRNoiseSynth.hop_length */
  int32_t newt_mlp_depth; /* 4 (NEWT hard-codes depth=4, shaping.py:53-55) */
  int32_t hgen_depth, hgen_hidden;
  int32_t fir_len;        /* FIRNoiseSynth.ir_length (even); the h_generator emits fir_len/2 + 1 bands */
  int32_t out_channels;   /* NEWT.out_channels (summed by forward's cat + sum(1)) */
  int32_t ir_len;         /* len(reverb.ir) = sr * length_in_seconds - 1 */
  float ln_eps, leaky_slope;
  const float *gru_w_ih, *gru_w_hh, *gru_b_ih, *gru_b_hh; /* (3H, control_size), (3H, H), (3H), (3H) */
  const float *proj_w, *proj_b;                           /* (embedding, H), (embedding) */
  const float *mixer_w, *mixer_b;                         /* (n_shapers, n_harmonics), (n_shapers) */
  const float* newt_mlp_w[8];
  const float* newt_mlp_b[8];
  const float* newt_ln_g[8];
  const float* newt_ln_b[8];
  const float* hgen_w[8];
  const float* hgen_b[8];
  const float* hgen_ln_g[8];
  const float* hgen_ln_b[8];
  const float *newt_out_w, *newt_out_b; /* (out_channels, n_shapers), (out_channels) */
  const float* noise_window;            /* (fir_len) */
  const float* ir;                      /* (ir_len) */
  NwsShaperDesc shaper;
} NwsGenericModel;

size_t nws_g_gru_workspace_bytes(int hidden);
/* torch.nn.GRU(C_in -> hidden, batch_first) over control[:, 0:C_in] of (B, C_total, T); out (B, T, hidden); h0 / hT optional */
int nws_g_gru(const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* control, int B,
              int C_total, int C_in, int hidden, int T, const float* h0, float* out, float* hT, void* workspace,
              size_t workspace_bytes, void* stream);
int nws_g_bth_to_bht(const float* x, int B, int T, int H, float* y, void* stream);
/* F.upsample(f0, T*hop, "linear") (f0 (B,T); or f0_up (B, T) given with hop = 1) + the double-accumulated cumsum and the
 * fp32 chain of generators.py:59: phase = fl(fl(tau * c) / sr).  f0_up_out optional. */
int nws_g_phase(const float* f0, const float* f0_up, int B, int T, int hop, float sample_rate, float* f0_up_out,
                float* phase_out, void* stream);
/* F.upsample(x, T * hop, mode="linear") row by row: x (rows, T) -> y (rows, T * hop) (neural_waveshaping.py:75, shaping.py:69) */
int nws_g_upsample(const float* x, int64_t rows, int T, int hop, float* y, void* stream);
/* HarmonicOscillator.forward (generators.py:58-66) for K harmonics: out (B, K, N) */
int nws_g_oscillator(const float* f0_up, const float* phase, const float* phase_u, const float* rand_phase, int K, int B, int N,
                     float sample_rate, float* out, void* stream);
/* nn.Conv1d(Cin, Cout, 1) on (B, Cin, N); bias may be NULL */
int nws_g_conv1x1(const float* x, const float* w, const float* bias, int B, int Cin, int Cout, int N, float* y, void* stream);
/* TrainableNonlinearity.forward / FastNEWT.shaping_fn on (rows = B * n_shapers, N) */
int nws_g_shaper_apply(const NwsShaperDesc* d, const float* x, int64_t rows, int64_t N, float* y, void* stream);
int nws_g_shaper_table(const NwsShaperDesc* d, int table_size, float table_min, float table_max, float* table_out, void* stream);
/* NEWT.forward before the mixer (shaping.py:68-76): exciter (B,S,N), film (B,4S,T) channel-major -> (B,S,N) */
int nws_g_film_shaper(const NwsShaperDesc* d, const float* exciter, const float* film, int B, int T, int hop, float* out,
                      void* stream);
/* FIRNoiseSynth (generators.py:21-35) for any even ir_length >= hop: H (B, L/2+1, T) -> taps (B, T, L); then the noise
 * branch (+ the sum over `add_channels` channels of add_in (B, add_channels, N), NULL = none) -> out (B, N) */
int nws_g_fir_design(const float* H, const float* window, int fir_len, int B, int T, float* fir_out, void* stream);
int nws_g_fir_noise(const float* fir, const float* noise, int fir_len, int hop, int B, int T, const float* add_in,
                    int add_channels, float* out, void* stream);
/* Reverb.forward without a transform plan (nws_reverb_plan serves every even circular length; this entry point is what is left):
 * the circular convolution summed in the time domain when L = max(N, ir_len + 1) is even; for ODD L the reference's own
 * rfft(L) / irfft(L - 1) result (shaping.py:171-173 passes no length to irfft: not a circular convolution), evaluated in float64
 * at O(L^2) with stream-ordered scratch (hipMallocAsync); NWS_ERR_UNSUPPORTED for odd L > 2^22 */
int nws_g_reverb_direct(const float* x, const float* ir, int ir_len, int B, int N, float* y, void* stream);

size_t nws_forward_generic_workspace_bytes(const NwsGenericModel* m, int B, int T);
/* plan / reverb_tables / reverb_spectrum / reverb_workspace: from nws_reverb_plan & co when the plan exists for
 * (T*hop, ir_len+1), else all NULL (time-domain reverb) */
int nws_forward_generic(const NwsGenericModel* m, const float* f0, const float* control, int B, int C, int T, float sample_rate,
                        const float* phase_u, const float* rand_phase, const float* noise, const NwsReverbPlan* plan,
                        const void* reverb_tables, const void* reverb_spectrum, void* reverb_workspace,
                        size_t reverb_workspace_bytes, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* Diagnostic entry points (kernel ablations for timing, the co-execution hazard probes) are declared in nws_hip_debug.h: they are
 * exported by the same library but are not part of the product ABI. */

/*
 * Live profiling of nws_forward (bench.py's roofline leg): hipEvents are recorded on the launch stream around
 * the stages selected by stage_mask (bit 0 phase carries, 1 GRU, 2 frame MLPs, 3 exciter+NEWT, 4 FIR noise,
 * 5 reverb) for the next `slots` calls.  nws_profile_collect synchronises the events and writes
 * ms_out[call][6] (-1 for unselected stages).
 */
#define NWS_N_STAGES 6
int nws_profile_begin(int slots, unsigned stage_mask);
int nws_profile_collect(float* ms_out /* host, slots*6 */, int* n_out /* host */);
int nws_profile_end(void);

/* Waveform exchange of the multi-GPU path, host side (ABI v6; SURVEY 8(e) - the reference has no collective code:
 * gin/train/train_newt.gin:13 is its only trace of more than one GPU).  nws_peer_push: for i < n, one asynchronous device-to-device
 * copy of `bytes` from src to dst[i] (a peer-mapped or local device pointer) on streams[i], then hipEventRecord(events[i],
 * streams[i]) when events and events[i] are non-NULL - one call per step instead of n copies + n records through the host
 * language.  nws_events_wait: hipEventSynchronize on every non-NULL event (HOST-side wait; the calling thread blocks). */
int nws_peer_push(int n, void* const* dst, const void* src, size_t bytes, void* const* streams, void* const* events);
int nws_events_wait(int n, void* const* events);
/* hipStreamSynchronize on every stream (HOST-side wait): the pushes of the previous step are awaited this way - no event record
 * per copy on the copy queues (every record is one more packet for the command processor beside the pipeline's own). */
int nws_streams_wait(int n, void* const* streams);

/* Which pipe of the command processor serves a stream's hardware queue (ABI v6; no reference counterpart: the reference enqueues
 * everything on one stream, models/neural_waveshaping.py:74-90 - this serves the throughput mode of that forward,
 * ForwardPipeline, whose five streams must sit on the pipes in a fixed pattern: DESIGN.md 5.2).
 * Launches a grid of `groups` one-wave workgroups on `stream_hold` that each hold 64 000 B of LDS (two per CU: the grid is handed
 * out in rounds of 512) and spin for `spin_us` microseconds, and right behind it ONE wave on `stream_touch` that stores the
 * wall clock; synchronises both streams (a measurement, not a hot-path call) and returns in *frac_out
 *     (touch time - first workgroup start) / (last workgroup start - first workgroup start):
 * ~0 when the second queue was served while the first was dispatching (different pipes), >= ~1 when it had to wait for the
 * whole grid to be handed out (same pipe, or one hardware queue shared by both streams); -1 when the grid was too small to
 * be held (groups <= 512).  `scratch`: device memory, 32 bytes. */
int nws_queue_probe(void* stream_hold, void* stream_touch, int groups, int spin_us, unsigned long long* scratch, float* frac_out /* host */);

#ifdef __cplusplus
}
#endif
#endif /* NWS_HIP_H */
