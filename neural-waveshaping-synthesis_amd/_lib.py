"""ctypes binding of libnws_hip.so (the C-ABI declared in include/nws_hip.h).

There is NO fallback: if the library cannot be loaded every entry point raises, so a GPU run can
never silently execute anything but the hand-written HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnws_hip.so")

PROBE_LIB_PATH = os.path.join(_HERE, "libnws_probe.so")     # include/nws_probe.h: the co-execution hazard probe (tools / tests only)

ABI_VERSION = 6          # include/nws_hip.h NWS_ABI_VERSION
EXCITER_VALU_FILM = 1    # NwsWeights.exciter_opts bits (include/nws_hip.h)
EXCITER_ONE_TERM = 2
EXCITER_HYBRID = 4
EXCITER_HYBRID_W = 8
EXCITER_BANK_NOFRACT = 16
N_HARMONICS = 101
N_SHAPERS = 64
HIDDEN = 128
HOP = 128
FIR_LEN = 256
FIR_HALF = 128           # taps per row handed from the frame MLPs to the noise kernels: h[128 .. 255] (include/nws_hip.h)
MLP_FRAGS_BYTES = 1474560
N_BANDS = 129
FILM_CH = 256
SHAPER_WIDTH = 8
SHAPER_TURNS_ROW = 176
FIR_DESIGN_COLS = 132

_fp = C.c_void_p  # device pointers travel as integers


class NwsWeights(C.Structure):
    _fields_ = [
        ("gru_w_ih", _fp), ("gru_w_hh", _fp), ("gru_b_ih", _fp), ("gru_b_hh", _fp),
        ("proj_w", _fp), ("proj_b", _fp),
        ("mixer_w", _fp), ("mixer_b", _fp), ("mixer_frags", _fp),
        ("newt_mlp_w", _fp * 4), ("newt_mlp_b", _fp * 4), ("newt_ln_g", _fp * 3), ("newt_ln_b", _fp * 3),
        ("hgen_w", _fp * 4), ("hgen_b", _fp * 4), ("hgen_ln_g", _fp * 3), ("hgen_ln_b", _fp * 3),
        ("mlp_frags", _fp),
        ("shaper_in_scale", _fp),
        ("shaper_w0", _fp), ("shaper_b0", _fp), ("shaper_w2", _fp), ("shaper_b2", _fp),
        ("shaper_w4", _fp), ("shaper_b4", _fp), ("shaper_w6", _fp), ("shaper_b6", _fp), ("shaper_turns", _fp),
        ("lut", _fp), ("lut_pairs", _fp), ("lut_size", C.c_int32), ("lut_min", C.c_float), ("lut_max", C.c_float),
        ("newt_out_w", _fp), ("newt_out_b", _fp),
        ("noise_window", _fp),
        ("exciter_opts", C.c_int32),
        ("exciter_bound", _fp),
    ]


class NwsReverbPlan(C.Structure):
    _fields_ = [("L", C.c_int32), ("N1", C.c_int32), ("N2", C.c_int32), ("Lc", C.c_int32), ("hist", C.c_int32), ("nblk", C.c_int32),
                ("reserved", C.c_int32 * 2)]

    def as_tensor(self):
        import torch
        return torch.tensor([self.L, self.N1, self.N2, self.Lc, self.hist, self.nblk, 0, 0], dtype=torch.int32)


class NwsForwardAux(C.Structure):
    _fields_ = [("fir_design", _fp), ("plan", C.POINTER(NwsReverbPlan)), ("reverb_tables", _fp),
                ("reverb_spectrum", _fp)]


class NwsShaperDesc(C.Structure):
    _fields_ = [("n_shapers", C.c_int32), ("width", C.c_int32), ("depth", C.c_int32), ("lut_size", C.c_int32),
                ("lut_min", C.c_float), ("lut_max", C.c_float),
                ("in_scale", _fp), ("w", _fp * 8), ("b", _fp * 8), ("lut", _fp)]


class NwsGenericModel(C.Structure):
    _fields_ = [("control_size", C.c_int32), ("gru_hidden", C.c_int32), ("embedding", C.c_int32), ("n_harmonics", C.c_int32),
                ("n_shapers", C.c_int32), ("hop", C.c_int32), ("newt_mlp_depth", C.c_int32), ("hgen_depth", C.c_int32),
                ("hgen_hidden", C.c_int32), ("fir_len", C.c_int32), ("out_channels", C.c_int32), ("ir_len", C.c_int32),
                ("ln_eps", C.c_float), ("leaky_slope", C.c_float),
                ("gru_w_ih", _fp), ("gru_w_hh", _fp), ("gru_b_ih", _fp), ("gru_b_hh", _fp),
                ("proj_w", _fp), ("proj_b", _fp), ("mixer_w", _fp), ("mixer_b", _fp),
                ("newt_mlp_w", _fp * 8), ("newt_mlp_b", _fp * 8), ("newt_ln_g", _fp * 8), ("newt_ln_b", _fp * 8),
                ("hgen_w", _fp * 8), ("hgen_b", _fp * 8), ("hgen_ln_g", _fp * 8), ("hgen_ln_b", _fp * 8),
                ("newt_out_w", _fp), ("newt_out_b", _fp), ("noise_window", _fp), ("ir", _fp),
                ("shaper", NwsShaperDesc)]


_PROTOTYPES = {
    "nws_abi_version": (C.c_int, []),
    "nws_sizeof": (C.c_size_t, [C.c_int]),
    "nws_error_string": (C.c_char_p, [C.c_int]),
    "nws_selftest_mfma": (C.c_int, [_fp, _fp]),
    "nws_peer_push": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), _fp, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "nws_events_wait": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nws_streams_wait": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nws_queue_probe": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, C.POINTER(C.c_float)]),
    "nws_debug_queue_busy": (C.c_int, [C.c_int, C.c_int, _fp, _fp]),
    "nws_sin": (C.c_int, [_fp, _fp, C.c_int64, _fp]),
    "nws_phase_carry": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "nws_exciter_newt": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float,
                                   _fp, _fp, _fp]),
    "nws_exciter_newt_add": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float,
                                       _fp, _fp, _fp]),
    "nws_control_gru": (C.c_int, [C.POINTER(NwsWeights), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_control_gru_state": (C.c_int, [C.POINTER(NwsWeights), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "nws_control_gru_carry": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "nws_control_gru_batched": (C.c_int, [C.POINTER(NwsWeights), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "nws_frame_mlps": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    "nws_mlp_frags": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, _fp]),
    "nws_debug_frame_mlps_kernel": (C.c_int, [C.c_int]),
    "nws_debug_frame_mlps_probe": (C.c_int, [_fp]),
    "nws_fir_design_matrix": (C.c_int, [_fp, _fp, _fp]),
    "nws_fir_noise": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "nws_fir_noise_window": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, _fp]),
    "nws_reverb_linear_chunk": (C.c_int, [C.POINTER(NwsReverbPlan), _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, _fp,
                                          C.c_size_t, _fp]),
    "nws_reverb_plan": (C.c_int, [C.c_int, C.c_int, C.POINTER(NwsReverbPlan)]),
    "nws_reverb_plan_serves": (C.c_int, [C.POINTER(NwsReverbPlan), C.c_int, C.c_int]),
    "nws_reverb_table_bytes": (C.c_size_t, [C.POINTER(NwsReverbPlan)]),
    "nws_reverb_spectrum_bytes": (C.c_size_t, [C.POINTER(NwsReverbPlan)]),
    "nws_reverb_workspace_bytes": (C.c_size_t, [C.POINTER(NwsReverbPlan), C.c_int]),
    "nws_reverb_build_tables": (C.c_int, [C.POINTER(NwsReverbPlan), _fp, _fp]),
    "nws_reverb_ir_spectrum": (C.c_int, [C.POINTER(NwsReverbPlan), _fp, _fp, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "nws_reverb": (C.c_int, [C.POINTER(NwsReverbPlan), _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "nws_shaper_table": (C.c_int, [C.POINTER(NwsWeights), C.c_int, C.c_float, C.c_float, _fp, _fp]),
    "nws_mixer_frags": (C.c_int, [_fp, _fp, _fp, _fp]),
    "nws_exciter_bound": (C.c_int, [_fp, _fp, _fp, _fp]),
    "nws_debug_film_frags": (C.c_int, [C.POINTER(NwsWeights), _fp, C.c_int, C.c_int, _fp, _fp]),
    "nws_shaper_turns": (C.c_int, [_fp, _fp, _fp]),
    "nws_lut_pairs": (C.c_int, [_fp, C.c_int, _fp, _fp]),
    "nws_shaper_apply": (C.c_int, [C.POINTER(NwsWeights), _fp, C.c_int64, C.c_int64, _fp, _fp]),
    "nws_forward_workspace_bytes": (C.c_size_t, [C.POINTER(NwsReverbPlan), C.c_int, C.c_int]),
    "nws_forward": (C.c_int, [C.POINTER(NwsWeights), C.POINTER(NwsForwardAux), _fp, _fp, C.c_int, C.c_int, C.c_int,
                              C.c_float, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "nws_forward_control_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "nws_forward_control": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, _fp]),
    "nws_forward_audio": (C.c_int, [C.POINTER(NwsWeights), C.POINTER(NwsForwardAux), _fp, C.c_int, C.c_int, C.c_float,
                                    _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "nws_forward_audio_ev": (C.c_int, [C.POINTER(NwsWeights), C.POINTER(NwsForwardAux), _fp, C.c_int, C.c_int, C.c_float,
                                       _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp, _fp, _fp]),
    "nws_forward_audio_pre": (C.c_int, [C.POINTER(NwsWeights), C.POINTER(NwsForwardAux), _fp, C.c_int, C.c_int, C.c_float,
                                        _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
    "nws_forward_reverb_rows": (C.c_int, [C.POINTER(NwsForwardAux), C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "nws_forward_audio_blocks": (C.c_int, [C.POINTER(NwsWeights), C.POINTER(NwsForwardAux), _fp, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp, _fp,
                                           _fp, C.c_size_t, _fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.c_int]),
    "nws_loudness_dft_bytes": (C.c_size_t, [C.c_int]),
    "nws_loudness_dft_matrix": (C.c_int, [C.c_int, _fp, _fp]),
    "nws_loudness_frames": (C.c_int, [C.c_int, C.c_int]),
    "nws_loudness_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "nws_loudness": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_float, C.c_float, C.c_int, _fp, _fp,
                               C.c_size_t, _fp]),
    "nws_debug_exciter_newt": (C.c_int, [C.c_int, C.POINTER(NwsWeights), _fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float,
                                         _fp, _fp]),
    "nws_debug_control_gru": (C.c_int, [C.c_int, C.POINTER(NwsWeights), _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_debug_sin": (C.c_int, [C.c_int, _fp, _fp, C.c_int64, C.c_int, _fp]),
    "nws_oscillator": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_float, _fp, _fp]),
    "nws_newt_apply": (C.c_int, [C.POINTER(NwsWeights), _fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "nws_td_mlp": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_fp), C.POINTER(_fp),
                             C.POINTER(_fp), C.POINTER(_fp), C.c_float, C.c_float, _fp, _fp]),
    "nws_td_layer_norm": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp]),
    "nws_film": (C.c_int, [_fp, _fp, _fp, C.c_int64, _fp, _fp]),
    "nws_fir_from_h": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp, _fp]),
    "nws_g_gru_workspace_bytes": (C.c_size_t, [C.c_int]),
    "nws_g_gru": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp,
                            C.c_size_t, _fp]),
    "nws_g_bth_to_bht": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_g_phase": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp]),
    "nws_g_upsample": (C.c_int, [_fp, C.c_int64, C.c_int, C.c_int, _fp, _fp]),
    "nws_g_oscillator": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp]),
    "nws_g_conv1x1": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_g_shaper_apply": (C.c_int, [C.POINTER(NwsShaperDesc), _fp, C.c_int64, C.c_int64, _fp, _fp]),
    "nws_g_shaper_table": (C.c_int, [C.POINTER(NwsShaperDesc), C.c_int, C.c_float, C.c_float, _fp, _fp]),
    "nws_g_film_shaper": (C.c_int, [C.POINTER(NwsShaperDesc), _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_g_fir_design": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_g_fir_noise": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp]),
    "nws_g_reverb_direct": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "nws_forward_generic_workspace_bytes": (C.c_size_t, [C.POINTER(NwsGenericModel), C.c_int, C.c_int]),
    "nws_forward_generic": (C.c_int, [C.POINTER(NwsGenericModel), _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp, _fp,
                                      C.POINTER(NwsReverbPlan), _fp, _fp, _fp, C.c_size_t, _fp, _fp, C.c_size_t, _fp]),
    "nws_stream_state_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.POINTER(NwsReverbPlan)]),
    "nws_stream_reset": (C.c_int, [_fp, C.c_size_t, _fp]),
    "nws_stream_out_samples": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "nws_stream_noise_start": (C.c_longlong, [C.c_int, C.c_longlong]),
    "nws_stream_noise_draws": (C.c_int, [C.c_int, C.c_int, C.c_longlong]),
    "nws_stream_step": (C.c_int, [C.POINTER(NwsWeights), _fp, C.POINTER(NwsReverbPlan), _fp, _fp, _fp, C.c_size_t, C.c_int, C.c_int,
                                  _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_float, _fp, _fp,
                                  _fp, _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp]),
    "nws_stream_reverb_tail": (C.c_int, [C.POINTER(NwsReverbPlan), _fp, _fp, _fp, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp, _fp,
                                         C.c_size_t, _fp]),
    "nws_profile_begin": (C.c_int, [C.c_int, C.c_uint]),
    "nws_profile_collect": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "nws_profile_end": (C.c_int, []),
}

_PROBE_PROTOTYPES = {
    "nws_coexec_pk_probe": (C.c_int, [C.c_int, C.c_int, _fp, _fp]),
    "nws_coexec_pk_probe2": (C.c_int, [C.c_int, C.c_int, _fp, _fp]),
    "nws_coexec_pk_probe_mixed": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "nws_coexec_mfma_load": (C.c_int, [C.c_int, C.c_int, C.c_int, _fp, _fp]),
}

STAGE_NAMES = ("phase_carry", "control_gru", "frame_mlps", "exciter_newt", "fir_noise", "reverb")

EXPORTED_SYMBOLS = tuple(_PROTOTYPES)
PROBE_SYMBOLS = tuple(_PROBE_PROTOTYPES)

_lib = None
_probe = None


class NwsError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises if it is missing - never falls back."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NwsError(
                f"{LIB_PATH} not found: build the HIP kernels first "
                "(python __graft_entry__.py build, or python neural-waveshaping-synthesis_amd/build.py). "
                "There is no CPU/PyTorch fallback for the NEWT forward path.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.nws_abi_version() != ABI_VERSION:
            raise NwsError(f"{LIB_PATH} has ABI version {handle.nws_abi_version()}, this package binds version "
                           f"{ABI_VERSION}: rebuild it (python __graft_entry__.py build)")
        for which, struct in enumerate((NwsWeights, NwsReverbPlan, NwsForwardAux, NwsShaperDesc, NwsGenericModel)):
            if handle.nws_sizeof(which) != C.sizeof(struct):
                raise NwsError(f"struct layout mismatch for {struct.__name__}: library {handle.nws_sizeof(which)} B, "
                               f"binding {C.sizeof(struct)} B")
        _lib = handle
    return _lib


def probe_lib():
    """libnws_probe.so (include/nws_probe.h): the hazard probe kernels, for tools/ and tests/ - never needed by the product path."""
    global _probe
    if _probe is None:
        if not os.path.exists(PROBE_LIB_PATH):
            raise NwsError(f"{PROBE_LIB_PATH} not found: python neural-waveshaping-synthesis_amd/build.py builds it next to libnws_hip.so")
        handle = C.CDLL(PROBE_LIB_PATH)
        for name, (res, args) in _PROBE_PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _probe = handle
    return _probe


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().nws_error_string(rc)
        raise NwsError(f"{what or 'nws call'} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a contiguous fp32 (or fp64) CUDA tensor; None -> NULL."""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    import torch

    return torch.cuda.current_stream().cuda_stream
