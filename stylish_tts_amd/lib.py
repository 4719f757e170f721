"""ctypes binding of libstylish_hip.so (include/stylish_hip.h).  Fails loudly when the library is missing."""
import ctypes as C
import os

# Hardware queues per process: the training step uses four HIP streams and a communication library adds more; above
# four concurrently active streams the runtime's default (4 queues) degrades badly, two queues do not (DESIGN.md
# section 4).  Effective only when this module is imported before the HIP runtime is loaded (before `import torch`);
# otherwise export GPU_MAX_HW_QUEUES=2 in the launch environment.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libstylish_hip.so")
if os.environ.get("STY_LIB_VARIANT"):  # kernel-tuning aid (tools/build_variant.sh): same ABI, different build flags
    LIB_PATH = os.path.join(HERE, f"libstylish_hip_{os.environ['STY_LIB_VARIANT']}.so")


class StyError(RuntimeError):
    pass


class VocoderIO(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int)] + [
        (n, C.c_void_p) for n in ("mel", "style", "pitch", "voiced", "noise", "prior_override")
    ] + [("seed", C.c_uint64), ("audio", C.c_void_p)] + [
        (n, C.c_void_p) for n in ("tap_conformer_out", "tap_prior", "tap_har_spec", "tap_har_phase",
                                  "tap_logamp_prior", "tap_phase_prior", "tap_trunk", "tap_logamp")
    ]


class SpeechIO(C.Structure):
    _fields_ = [("B", C.c_int), ("L", C.c_int), ("T", C.c_int)] + [
        (n, C.c_void_p) for n in ("texts", "text_lengths", "alignment", "pitch", "energy", "voiced", "style",
                                  "denormal_pitch", "noise", "prior_override")
    ] + [("seed", C.c_uint64), ("audio", C.c_void_p), ("tap_text_encoding", C.c_void_p),
         ("tap_decoder_out", C.c_void_p), ("voc_taps", VocoderIO), ("style_stream", C.c_void_p)]


class TrainOpts(C.Structure):
    _fields_ = [("bn_batch_stats", C.c_int), ("sn_power_iter", C.c_int), ("f0_smooth", C.c_int),
                ("energy_smooth", C.c_int), ("bn_momentum", C.c_float), ("dropout_seed", C.c_uint),
                ("text_dropout", C.c_float), ("compute_bf16", C.c_int), ("frozen", C.c_int),
                ("block_dropout", C.c_float)]


class SpecDiscPtrs(C.Structure):
    """sty_specdisc_params / sty_specdisc_grads: g, v, bias of `discriminators.0..4` then `out.0..4`."""
    _fields_ = [("g", C.c_void_p * 10), ("v", C.c_void_p * 10), ("bias", C.c_void_p * 10)]


class CfDiscParams(C.Structure):
    """sty_cfdisc_params"""
    _fields_ = [("conv_w", C.c_void_p * 12), ("conv_b", C.c_void_p * 12), ("bn_w", C.c_void_p * 9), ("bn_b", C.c_void_p * 9),
                ("bn_rm", C.c_void_p * 9), ("bn_rv", C.c_void_p * 9)]


class CfDiscGrads(C.Structure):
    """sty_cfdisc_grads"""
    _fields_ = [("conv_w", C.c_void_p * 12), ("conv_b", C.c_void_p * 12), ("bn_w", C.c_void_p * 9), ("bn_b", C.c_void_p * 9)]


# every symbol include/stylish_hip.h declares: name -> (restype, argtypes)
_P, _I, _SZP = C.c_void_p, C.c_int, C.POINTER(C.c_size_t)
SYMBOLS = {
    "sty_version": (C.c_int, []),
    "sty_last_error": (C.c_char_p, []),
    "sty_stft64_bases_host": (None, [_P]),
    "sty_model_create": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "sty_model_destroy": (None, [_P]),
    "sty_model_bind": (C.c_int, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64)]),
    "sty_model_finalize": (C.c_int, [_P]),
    "sty_model_num_keys": (C.c_int, [_P]),
    "sty_model_key": (C.c_char_p, [_P, _I]),
    "sty_model_prepare": (C.c_int, [_P, _P]),
    "sty_model_invalidate": (C.c_int, [_P]),
    "sty_vocoder_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_vocoder_fwd": (C.c_int, [_P, C.POINTER(VocoderIO), _P, C.c_size_t, _P]),
    "sty_speech_workspace_bytes": (C.c_int, [_P, _I, _I, _I, _SZP]),
    "sty_speech_fwd": (C.c_int, [_P, C.POINTER(SpeechIO), _P, C.c_size_t, _P]),
    "sty_style_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_style_fwd": (C.c_int, [_P, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "sty_mel_workspace_bytes": (C.c_int, [_I, _I, _I, _I, _SZP]),
    "sty_mel_fwd": (C.c_int, [_I, _I, _P, _I, _I, _I, C.c_float, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    "sty_multispec_workspace_bytes": (C.c_int, [_I, _I, _SZP]),
    "sty_multispec_fwd": (C.c_int, [_I, _I, _P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, C.c_size_t, _P]),
    "sty_alignment_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P]),
    "sty_convnext_fwd": (C.c_int, [_P, C.c_char_p, _I, _I, _I, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_resblock_fwd": (C.c_int, [_P, C.c_char_p, _I, _I, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_block_train_workspace_bytes": (C.c_int, [_P, C.c_char_p, C.c_char_p, _I, _I, _I, _SZP]),
    "sty_block_fwd_bwd": (C.c_int, [_P, C.c_char_p, C.c_char_p, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_attention_workspace_bytes": (C.c_int, [_I, _I, _I, _SZP]),
    "sty_attention_fwd_bwd": (C.c_int, [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_stft64_fwd": (C.c_int, [_I, _I, _P, _P, _P, _P]),
    "sty_istft64_fwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P]),
    "sty_source_fwd": (C.c_int, [_I, _I, _P, _P, _P, C.c_uint64, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_source_workspace_bytes": (C.c_int, [_I, _I, _SZP]),
    "sty_model_enable_training": (C.c_int, [_P]),
    "sty_model_bind_grad": (C.c_int, [_P, C.c_char_p, _P]),
    "sty_model_set_train_opts": (C.c_int, [_P, C.POINTER(TrainOpts)]),
    "sty_vocoder_train_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_vocoder_fwd_train": (C.c_int, [_P, C.POINTER(VocoderIO), _P, C.c_size_t, _P]),
    "sty_vocoder_bwd": (C.c_int, [_P, _P, _P, _P, _P]),
    "sty_speech_train_workspace_bytes": (C.c_int, [_P, _I, _I, _I, _SZP]),
    "sty_speech_fwd_train": (C.c_int, [_P, C.POINTER(SpeechIO), _P, C.c_size_t, _P]),
    "sty_speech_bwd": (C.c_int, [_P, _P, _P, _P, _P]),
    "sty_speech_prepare_train": (C.c_int, [_P, _P]),
    "sty_speech_bwd_pe": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "sty_speech_d_style_ready": (C.c_int, [_P, _P]),
    "sty_style_train_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_style_fwd_train": (C.c_int, [_P, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "sty_style_prepare_train": (C.c_int, [_P, _P]),
    "sty_style_bwd": (C.c_int, [_P, _P, _P]),
    "sty_style_tap": (C.c_int, [_P, _I, _I, _P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _P]),
    "sty_duration_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_duration_fwd": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_pitch_energy_workspace_bytes": (C.c_int, [_P, _I, _I, _I, _SZP]),
    "sty_pitch_energy_fwd": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_pitch_style_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_pitch_style_fwd": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_conv1d_workspace_bytes": (C.c_int, [_I, _I, _I, _SZP]),
    "sty_conv1d_fwd": (C.c_int, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _I, _P]),
    "sty_conv1d_bwd_workspace_bytes": (C.c_int, [_I, _I, _I, _I, _I, _SZP]),
    "sty_conv1d_bwd": (C.c_int, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _I, _P]),
    "sty_adamw_step": (C.c_int, [C.c_size_t, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 _I, C.c_float, _P]),
    "sty_adamw_step_scaled": (C.c_int, [C.c_size_t, _P, _P, _P, _P, C.c_double, _P, C.c_float, C.c_float, C.c_float,
                                        C.c_float, _I, C.c_float, _P]),
    "sty_disc_lr_track": (C.c_int, [_P, _P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "sty_model_set_grad_hook": (C.c_int, [_P, _P, _P]),
    "sty_acoustic_loss_workspace_bytes": (C.c_int, [_I, _I, _SZP]),
    "sty_acoustic_loss_target": (C.c_int, [_I, _I, _P, _P, C.c_size_t, _P]),
    "sty_acoustic_loss_fwd_bwd": (C.c_int, [_I, _I, _P, _P, C.c_float, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    "sty_specdisc_workspace_bytes": (C.c_int, [_I, _I, _I, _I, _SZP]),
    "sty_specdisc_forward": (C.c_int, [_P, _I, _I, _I, _P, _P, _I, _P, C.c_size_t, _P]),
    "sty_specdisc_losses": (C.c_int, [_P, _I, _I, _I, _P, _P, C.c_float, _P, _P, C.c_float, _P, _P, _I, _P, C.c_size_t,
                                      _P]),
    "sty_prediction_to_duration": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P]),
    "sty_duration_loss_fwd_bwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P,
                                            C.c_size_t, _P]),
    "sty_duration_train_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_duration_fwd_train": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_duration_bwd": (C.c_int, [_P, _P, _P, _P]),
    "sty_pitch_loss_fwd_bwd": (C.c_int, [_I, _I, _P, _P, C.c_float, _I, _P, _P, _P, C.c_size_t, _P]),
    "sty_pitch_energy_train_workspace_bytes": (C.c_int, [_P, _I, _I, _I, _SZP]),
    "sty_pitch_energy_fwd_train": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_pitch_energy_bwd": (C.c_int, [_P, _P, _P, _P, _P]),
    "sty_pitch_style_train_workspace_bytes": (C.c_int, [_P, _I, _I, _SZP]),
    "sty_pitch_style_fwd_train": (C.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "sty_pitchdisc_workspace_bytes": (C.c_int, [_I, _I, _I, _I, _I, _SZP]),
    "sty_pitchdisc_forward": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, _P, C.c_size_t, _P]),
    "sty_pitchdisc_losses": (C.c_int, [_P, _I, _I, _I, _I, _P, _P, C.c_float, _P, _P, C.c_float, _P, _P, _P, C.c_size_t,
                                       _P]),
    "sty_cfdisc_workspace_bytes": (C.c_int, [_I, _I, _I, _SZP]),
    "sty_cfdisc_forward": (C.c_int, [_P, _I, _I, _P, _P, C.c_float, _I, _P, C.c_size_t, _P]),
    "sty_cfdisc_losses": (C.c_int, [_P, _I, _I, _P, _P, C.c_float, _P, _P, C.c_float, _P, _P, C.c_float, _I, _P,
                                    C.c_size_t, _P]),
    "sty_acoustic_gan_workspace_bytes": (C.c_int, [_I, _I, _I, _SZP]),
    "sty_acoustic_gan_loss_fwd_bwd": (C.c_int, [_I, _I, _P, _P, C.c_float, C.c_float, C.c_float, _P, C.c_float, _P, _I, _P,
                                                _P, _P, _P, C.c_size_t, _P, C.c_size_t, _I, _P]),
    "sty_comm_unique_id": (C.c_int, [_P]),
    "sty_comm_init": (C.c_int, [_P, _I, _I, _I, C.POINTER(C.c_void_p)]),
    "sty_comm_allreduce_bucket": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "sty_comm_wait": (C.c_int, [_P, _P]),
    "sty_comm_set_stream": (C.c_int, [_P, _P]),
    "sty_comm_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "sty_comm_destroy": (C.c_int, [_P]),
    "sty_prof_enable": (C.c_int, [_I]),
    "sty_prof_only": (C.c_int, [C.c_char_p]),
    "sty_set_single_stream": (C.c_int, [_I]),
    "sty_prof_report": (C.c_int, [_P, _I]),
}


class ProfRow(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("inst", C.c_char * 144), ("launches", C.c_uint64), ("ms", C.c_double),
                ("flops", C.c_double), ("bytes", C.c_double)]


def prof_report(cap=64, by_inst=False):
    """Drain the in-situ kernel timers: list of dicts (name, launches, ms, flops, bytes, insts).  The library keeps one row per
    (family, instantiation); by default they are summed per family with `insts` = {instantiation: launches}; by_inst=True
    returns the library's rows as they are (with `inst`)."""
    cap = max(cap, 4096)  # rows are per (family, instantiation) before they are grouped
    rows = (ProfRow * cap)()
    n = LIB.sty_prof_report(C.cast(rows, C.c_void_p), cap)
    if n < 0:
        check(n)
    raw = [dict(name=r.name.decode(), inst=r.inst.decode(), launches=int(r.launches), ms=r.ms, flops=r.flops, bytes=r.bytes)
           for r in rows[:min(n, cap)]]
    return raw if by_inst else group_families(raw)


def group_families(raw):
    fam = {}
    for r in raw:
        f = fam.setdefault(r["name"], dict(name=r["name"], launches=0, ms=0.0, flops=0.0, bytes=0.0, insts={}))
        for k in ("launches", "ms", "flops", "bytes"):
            f[k] += r[k]
        if r["inst"]:
            f["insts"][r["inst"]] = f["insts"].get(r["inst"], 0) + r["launches"]
    return list(fam.values())

GRAD_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_int)  # sty_grad_hook(user, segment)

LIB = None


def load():
    """Load the in-tree shared library (built by `python -m stylish_tts_amd.build` / __graft_entry__.build())."""
    global LIB
    if LIB is not None:
        return LIB
    if not os.path.exists(LIB_PATH):
        raise StyError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` (hipcc, gfx950). "
                       "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    LIB = lib
    return lib


def check(rc):
    if rc != 0:
        raise StyError(f"libstylish_hip: status {rc}: {LIB.sty_last_error().decode()}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
