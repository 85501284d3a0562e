"""ctypes binding of libvtd_hip.so (the C ABI in include/vtd.h).

Loading is strict: there is no CPU or eager-PyTorch fallback anywhere in this package.  ``require()``
raises when the library is missing (run ``python __graft_entry__.py`` / ``build_native.py``) or when
no HIP device is visible, so a mis-provisioned GPU box fails loudly instead of silently passing.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VTD_LIB_VARIANT=<tag>: an instrumented build (build_native.py, tools/conv_experiment.sh) living beside the product library
_VARIANT = os.environ.get("VTD_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "_lib", f"libvtd_hip_{_VARIANT}.so" if _VARIANT else "libvtd_hip.so")

_lib = None


class NativeError(RuntimeError):
    pass


class TrocrConfig(C.Structure):
    """vtd_trocr_config of include/vtd.h"""
    _fields_ = [("image_size", C.c_int32), ("patch_size", C.c_int32), ("enc_hidden", C.c_int32), ("enc_layers", C.c_int32),
                ("enc_heads", C.c_int32), ("enc_ffn", C.c_int32), ("enc_qkv_bias", C.c_int32), ("enc_ln_eps", C.c_float),
                ("dec_hidden", C.c_int32), ("dec_layers", C.c_int32), ("dec_heads", C.c_int32), ("dec_ffn", C.c_int32),
                ("vocab_size", C.c_int32), ("max_positions", C.c_int32), ("dec_ln_eps", C.c_float),
                ("decoder_start_token_id", C.c_int32), ("eos_token_id", C.c_int32), ("pad_token_id", C.c_int32), ("max_length", C.c_int32)]


class Detection(C.Structure):
    _fields_ = [("bbox", C.c_int32 * 4), ("polygon", C.c_int32 * 8), ("confidence", C.c_float), ("area", C.c_float),
                ("first_x", C.c_int32), ("first_y", C.c_int32)]


# name -> (restype, argtypes); kept in one table so tests can check it against include/vtd.h
SIGNATURES = {
    "vtd_version": (C.c_char_p, []),
    "vtd_strerror": (C.c_char_p, [C.c_int]),
    "vtd_device_count": (C.c_int, []),
    "vtd_detector_create": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vtd_detector_destroy": (None, [C.c_void_p]),
    "vtd_detector_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "vtd_detector_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vtd_detector_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtd_detector_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "vtd_detector_set_input_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_detector_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtd_detector_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "vtd_detector_macs_per_frame": (C.c_int64, [C.c_void_p]),
    "vtd_detector_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "vtd_detector_num_ops": (C.c_int, [C.c_void_p]),
    "vtd_detector_get_profile": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_double), C.c_void_p]),
    "vtd_detector_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vtd_detector_get_tuning": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "vtd_detector_tuning_measured": (C.c_int, [C.c_void_p]),
    "vtd_postproc_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vtd_postproc_destroy": (None, [C.c_void_p]),
    "vtd_copy_to_pinned_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "vtd_postproc_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "vtd_recognizer_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "vtd_recognizer_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vtd_recognizer_destroy": (None, [C.c_void_p]),
    "vtd_recognizer_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "vtd_recognizer_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtd_recognizer_crop_resize": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_recognizer_set_input_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_recognizer_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "vtd_recognizer_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "vtd_recognizer_macs_per_crop": (C.c_int64, [C.c_void_p]),
    "vtd_recognizer_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vtd_recognizer_get_tuning": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "vtd_recognizer_tuning_measured": (C.c_int, [C.c_void_p]),
    "vtd_trocr_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "vtd_trocr_destroy": (None, [C.c_void_p]),
    "vtd_trocr_set_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "vtd_trocr_finalize": (C.c_int, [C.c_void_p, C.c_void_p]),
    "vtd_trocr_encode_crops": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_trocr_encode_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_trocr_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtd_trocr_num_slots": (C.c_int, [C.c_void_p]),
    "vtd_trocr_encode_crops_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_trocr_encode_pixels_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "vtd_trocr_stage_crops_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vtd_trocr_encode_staged_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "vtd_trocr_generate_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "vtd_trocr_last_steps": (C.c_int, [C.c_void_p]),
    "vtd_dbloss_workspace_bytes": (C.c_int64, []),
    "vtd_dbloss_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "vtd_stream_create_masked": (C.c_int, [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_void_p)]),
    "vtd_stream_destroy": (C.c_int, [C.c_void_p]),
    "vtd_trocr_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "vtd_trocr_get_option": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]),
    "vtd_trocr_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "vtd_trocr_get_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]),
    "vtd_trocr_get_gemm_profile": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_void_p]),
    "vtd_trocr_read_tap": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "vtd_trocr_encoder_tokens": (C.c_int, [C.c_void_p]),
    "vtd_trocr_logits_stride": (C.c_int, [C.c_void_p]),
    "vtd_trocr_macs_per_crop": (C.c_int64, [C.c_void_p]),
    "vtd_trocr_set_tuning": (C.c_int, [C.c_void_p, C.c_char_p]),
    "vtd_trocr_get_tuning": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_int64]),
    "vtd_trocr_tuning_measured": (C.c_int, [C.c_void_p]),
    "vtd_ctc_greedy_decode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
}


def load():
    """dlopen the library and bind every declared symbol (no device needed)."""
    global _lib
    if _lib is None:
        # torch first: it ships its own libamdhip64; loaded after ours the process ends up with two HIP runtimes and the
        # second one to initialise sees no device (observed with build() + smoke() in one process)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise NativeError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def require():
    """The library plus a visible GPU, or an exception.  Product code paths call this, never load()."""
    lib = load()
    n = lib.vtd_device_count()
    if n <= 0:
        raise NativeError("libvtd_hip.so loaded but no HIP device is visible; this package has no CPU fallback")
    return lib


def check(code, what=""):
    if code != 0:
        msg = load().vtd_strerror(code)
        raise NativeError(f"{what or 'vtd call'} failed: {msg.decode() if msg else code} ({code})")
