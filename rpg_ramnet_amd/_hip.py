"""ctypes binding of librpg_ramnet_hip.so (the C ABI declared in include/ramnet_hip.h).

The product path has NO CPU fallback: importing works without the library (so that `-m "not gpu"` host-logic
tests can run), but the first kernel call raises if the library is missing.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RAMNET_HIP_LIB") or os.path.join(_PKG, "librpg_ramnet_hip.so")     # (override: A/B builds)

IN_PLAIN, IN_CAT, IN_CAT_MUL, IN_UP2X, IN_UP2X_SKIP, IN_RELUMASK, IN_S2D, IN_PARITY4 = range(8)
ALGO_DIRECT, ALGO_WINOGRAD, ALGO_HEAD, ALGO_WINOGRAD24, ALGO_WINOGRAD_2X4, ALGO_WINOGRAD_2X4_SPLIT, ALGO_DIRECT_SPLIT = 0, 1, 2, 3, 4, 5, 6
EPI_LINEAR, EPI_RELU, EPI_SIGMOID, EPI_RES_RELU, EPI_GRU_BLEND, EPI_LSTM, EPI_GRU_BWD, EPI_SIGMOID_HR = range(8)

_fp = C.c_void_p


class ConvDesc(C.Structure):
    _fields_ = [
        ("x0", _fp), ("x1", _fp), ("xm", _fp),
        ("ld0", C.c_int), ("ld1", C.c_int), ("ldm", C.c_int),
        ("C0", C.c_int), ("C1", C.c_int), ("in_mode", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int),
        ("ntaps", C.c_int), ("stride", C.c_int),
        ("dy", C.c_int8 * 25), ("dx", C.c_int8 * 25), ("wtap", C.c_uint8 * 25),
        ("w", _fp), ("bias", _fp),
        ("Cout", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int), ("HoF", C.c_int), ("WoF", C.c_int),
        ("osy", C.c_int), ("osx", C.c_int), ("ooy", C.c_int), ("oox", C.c_int),
        ("epi", C.c_int), ("beta", C.c_float),
        ("e0", _fp), ("e1", _fp), ("lde0", C.c_int), ("lde1", C.c_int),
        ("out", _fp), ("o1", _fp), ("o2", _fp),
        ("ldo", C.c_int), ("ldo1", C.c_int), ("ldo2", C.c_int),
        ("algo", C.c_int),
        ("frame", C.c_int),
        ("out_s2d", C.c_int),
        ("head_cin", C.c_int),
        ("s2d_5x5", C.c_int),
        ("splitk_ws", C.c_void_p), ("splitk_floats", C.c_size_t),
    ]


class WgradSeg(C.Structure):
    _fields_ = [("x0", _fp), ("x1", _fp), ("xm", _fp), ("dout", _fp), ("gmask", _fp)]


WGRAD_MAX_SEGMENTS = 48


class WgradDesc(C.Structure):
    _fields_ = [
        ("x0", _fp), ("x1", _fp), ("xm", _fp),
        ("ld0", C.c_int), ("ld1", C.c_int), ("ldm", C.c_int),
        ("C0", C.c_int), ("C1", C.c_int), ("in_mode", C.c_int),
        ("B", C.c_int), ("Hin", C.c_int), ("Win", C.c_int),
        ("ntaps", C.c_int), ("stride", C.c_int),
        ("dy", C.c_int8 * 25), ("dx", C.c_int8 * 25),
        ("dout", _fp), ("gmask", _fp), ("ldg", C.c_int), ("ldgm", C.c_int),
        ("Cout", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("dw", _fp), ("dbias", _fp),
        ("algo", C.c_int),
        ("gsy", C.c_int), ("gsx", C.c_int), ("goy", C.c_int), ("gox", C.c_int), ("HoG", C.c_int), ("WoG", C.c_int),
        ("head_cin", C.c_int),
        ("dw_slabs", C.c_int),
        ("nseg", C.c_int), ("segs", C.POINTER(WgradSeg)),
    ]


_SIGS = {
    "ramnet_abi_version": (C.c_int, []),
    "ramnet_last_error": (C.c_char_p, []),
    "ramnet_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "ramnet_get_option": (C.c_int, [C.c_char_p]),
    "ramnet_last_kernel": (C.c_char_p, []),
    "ramnet_stream_fork": (C.c_int, [_fp, _fp]),
    "ramnet_gemm": (C.c_int, [_fp, _fp, _fp] + [C.c_int] * 9 + [C.c_long] * 3 + [_fp]),
    "ramnet_gemm2": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_long, C.c_long, C.c_long, C.c_int, _fp, _fp, _fp, C.c_int, C.c_long, C.c_long, C.c_long,
                               C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_nchw_to_nhwc_pad": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_reflect_pad": (C.c_int, [_fp, _fp] + [C.c_int] * 10 + [_fp]),
    "ramnet_wgrad_wino_slabs": (C.c_int, [C.c_int, C.c_int]),
    "ramnet_wgrad_wino2x4_slabs": (C.c_int, [C.c_int, C.c_int]),
    "ramnet_wgrad_dsplit_slabs": (C.c_int, [C.c_int, C.c_int]),
    "ramnet_wgrad_dsplit_ws_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ramnet_unpack_wgrad_dsplit": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_wgrad_wino2x4_ws_floats": (C.c_size_t, [C.c_int, C.c_int]),
    "ramnet_reduce_slabs": (C.c_int, [_fp, C.c_int, C.c_size_t, _fp]),
    "ramnet_packed_weight_elems": (C.c_size_t, [C.c_int] * 6),
    "ramnet_pack_weight": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_packed_weight_elems_wino": (C.c_size_t, [C.c_int] * 4),
    "ramnet_pack_weight_wino": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_packed_weight_elems_wino2x4": (C.c_size_t, [C.c_int] * 3),
    "ramnet_pack_weight_wino2x4": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_wino2x4_config": (C.c_int, [C.c_int]),
    "ramnet_conv_wino_variant": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "ramnet_conv_wino_split_ok": (C.c_int, [C.POINTER(ConvDesc), C.c_int]),
    "ramnet_packed_weight_elems_wino2x4_split": (C.c_size_t, [C.c_int] * 3),
    "ramnet_pack_weight_wino2x4_split": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_conv_splitk_floats": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "ramnet_packed_weight_elems_head": (C.c_size_t, [C.c_int]),
    "ramnet_head_supported": (C.c_int, [C.c_int, C.c_int]),
    "ramnet_pack_weight_head": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    "ramnet_fold_wino_supported": (C.c_int, [C.c_int, C.c_int]),
    "ramnet_packed_weight_elems_fold_wino": (C.c_size_t, [C.c_int, C.c_int]),
    "ramnet_pack_weight_fold_wino": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    "ramnet_pack_weight_fold_wino_dgrad": (C.c_int, [_fp, _fp, C.c_int, C.c_int, _fp]),
    "ramnet_pack_border_weights": (C.c_int, [_fp] * 5 + [C.c_int, C.c_int, _fp]),
    "ramnet_fold_unpack_wgrad": (C.c_int, [_fp] * 5 + [C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_unpad2_fold": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_up2x_border_col2im": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_pad2_sum": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_up2x_border_im2col": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_pad2_sum_im2col": (C.c_int, [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_space_to_depth2": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_frame_gather": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_unpack_wgrad": (C.c_int, [_fp, _fp] + [C.c_int] * 7 + [_fp]),
    "ramnet_unpack_wgrad_wino": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp]),
    "ramnet_unpack_wgrad_wino2x4": (C.c_int, [_fp, _fp] + [C.c_int] * 5 + [_fp]),
    "ramnet_conv_launch": (C.c_int, [C.POINTER(ConvDesc), _fp]),
    "ramnet_conv_launch_multi": (C.c_int, [C.POINTER(ConvDesc), C.c_int, _fp]),
    "ramnet_wgrad_launch": (C.c_int, [C.POINTER(WgradDesc), _fp]),
    "ramnet_pred_sigmoid_fwd": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_pred_sigmoid_bwd": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_pred_si_scratch_doubles": (C.c_size_t, [C.c_size_t, C.c_int]),
    "ramnet_pred_sigmoid_si_fwd": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.c_float, C.c_float,
                                             _fp, _fp, _fp, _fp]),
    "ramnet_pred_sigmoid_si_bwd": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), _fp, _fp, C.c_float,
                                             C.c_float, _fp, C.c_int, _fp, _fp, _fp, C.c_int, _fp]),
    "ramnet_pred_linear_fwd": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_pred_linear_bwd": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_cat_batch_add": (C.c_int, [_fp, C.c_int, C.c_size_t, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ramnet_cat_batch_add_masked": (C.c_int, [_fp, C.c_int, C.c_size_t, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "ramnet_relu_bwd": (C.c_int, [_fp, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_upsample2x_bwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_gru_bwd_a": (C.c_int, [_fp] * 7 + [C.c_size_t, C.c_int, C.c_int, _fp]),
    "ramnet_gru_bwd_a2": (C.c_int, [_fp] * 7 + [C.c_size_t, C.c_int, C.c_int, C.c_int, _fp]),
    "ramnet_gru_bwd_b": (C.c_int, [_fp] * 5 + [C.c_size_t, C.c_int, _fp]),
    "ramnet_lstm_bwd": (C.c_int, [_fp] * 7 + [C.c_size_t, C.c_int, _fp]),
    "ramnet_norm_slabs": (C.c_int, [C.c_int, C.c_long, C.c_int]),
    "ramnet_norm_partial": (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, _fp, _fp]),
    "ramnet_norm_finalize": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_long, C.c_double, _fp, _fp, _fp, _fp, C.c_double, C.c_int, C.c_int,
                                       _fp, _fp, _fp, _fp, _fp, _fp]),
    "ramnet_norm_finalize_bwd": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_long, _fp, _fp, _fp, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]),
    "ramnet_norm_apply": (C.c_int, [_fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_long, C.c_int, _fp]),
    "ramnet_norm_bwd": (C.c_int, [_fp, C.c_int, _fp, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp, C.c_int,
                                  C.c_int, C.c_long, C.c_int, _fp]),
    "ramnet_add": (C.c_int, [_fp, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_split2": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp]),
    "ramnet_concat2": (C.c_int, [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp, C.c_size_t, _fp]),
    "ramnet_bias_grad": (C.c_int, [_fp, _fp, _fp, C.c_size_t, C.c_int, _fp]),
    "ramnet_si_loss_fwd": (C.c_int, [_fp, _fp, C.c_size_t, C.c_float, C.c_float, _fp, _fp, _fp]),
    "ramnet_si_loss_from_stats": (C.c_int, [_fp, C.c_float, C.c_float, _fp, _fp]),
    "ramnet_si_loss_bwd": (C.c_int, [_fp, _fp, C.c_size_t, C.c_float, C.c_float, _fp, _fp, _fp, _fp]),
    "ramnet_si_log_loss_fwd": (C.c_int, [_fp, _fp, C.c_size_t, C.c_float, _fp, _fp, _fp]),
    "ramnet_si_log_loss_bwd": (C.c_int, [_fp, _fp, C.c_size_t, C.c_float, _fp, _fp, _fp, _fp]),
    "ramnet_mse_loss_fwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ramnet_mse_loss_bwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "ramnet_depth_metrics": (C.c_int, [_fp, _fp, C.c_size_t, C.c_float, C.c_float, C.c_float, _fp, _fp]),
    "ramnet_msg_workspace_elems": (C.c_size_t, [C.c_int] * 4),
    "ramnet_msg_loss_fwd": (C.c_int, [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp]),
    "ramnet_msg_loss_bwd": (C.c_int, [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ramnet_voxelize": (C.c_int, [_fp, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ramnet_voxel_indices": (C.c_int, [_fp, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]),
    "ramnet_normalize_nonzero": (C.c_int, [_fp, C.c_size_t, _fp, _fp]),
    "ramnet_voxelize_batch": (C.c_int, [_fp, _fp, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, _fp, _fp]),
    "ramnet_normalize_nonzero_batch": (C.c_int, [_fp, C.c_int, C.c_size_t, _fp, _fp]),
}
EXPORTS = tuple(_SIGS)

_lib = None
_tracer = None          # profiling hook: tracer(name, fn, args) -> result, for every ramnet_* call (bench.py); None = direct calls


class _Traced:
    def __init__(self, l):
        self._l = l

    def __getattr__(self, name):
        fn = getattr(self._l, name)
        tr = _tracer
        if tr is None:
            return fn
        return lambda *a: tr(name, fn, a)


def set_tracer(fn):
    """Route every library call through fn(name, c_function, args) (None restores direct calls)."""
    global _tracer
    _tracer = fn


class HipLibraryMissing(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). rpg_ramnet_amd has no CPU fallback." % LIB_PATH)
        # bind to the SAME HIP runtime torch uses (torch ships its own libamdhip64.so.7): load torch's copy first so
        # that our DT_NEEDED entry resolves to it instead of a second runtime with no initialised device
        import torch
        rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(rt):
            C.CDLL(rt, mode=C.RTLD_GLOBAL)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.ramnet_abi_version() != 24:
            raise RuntimeError("ABI version mismatch in %s" % LIB_PATH)
        _lib = l
    return _lib if _tracer is None else _Traced(_lib)


def check(code, what):
    if code != 0:
        msg = lib().ramnet_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, code, msg.decode() if msg else "?"))
