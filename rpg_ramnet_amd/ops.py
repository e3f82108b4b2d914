"""Autograd operators of the RAM-Net hot path; every forward AND backward is a HIP kernel launch
through the C ABI (include/ramnet_hip.h).  torch supplies device memory, streams and the autograd graph.

Activations are NHWC tensors [B, H, W, C] (fp32).  Weight gradients are accumulated by the weight-gradient
kernel into per-layer [tap][Cin][Cout] workspaces over ALL time steps of a BPTT backward pass and folded into
``param.grad`` (OIHW) once, by a callback queued on the autograd engine (the reference gets the same sums from
autograd's per-use accumulation, lstm_trainer.py:450).
"""
import contextlib
import ctypes as C
import os as _os

import torch
from torch.autograd import Function

from . import _hip as H

_NULL = None
_ABL_SKIP_WGRAD = _os.environ.get("RAMNET_ABL_SKIP_WGRAD") == "1"
if _ABL_SKIP_WGRAD:
    import warnings as _warnings
    _warnings.warn("RAMNET_ABL_SKIP_WGRAD=1: backward-weights launches are SKIPPED — every weight gradient of this process is zero "
                   "(tuning runs only; unset the variable for anything else)", RuntimeWarning)

# 3x3 stride-1 layers (ConvGRU gates / candidate, residual blocks) run Winograd F(2x2,3x3) in fp32 — 2.25x fewer
# multiplies, rounding error ~1e-6 relative (tests/test_hip_ops.py) — unless switched off (RAMNET_WINOGRAD=0).
_WINOGRAD = True
_WINO_MIN_CIN = 32        # smallest reduction depth that uses the Winograd kernels


def set_winograd(on):
    global _WINOGRAD
    _WINOGRAD = bool(on)


_WINO_2X4 = "auto"
_WGRAD_SLABS = True
_WGRAD_2X4 = "auto"       # F(2x4,3x3) backward-weights (csrc/conv_wgrad_wino6.hip): "auto" (with set_wgrad_overlap(True)) | "off" | "force"


def set_wgrad_slabs(on):
    """Winograd backward-weights: tile splits join per-split slabs of the workspace by plain read-modify-write (default: no atomics,
    bit-reproducible gradients) or meet in slab 0 by atomic adds (A/B runs)."""
    global _WGRAD_SLABS
    _WGRAD_SLABS = bool(on)



def set_winograd_2x4(mode, min_wgs=None):
    """F(2x4,3x3) variant of the Winograd forward / backward-data launches (csrc/conv_wino6.hip): "auto" = where the library's size
    heuristics pick it (large maps: the two fine scales at the training batch), "off" = F(2x2,3x3) everywhere, "force" = every
    structurally eligible launch (tests).  min_wgs: launch-size threshold of "auto" (library default when None)."""
    global _WINO_2X4
    assert mode in ("auto", "off", "force")
    _WINO_2X4 = mode
    if min_wgs is not None:
        H.check(H.lib().ramnet_wino2x4_config(int(min_wgs)), "wino2x4_config")
        invalidate_descs()


def get_winograd():
    return _WINOGRAD


# UpsampleConvLayer forward as four 4x4 parity convolutions of the low-resolution input (16 instead of 25 MACs per output,
# no interpolation in the loader) plus small border-correction GEMMs.  RAMNET_FOLD_UPSAMPLE=0 /
# set_fold_upsample(False) keeps the direct 5x5 kernel with the bilinear loader.
_FOLD_UP = True


_FOLD_WINO = True
_FOLD_WINO_MIN_COUT = 32
_FOLD_WINO_WGRAD = True
_SAVE_XPAD = False         # keep pad2(x + skip) of a folded decoder layer for its backward: +2 GB, no measurable gain


def set_fold_winograd_wgrad(on):
    """Folded upsample-conv backward-weights in the Winograd F(2x2,4x4) domain (32 x 64 or 64 x 32 channel workgroups) or as four
    direct 16-tap parity launches."""
    global _FOLD_WINO_WGRAD
    _FOLD_WINO_WGRAD = bool(on)


def set_fold_pair(on):
    """32-channel folded decoders in the pair form (default) or as 64 tiles x 32 channels; the library option and the packs follow."""
    H.check(H.lib().ramnet_set_option(b"fold_pair", int(bool(on))), "set_option")
    invalidate_packs()
    invalidate_descs()


_GRU_BWD_FUSED = True


def set_gru_bwd_fused(on):
    """Stage B of the ConvGRU backward (ramnet_gru_bwd_b) in the epilogue of the candidate convolution's backward-data launch
    (RAMNET_EPI_GRU_BWD; hidden sizes that are multiples of 64): on by default, off for A/B runs and the unfused path's tests."""
    global _GRU_BWD_FUSED
    _GRU_BWD_FUSED = bool(on)


_GRU_HR = _os.environ.get("RAMNET_GRU_HR", "1") != "0"


def set_gru_materialize_hr(on):
    """The ConvGRU gates launch also writes h.r (RAMNET_EPI_SIGMOID_HR): the candidate convolution and its backward-weights launch then read
    the plain concatenation [x | h.r] instead of forming the product in their loaders (one more state-sized tensor per update, saved for the
    backward; the same single-rounded products either way).  On by default; off for A/B runs and the loader path's tests."""
    global _GRU_HR
    _GRU_HR = bool(on)


_SPLIT_OPERANDS = False


def set_split_operands(on):
    """F(2x4,3x3) forward / backward-data launches with three-term bf16 splits of both operands on the bf16 matrix pipe (RAMNET_ALGO_WINOGRAD_2X4_SPLIT,
    csrc/conv_wino6s.hip: six of the nine partial products, fp32 accumulation — fp32-level accuracy, the parity suites run it at unchanged
    tolerances) instead of the exact-fp32 MFMA.  Off by default: the headline arithmetic is exact fp32."""
    global _SPLIT_OPERANDS
    _SPLIT_OPERANDS = bool(on)
    invalidate_descs()


def get_split_operands():
    return _SPLIT_OPERANDS


_SPLIT_WGRAD = _os.environ.get("RAMNET_SPLIT_WGRAD", "0") == "1"


def set_split_wgrad(on):
    """With split operands on (set_split_operands): the backward-weights launches of the layers that would run F(2x4,3x3) run the DIRECT
    3x3 form on the bf16 matrix pipe instead (RAMNET_ALGO_DIRECT_SPLIT, csrc/conv_wgrad_dsplit.hip: operands split once per element when a
    strip is staged, no Winograd transform in the loop).  OFF by default — measured (profiles/r06_dsplit_notes.md): the six backward-weights
    launches of a ConvGRU update 1.19 ms against 1.20 ms for the exact-fp32 F(2x4) kernel on their own, and the co-scheduled training step
    LOSES 7 % (237 against 255 samples/s with split forward / backward-data launches): kept as a correct, tested algorithm and a record of
    why its loop does not reach the matrix pipe's rate.  Takes effect at the next backward pass."""
    global _SPLIT_WGRAD
    _SPLIT_WGRAD = bool(on)


def get_split_wgrad():
    return _SPLIT_WGRAD


def set_wgrad_winograd_2x4(mode):
    """F(2x4,3x3) backward-weights for the plain 3x3 layers (ConvGRU / ConvLSTM / residual layers of >= 64 reduction channels): "auto"
    (default) = when the backward-weights launches are co-scheduled with the backward-data chain (set_wgrad_overlap(True): training step
    +1.1 % same-box; in stream order the 32 x 32-channel F(2x2) workgroups are faster: 200.4 vs 196.5 samples/s), "force" = always (tests),
    "off" = F(2x2,3x3) everywhere.  Takes effect at the next backward pass (a pass uses one workspace layout per layer)."""
    global _WGRAD_2X4
    assert mode in ("auto", "off", "force")
    _WGRAD_2X4 = mode


def set_winograd_split(on):
    """Split channel reduction of latency-bound F(2x2,3x3) launches (batch-1 streaming on the coarse scales; csrc/conv_wino.hip): on by
    default, off for A/B runs.  Descriptors are built per launch, graphs captured before the call keep what they captured."""
    H.check(H.lib().ramnet_set_option(b"wino_ksplit", int(on)), "set_option")         # (2..16: that many splits, tuning runs)
    invalidate_descs()


def _fold_pair(Cout, Cin):
    """32-channel layers (the last decoder): both column parities of a row parity in one 64-column workgroup that shares the
    transformed input (conv_wino24_kernel<.., PAIR>); RAMNET_FOLD_PAIR=0 keeps the 64-tile x 32-channel form."""
    return Cout == 32 and Cin % 32 == 0 and bool(H.lib().ramnet_get_option(b"fold_pair"))


def _fold_wino_ok(Cin, Cout):
    """conv_wino24_kernel shapes: 64-column workgroups with chunks of 16, else 32-channel ones with chunks of 8; an even
    number of chunks."""
    kc = 16 if ((Cout % 64 == 0 and Cin % 16 == 0) or _fold_pair(Cout, Cin)) else 8
    return Cout % 32 == 0 and Cout >= _FOLD_WINO_MIN_COUT and Cin % (2 * kc) == 0


def set_fold_winograd(on):
    """Folded upsample-conv forward on the Winograd F(2x2,4x4) kernel (shapes: _fold_wino_ok) or the direct one."""
    global _FOLD_WINO
    _FOLD_WINO = bool(on)


def set_fold_upsample(on):
    global _FOLD_UP
    _FOLD_UP = bool(on)


def get_fold_upsample():
    return _FOLD_UP


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _st():
    """The current HIP stream of the current device as a void* — through torch's C entry points (0.3 us): torch.cuda.current_stream() builds
    a Stream object behind three Python wrappers (~10 us, ~600 calls per training step on the launch path)."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, off=0):
    return None if t is None else C.c_void_p(t.data_ptr() + 4 * off)


def dense(t):
    """NHWC tensor usable by the kernels as (ptr, ld): unit channel stride, 16-byte aligned pixels."""
    if t is None:
        return None
    B, Hh, W, Cc = t.shape
    s = t.stride()
    ok = (s[3] == 1 or Cc == 1) and s[2] % 4 == 0 and s[1] == W * s[2] and s[0] == Hh * s[1] and t.data_ptr() % 16 == 0
    if ok and t.dtype == torch.float32:
        return t
    return t.contiguous().float()


def ld(t):
    return t.stride(2)


# ------------------------------------------------------------------------------------------------ tap lists
class Taps:
    """(dy, dx, weight-slice) triples of one launch, pre-converted to ctypes arrays."""
    _cache = {}

    def __init__(self, triples):
        self.n = len(triples)
        assert 1 <= self.n <= 25
        self.dy = (C.c_int8 * 25)(*[t[0] for t in triples])
        self.dx = (C.c_int8 * 25)(*[t[1] for t in triples])
        self.wt = (C.c_uint8 * 25)(*[t[2] for t in triples])
        self.wino = False
        self.head = False                # dense padded 5x5 forward window (head kernel eligible)
        self.flop_taps = self.n          # taps of the convolution this launch stands for (algorithmic FLOP accounting)

    @classmethod
    def get(cls, kind, k, pad, py=0, px=0):
        key = (kind, k, pad, py, px)
        if key not in cls._cache:
            if kind in ("conv", "conv_s2d"):        # forward taps: in(o*s + kh - pad)
                tr = [(kh - pad, kw - pad, kh * k + kw) for kh in range(k) for kw in range(k)]
            elif kind in ("dgrad1", "dgrad1_s2d"):  # backward-data of a stride-1 conv: dy(o + pad - kh)
                tr = [(pad - kh, pad - kw, kh * k + kw) for kh in range(k) for kw in range(k)]
            elif kind == "dgrad2":    # backward-data of a stride-2 conv, output parity class (py, px)
                tr = [((py + pad - kh) // 2, (px + pad - kw) // 2, kh * k + kw)
                      for kh in range(k) for kw in range(k)
                      if (py + pad - kh) % 2 == 0 and (px + pad - kw) % 2 == 0]
            elif kind == "fold":      # parity class (py, px) of the folded upsample-conv on the replicate-padded (by 2) low-res
                tr = [(py + ty, px + tx, (py * 2 + px) * 16 + ty * 4 + tx) for ty in range(4) for tx in range(4)]   # input
            else:
                raise KeyError(kind)
            cls._cache[key] = cls(tr)
            cls._cache[key].wino = kind in ("conv", "dgrad1", "conv_s2d", "dgrad1_s2d") and k == 3 and pad == 1   # dense padded 3x3
            cls._cache[key].head = kind == "conv" and k == 5 and pad == 2
            if kind.endswith("_s2d"):     # 3x3 over 4*Cin space-to-depth channels standing for a 5x5 stride-2 layer over Cin
                cls._cache[key].flop_taps = 25.0 / 4.0
            if kind == "fold":          # one output parity of a 5x5 convolution after the x2 upsample
                cls._cache[key].flop_taps = 25
        return cls._cache[key]


def conv_launch(x0, taps, w, out, Cout, **kw):
    d = _conv_desc(x0, taps, w, out, Cout, **kw)
    H.check(H.lib().ramnet_conv_launch(C.byref(d), _st()), "ramnet_conv_launch")


def conv_launch_multi(x0, w, out, Cout, classes, **kw):
    """One launch for several output classes: `classes` = [(taps, Ho, Wo, (osy, osx, ooy, oox)), ...] (<= 4)."""
    arr = (H.ConvDesc * len(classes))()
    if isinstance(w, PackRef):
        w = w.cp.pack(w.transposed, False)
    for i, (taps, Ho, Wo, os_) in enumerate(classes):
        arr[i] = _conv_desc(x0, taps, w, out, Cout, Ho=Ho, Wo=Wo, os=os_, **kw)
    H.check(H.lib().ramnet_conv_launch_multi(arr, len(classes), _st()), "ramnet_conv_launch_multi")


def uses_winograd(taps, w, stride, epi, in_mode, C0, C1):
    """Does this forward / backward-data launch run the Winograd F(2x2,3x3) kernel?  (3x3 stride-1 window, fp32, no
    upsampling loader, >= 32 reduction channels, concatenation boundary on a chunk of 8; the ConvLSTM cell epilogue
    needs a hidden size that is a multiple of 4.)"""
    return bool(isinstance(w, PackRef) and _WINOGRAD and taps.wino and stride == 1
                and (w.cp.gates == 1 or w.transposed or (epi == H.EPI_LSTM and w.cp.Cout % 16 == 0))
                and (epi != H.EPI_LSTM or w.cp.gates == 4) and in_mode not in (H.IN_UP2X, H.IN_UP2X_SKIP)
                and C0 + C1 >= _WINO_MIN_CIN and (C1 == 0 or C0 % 8 == 0))


# The two 5x5 head layers (1 / 5 real input channels -> 32 maps at full resolution) on their own kernel:
# dense (tap, channel) reduction with the weights in registers.  RAMNET_HEAD_KERNEL=0 / set_head_kernel(False): generic kernel.
_HEAD = True


def set_head_kernel(on):
    global _HEAD
    _HEAD = bool(on)


def get_head_kernel():
    return _HEAD


def uses_head(taps, w, stride, epi, in_mode):
    """Does this forward launch run the head kernel?  (dense 5x5 stride-1 window, fp32, plain input, 1/3/5 real input
    channels, <= 32 outputs, bias + optional ReLU only.)"""
    return bool(isinstance(w, PackRef) and _HEAD and taps.head and stride == 1 and not w.transposed
                and w.cp.gates == 1 and len(w.cp.weights) == 1 and in_mode == H.IN_PLAIN and epi in (H.EPI_LINEAR, H.EPI_RELU)
                and H.lib().ramnet_head_supported(w.cp.Cin, w.cp.Cout))


# Launch descriptors are cached per call site (round 5): everything of a ramnet_conv_desc but its pointers is a function of the layer,
# the shapes / strides and the kernel-selection switches — filling ~45 ctypes fields and asking the library twice (F(2x4) variant, split
# reduction) cost ~17 us per launch, ~1100 launches per training step on the two host threads.  A hit copies the template and sets the
# pointers (~5 us).  The key holds every argument that is not a tensor's address, the strides, the 16-byte alignment of the epilogue
# operands (the library's variant choice looks at it) and the switches; _DESC_EPOCH covers library-side options.
_DESC_CACHE = {}
_DESC_EPOCH = 0
_DESC_CACHE_ON = _os.environ.get("RAMNET_DESC_CACHE", "1") != "0"


def invalidate_descs():
    global _DESC_EPOCH
    _DESC_EPOCH += 1
    _DESC_CACHE.clear()
    _WDESC_CACHE.clear()


def _sd(t):
    return -1 if t is None else t.stride(2)


def _conv_desc(x0, taps, w, out, Cout, *, stride=1, x1=None, xm=None, xm_off=0, in_mode=H.IN_PLAIN,
               C0=None, C1=0, Hin=None, Win=None, bias=None, epi=H.EPI_LINEAR, beta=0.0, e0=None, e1=None,
               o1=None, o2=None, Ho=None, Wo=None, os=(1, 1, 0, 0), out_off=0, frame=0, out_s2d=0, wino24=False, ws_owner=None):
    is_ref = isinstance(w, PackRef)
    al = out.data_ptr() | (4 * out_off)
    for t in (e0, e1, o1, o2, bias):
        if t is not None:
            al |= t.data_ptr()
    key = (id(taps), id(w.cp) if is_ref else 0, w.transposed if is_ref else -1, id(ws_owner), Cout, stride, in_mode, C0, C1, Hin, Win, epi, beta, Ho,
           Wo, os, out_off, frame, out_s2d, wino24, xm_off, tuple(x0.shape), x0.stride(2), _sd(x1), _sd(xm), out.shape[1], out.shape[2],
           out.stride(2), _sd(e0), _sd(e1), _sd(o1), _sd(o2), bias is None, al & 15, _WINOGRAD, _WINO_2X4, _SPLIT_OPERANDS, _HEAD, _S2D_SPARSE, _S2D_2X4,
           _DESC_EPOCH, x0.device.index)
    hit = _DESC_CACHE.get(key) if _DESC_CACHE_ON else None
    if hit is not None and (not is_ref or hit[3]() is w.cp) and (ws_owner is None or hit[4]() is ws_owner):
        tmpl, kind, nsplit = hit[0], hit[1], hit[2]
        d = H.ConvDesc.from_buffer_copy(tmpl)
        d.x0, d.x1, d.xm = _p(x0), _p(x1), _p(xm, xm_off)
        d.bias, d.e0, d.e1 = _p(bias), _p(e0), _p(e1)
        d.out, d.o1, d.o2 = _p(out, out_off), _p(o1), _p(o2)
        d.w = _p(w) if kind is None else _p(w.cp.pack(0, "head") if kind == "head" else w.cp.pack(w.transposed, kind))
        if nsplit:
            d.splitk_ws = _p((w.cp if is_ref else ws_owner).splitk_ws(nsplit, x0.device))
        return d
    meta = []
    d = _conv_desc_build(x0, taps, w, out, Cout, meta, stride=stride, x1=x1, xm=xm, xm_off=xm_off, in_mode=in_mode, C0=C0, C1=C1, Hin=Hin, Win=Win,
                         bias=bias, epi=epi, beta=beta, e0=e0, e1=e1, o1=o1, o2=o2, Ho=Ho, Wo=Wo, os=os, out_off=out_off, frame=frame,
                         out_s2d=out_s2d, wino24=wino24, ws_owner=ws_owner)
    import weakref
    if len(_DESC_CACHE) > 4096:
        _DESC_CACHE.clear()
    _DESC_CACHE[key] = (H.ConvDesc.from_buffer_copy(d), meta[0], meta[1], weakref.ref(w.cp) if is_ref else None,
                        weakref.ref(ws_owner) if ws_owner is not None else None)
    return d


def _conv_desc_build(x0, taps, w, out, Cout, meta, *, stride=1, x1=None, xm=None, xm_off=0, in_mode=H.IN_PLAIN,
                     C0=None, C1=0, Hin=None, Win=None, bias=None, epi=H.EPI_LINEAR, beta=0.0, e0=None, e1=None,
                     o1=None, o2=None, Ho=None, Wo=None, os=(1, 1, 0, 0), out_off=0, frame=0, out_s2d=0, wino24=False, ws_owner=None):
    """The descriptor from scratch; meta <- [which pack of the layer d.w points at (None: `w` is a packed tensor itself; "head"; False /
    True / "2x4": ConvParam.pack(transposed, kind)), floats of the split-reduction workspace (0: none)]."""
    B = x0.shape[0]
    kind, nsplit = None, 0
    d = H.ConvDesc()
    d.x0, d.x1, d.xm = _p(x0), _p(x1), _p(xm, xm_off)
    d.ld0, d.ld1, d.ldm = ld(x0), (ld(x1) if x1 is not None else 0), (ld(xm) if xm is not None else 0)
    d.C0, d.C1, d.in_mode = (x0.shape[3] if C0 is None else C0), C1, in_mode
    d.algo = H.ALGO_DIRECT
    ref = None
    cp = w.cp if isinstance(w, PackRef) else ws_owner            # the layer that owns the split-reduction workspace
    if isinstance(w, PackRef) and beta == 0.0 and frame == 0 and os == (1, 1, 0, 0) and uses_head(taps, w, stride, epi, in_mode):
        d.algo, d.head_cin = H.ALGO_HEAD, w.cp.Cin
        kind = "head"
        w = w.cp.pack(0, "head")
    elif isinstance(w, PackRef):
        wino = uses_winograd(taps, w, stride, epi, in_mode, d.C0, C1)
        d.algo = H.ALGO_WINOGRAD if wino else H.ALGO_DIRECT
        # the 3x3 view of a 5x5 stride-2 layer read / written in place: 11 of its 36 slices are zero by construction (s2d_weights)
        d.s2d_5x5 = int(wino and isinstance(w.cp, S2DConvParam) and _S2D_SPARSE and (in_mode == H.IN_S2D or out_s2d > 0))
        if wino and w.cp.gates == 1 and os == (1, 1, 0, 0) and (_S2D_2X4 or not isinstance(w.cp, S2DConvParam)):
            ref = w                      # candidate for F(2x4,3x3): the library decides once the descriptor is complete (below)
        kind = bool(wino)
        w = w.cp.pack(w.transposed, wino)
    d.B, d.Hin, d.Win = B, (x0.shape[1] if Hin is None else Hin), (x0.shape[2] if Win is None else Win)
    d.ntaps, d.stride = taps.n, stride
    d.dy, d.dx, d.wtap = taps.dy, taps.dx, taps.wt
    d.w, d.bias = _p(w), _p(bias)
    d.Cout = Cout
    d.HoF, d.WoF = out.shape[1], out.shape[2]
    d.Ho, d.Wo = (d.HoF if Ho is None else Ho), (d.WoF if Wo is None else Wo)
    d.osy, d.osx, d.ooy, d.oox = os
    d.epi, d.beta, d.frame, d.out_s2d = epi, beta, frame, out_s2d
    d.e0, d.e1 = _p(e0), _p(e1)
    d.lde0, d.lde1 = (ld(e0) if e0 is not None else 0), (ld(e1) if e1 is not None else 0)
    d.out, d.o1, d.o2 = _p(out, out_off), _p(o1), _p(o2)
    d.ldo, d.ldo1, d.ldo2 = ld(out), (ld(o1) if o1 is not None else 0), (ld(o2) if o2 is not None else 0)
    if wino24:          # w = ConvParam.pack_fold_wino(): all four parities in one Winograd F(2x2,4x4) launch
        d.algo = H.ALGO_WINOGRAD24
    if ref is not None and _WINO_2X4 != "off" and H.lib().ramnet_conv_wino_variant(C.byref(d), int(_WINO_2X4 == "force")):
        # F(2x4,3x3) on the fine scales (csrc/conv_wino6.hip): its own Winograd-domain pack of the same parameters
        d.algo, d.s2d_5x5 = H.ALGO_WINOGRAD_2X4, 0      # (dense: the F(2x4) kernel skips no zero slices)
        kind = "2x4"
        if _SPLIT_OPERANDS and H.lib().ramnet_conv_wino_split_ok(C.byref(d), int(_WINO_2X4 == "force")):
            d.algo, kind = H.ALGO_WINOGRAD_2X4_SPLIT, "2x4s"       # the same launch on the bf16 matrix pipe, split operands (csrc/conv_wino6s.hip)
        d.w = _p(ref.cp.pack(ref.transposed, kind))
    if d.algo in (H.ALGO_WINOGRAD, H.ALGO_WINOGRAD24) and cp is not None:
        # latency-bound launches (batch-1 streaming on the coarse scales) split their channel reduction: the library says how much
        # workspace the launch would use, the layer owns it (csrc/conv_wino.hip, ramnet_conv_desc.splitk_ws)
        n = H.lib().ramnet_conv_splitk_floats(C.byref(d))
        if n:
            d.splitk_ws, d.splitk_floats = _p(cp.splitk_ws(n, x0.device)), n
            nsplit = n
    meta[:] = [kind, nsplit]
    return d


_WDESC_CACHE = {}


def _wgrad_desc_build(x0, taps, dout, dw, Cout, stride, x1, xm, xm_off, in_mode, C0, C1, Hin, Win, gmask, dbias, Ho, Wo, gview, dw_off, wino24):
    d = H.WgradDesc()
    d.x0, d.x1, d.xm = _p(x0), _p(x1), _p(xm, xm_off)
    d.ld0, d.ld1, d.ldm = ld(x0), (ld(x1) if x1 is not None else 0), (ld(xm) if xm is not None else 0)
    d.C0, d.C1, d.in_mode = (x0.shape[3] if C0 is None else C0), C1, in_mode
    d.B, d.Hin, d.Win = x0.shape[0], (x0.shape[1] if Hin is None else Hin), (x0.shape[2] if Win is None else Win)
    d.ntaps, d.stride = taps.n, stride
    d.dy, d.dx = taps.dy, taps.dx
    d.dout, d.gmask = _p(dout), _p(gmask)
    d.ldg, d.ldgm = ld(dout), (ld(gmask) if gmask is not None else 0)
    d.Cout, d.Ho, d.Wo = Cout, (dout.shape[1] if Ho is None else Ho), (dout.shape[2] if Wo is None else Wo)
    d.dw, d.dbias = _p(dw, dw_off), _p(dbias)
    if gview is not None:      # dout / gmask addressed as (oy*gsy + goy, ox*gsx + gox) of [B, HoG, WoG]
        d.gsy, d.gsx, d.goy, d.gox, d.HoG, d.WoG = gview
    d.algo = H.ALGO_WINOGRAD if getattr(dw, "wino", False) else H.ALGO_DIRECT      # set by ConvParam.grad_ws()
    if getattr(dw, "wino6", False):          # F(2x4,3x3): dw = [slabs][24][Cin][Cout] (csrc/conv_wgrad_wino6.hip)
        d.algo = H.ALGO_WINOGRAD_2X4
    if getattr(dw, "wg_dsplit", False):         # direct 3x3, split bf16 operands: dw = [slabs][9][Cin][Cout] (csrc/conv_wgrad_dsplit.hip)
        d.algo = H.ALGO_DIRECT_SPLIT
    d.dw_slabs = getattr(dw, "slabs", 0) if (d.algo in (H.ALGO_WINOGRAD, H.ALGO_WINOGRAD_2X4, H.ALGO_DIRECT_SPLIT) and _WGRAD_SLABS) else 0     # per-split slabs (read-modify-write joins)
    if wino24:          # folded upsample-conv in the Winograd F(2x2,4x4) domain: dw = [4][25][C0][Cout]
        d.algo = H.ALGO_WINOGRAD24
    hc = getattr(dw, "head_cin", 0)
    if hc and _HEAD and taps.head and stride == 1 and in_mode == H.IN_PLAIN and gview is None and dw_off == 0:
        d.algo, d.head_cin = H.ALGO_HEAD, hc
    if d.algo in (H.ALGO_WINOGRAD, H.ALGO_WINOGRAD_2X4) and C1 and d.C0 % 32:
        raise RuntimeError("Winograd backward-weights needs the concatenation boundary at a multiple of 32 channels")
    return d


def wgrad_launch(x0, taps, dout, dw, Cout, *, stride=1, x1=None, xm=None, xm_off=0, in_mode=H.IN_PLAIN, C0=None,
                 C1=0, Hin=None, Win=None, gmask=None, dbias=None, Ho=None, Wo=None, gview=None, dw_off=0, wino24=False, segs=None, stream=None):
    """segs: a ctypes array of H.WgradSeg (ramnet_wgrad_desc.segs) — the tensors of several launches of the SAME layer and shape reduced
    in one launch (deferred ConvGRU cell updates, _wgrad_cell); x0 ... gmask then describe the first segment."""
    # (descriptor cache as for _conv_desc: the template holds everything but the six pointers)
    key = (id(taps), id(dw), getattr(dw, "wino", False), getattr(dw, "wino6", False), getattr(dw, "wg_dsplit", False), getattr(dw, "slabs", 0), getattr(dw, "head_cin", 0), Cout, stride,
           in_mode, C0, C1, Hin, Win, Ho, Wo, gview, dw_off, wino24, xm_off, tuple(x0.shape), x0.stride(2), _sd(x1), _sd(xm), dout.shape[1],
           dout.shape[2], dout.stride(2), _sd(gmask), _WGRAD_SLABS, _HEAD, x0.device.index)
    tmpl = _WDESC_CACHE.get(key) if _DESC_CACHE_ON else None
    if tmpl is not None:
        d = H.WgradDesc.from_buffer_copy(tmpl)
        d.x0, d.x1, d.xm = _p(x0), _p(x1), _p(xm, xm_off)
        d.dout, d.gmask = _p(dout), _p(gmask)
        d.dw, d.dbias = _p(dw, dw_off), _p(dbias)
    else:
        d = _wgrad_desc_build(x0, taps, dout, dw, Cout, stride, x1, xm, xm_off, in_mode, C0, C1, Hin, Win, gmask, dbias, Ho, Wo, gview, dw_off, wino24)
        if len(_WDESC_CACHE) > 4096:
            _WDESC_CACHE.clear()
        _WDESC_CACHE[key] = H.WgradDesc.from_buffer_copy(d)
    if segs is not None:
        d.nseg, d.segs = len(segs), C.cast(segs, C.POINTER(H.WgradSeg))
    if _ABL_SKIP_WGRAD:        # tuning runs only (RAMNET_ABL_SKIP_WGRAD=1: how much of the step the backward-weights launches cost; gradients are WRONG)
        return
    H.check(H.lib().ramnet_wgrad_launch(C.byref(d), _st() if stream is None else stream), "ramnet_wgrad_launch")


# ------------------------------------------------------------------------------------------------ side stream
# Backward-weights only feeds the gradient workspaces, never the backward-data chain, so it runs on a second HIP stream:
# the hardware co-schedules its workgroups with the backward-data kernel of the main stream and fills the CUs that tile
# quantisation / barrier stalls of one kernel leave idle.  Ordering: side waits for an event recorded on the main stream
# (its inputs are ready), the end-of-backward fold waits for the side stream; inputs are record_stream()'ed so the
# caching allocator does not recycle them while the side stream still reads them.
_SIDE = {}
_USE_SIDE = False


def set_wgrad_overlap(on):
    """Run backward-weights kernels on a side stream, co-scheduled with the backward-data chain (+4-6 % step rate on
    MI355X; per-kernel timings then include co-scheduled time, so profiling runs keep it off)."""
    global _USE_SIDE
    _USE_SIDE = bool(on)
    # the Winograd backward-weights kernel in the shape that suits the schedule: 32 x 64-channel workgroups (two per CU) share the CUs
    # with the backward-data chain of the other stream; 32 x 32-channel workgroups (168 registers: three per CU) are 13-20 % faster on
    # their own and 3 % slower in the co-scheduled step, where three of them leave a CU no room for a backward-data workgroup
    H.check(H.lib().ramnet_set_option(b"wgrad_wino_nf", 2 if _USE_SIDE else 1), "set_option")


# The decoder of update k (prediction for that measurement) and the state update k+1 both depend only on the state after
# update k: with set_decoder_overlap(True) the model runs its decoders on a second stream, in forward and — because
# autograd runs a node's backward on its forward stream — in backward, so the two kernel chains fill each other's tails.
_DECODE = {}
_USE_DECODE = False
_DECODE_USED = set()


def set_decoder_overlap(on):
    global _USE_DECODE
    _USE_DECODE = bool(on)


def decoder_overlap():
    return _USE_DECODE


def decode_stream(dev):
    st = _DECODE.get(dev)
    if st is None:
        st = _DECODE[dev] = torch.cuda.Stream(device=dev)
    _DECODE_USED.add(dev)
    return st


# Inference, asynchronous RAM-Net: the state update of scale i depends on the encoder feature x_i and the state h_i only
# (statenet.py:215-237 — x chains through the encoders, the states do not), so the three ConvGRU / ConvLSTM updates of one
# measurement are independent of each other and of the next encoder.  With set_branch_overlap(True) every scale but the last runs
# on a stream of its own (forked behind its encoder, joined at the end of the update): at batch 1 a launch fills a fraction of
# the chip, and the critical path of an update shrinks from head + 3 encoders + 6 state launches to head + 3 encoders + 2.
_BRANCH = {}
_USE_BRANCH = False


def set_branch_overlap(on):
    global _USE_BRANCH
    _USE_BRANCH = bool(on)


def branch_overlap():
    return _USE_BRANCH


def branch_stream(dev, i):
    st = _BRANCH.get((dev, i))
    if st is None:
        st = _BRANCH[(dev, i)] = torch.cuda.Stream(device=dev)
    return st


def _side_stream(dev):
    st = _SIDE.get(dev)
    if st is None:
        # (a lower stream priority measured no effect; a second side stream for the decoders' backward-weights path neither: 280.9 / 280.7
        # against 280.9 / 280.5 ms per step, round 5)
        st = _SIDE[dev] = torch.cuda.Stream(device=dev)
    return st


@contextlib.contextmanager
def side_work(tensors, dev):
    """Everything enqueued inside runs on the backward-weights side stream (behind what the current stream has enqueued so far);
    `tensors` = what it reads of the current stream's tensors (kept alive for the side stream).  Inline when the side stream is off."""
    if not _USE_SIDE:
        yield
        return
    side = _side_stream(dev)
    side.wait_event(torch.cuda.current_stream().record_event())
    with torch.cuda.stream(side):
        yield
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    _Engine.side_used = dev


def wgrad_side(tensors, *args, **kw):
    """wgrad_launch on the side stream (or inline when disabled)."""
    if not _USE_SIDE:
        return wgrad_launch(*args, **kw)
    dev = args[0].device
    side = _side_stream(dev)
    # (the launch takes the side stream explicitly and the fork is ONE library call: no Stream / Event objects, no context manager —
    # ~30 us of host time per launch less on the autograd thread)
    raw = C.c_void_p(side.cuda_stream)
    H.check(H.lib().ramnet_stream_fork(_st(), raw), "ramnet_stream_fork")
    wgrad_launch(*args, stream=raw, **kw)
    for t in tensors:
        if t is not None:
            t.record_stream(side)
    _Engine.side_used = dev


# Deferred backward-weights of the recurrent cells.  Every update of a ConvGRU scale runs the same two convolutions, so the weight
# gradients of several updates can be reduced by ONE launch over the updates' tensors (ramnet_wgrad_desc.segs) instead of one launch per
# update: a launch pays its prologue and — more — the join of every tile split's partial sums with the layer's workspace once (24 x 32 x 32
# floats read and written per workgroup: ~40 us of a ~250 us launch at the training batch) whatever its length.  The cell's backward
# queues its tensors on the layer (ConvParam._defer) and goes on with the backward-data chain; the queue is launched on the
# backward-weights stream when it holds `_WGRAD_DEFER` updates and, for what is left, when the engine finishes the pass.
_WGRAD_DEFER = 0


def set_wgrad_defer(n):
    """Cell updates per deferred multi-segment backward-weights launch of a ConvGRU layer (0 / 1: every update launches its own;
    <= 48).  The queued tensors (x, h, [u|r], the gate gradients: ~0.24 GB per update at the bench shape) stay alive until the launch."""
    global _WGRAD_DEFER
    n = int(n)
    assert 0 <= n <= H.WGRAD_MAX_SEGMENTS
    _WGRAD_DEFER = n


def wgrad_defer():
    return _WGRAD_DEFER


def _wgrad_cell(cp, dw, tensors, x0, taps, dout, Cout, **kw):
    """Backward-weights of one recurrent-cell convolution: launched now (wgrad_side) or queued on the layer for a multi-segment launch."""
    if _WGRAD_DEFER <= 1 or not getattr(dw, "wino6", False) or not _WGRAD_SLABS:
        return wgrad_side(tensors, x0, taps, dout, dw, Cout, **kw)
    x1, xm = kw.get("x1"), kw.get("xm")
    key = (id(dw), tuple(x0.shape), ld(x0), ld(x1) if x1 is not None else 0, ld(xm) if xm is not None else 0, ld(dout), Cout,
           kw.get("in_mode"), kw.get("C1"), kw.get("xm_off", 0), x0.device)
    q = cp._defer
    if q and q[0][0] != key:
        cp.flush_deferred()
    cp._defer.append((key, tensors, x0, taps, dout, dw, Cout, kw))
    if len(cp._defer) >= _WGRAD_DEFER:
        cp.flush_deferred()


# ------------------------------------------------------------------------------------------------ parameters
# Workspace zero-fills of the end-of-backward fold: inside _Engine.flush they are collected and issued as ONE multi-tensor launch behind the
# last layer's fold (round 6: ~76 fills of ~5 us each at the very end of a training step, where nothing runs beside them); outside a flush
# (a ConvParam finalised on its own) they run at once.
_ZERO_BATCH = None


def _zero_later(t):
    if _ZERO_BATCH is None:
        t.zero_()
    else:
        _ZERO_BATCH.append(t)


class _Engine:
    """Per-backward-pass bookkeeping: fold weight-gradient workspaces into .grad when the autograd engine finishes the pass.

    Keyed on the engine's graph task: the first weight-gradient launch of a backward pass queues ONE end-of-pass callback
    (flush).  The engine runs callbacks only when the pass completes; a pass that raises mid-way (OOM, a bad argument, a hook)
    leaves workspaces partly filled, so the next pass — recognised by its new task id — first discards them (reset)."""
    dirty = []
    task = -1                # graph task whose callback is queued (-1: none / launches outside autograd, caller flushes)
    side_used = None

    @classmethod
    def enter(cls):
        tid = torch._C._current_graph_task_id()
        if tid != cls.task:
            if cls.dirty:            # leftovers of a pass that never reached its callback
                cls.reset()
            cls.task = tid
            if tid != -1:
                torch.autograd.Variable._execution_engine.queue_callback(cls.flush)

    @classmethod
    def mark(cls, cp):
        cls.enter()
        if not cp._dirty:
            cp._dirty = True
            cls.dirty.append(cp)

    @classmethod
    def _join(cls):
        for dev in _DECODE_USED:           # decoder backward (and its weight-gradient launches) ran on the decode stream
            torch.cuda.current_stream(dev).wait_stream(_DECODE[dev])
        if cls.side_used is not None:      # all weight-gradient launches of this pass are done before the fold
            torch.cuda.current_stream().wait_event(_side_stream(cls.side_used).record_event())
            cls.side_used = None

    @classmethod
    def flush(cls):
        for cp in cls.dirty:               # queued cell updates go to the backward-weights stream before it is joined
            cp.flush_deferred()
        cls._join()
        dirty, cls.dirty, cls.task = cls.dirty, [], -1
        # data-parallel runs: parallel.FlatGradReducer watches the folds and sends a gradient bucket off as soon as its last
        # parameter is final — not while a hipGraph is being captured (collectives stay outside the training graph)
        hook = _FINALIZE_HOOK if (_FINALIZE_HOOK is not None and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing())) else None
        if hook is not None:
            hook.begin(dirty)
        global _ZERO_BATCH
        _ZERO_BATCH = []
        try:
            for cp in dirty:
                cp.finalize()
                if hook is not None:
                    hook.finalized(cp)
        finally:
            zs, _ZERO_BATCH = _ZERO_BATCH, None
            zs = [z for z in zs if z.numel() > 0]
            if zs:
                torch._foreach_zero_(zs)
        if hook is not None and hasattr(hook, "end"):
            hook.end()

    @classmethod
    def reset(cls):
        """Drop the partial weight-gradient sums of an aborted backward pass (nothing reaches .grad)."""
        for cp in cls.dirty:
            cp._defer = []
        cls._join()
        dirty, cls.dirty, cls.task = cls.dirty, [], -1
        for cp in dirty:
            cp.discard()


_FINALIZE_HOOK = None


def set_finalize_hook(hook):
    """hook.begin(list of ConvParam about to be folded) / hook.finalized(cp) are called by the end-of-backward fold; None removes it."""
    global _FINALIZE_HOOK
    _FINALIZE_HOOK = hook


def get_finalize_hook():
    return _FINALIZE_HOOK


def ensure_grad(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


# ---- folded upsample-conv: weight algebra of the direct-kernel pack (pure torch; the plain-torch restatements of the HIP pack / unpack
# kernels that the tests compare against live in tests/torch_restatements.py)
# high-res row 2i + p + k - 2 (k = 0..4) of the x2 bilinear upsample is a 0.25 / 0.75 blend of two low-res rows;
# FOLD_A[p][t][k] = weight of low-res row i + p - 2 + t (t = 0..3) in tap k of output parity p.
FOLD_A = ([[.25, 0, 0, 0, 0], [.75, .75, .25, 0, 0], [0, .25, .75, .75, .25], [0, 0, 0, .25, .75]],
          [[.75, .25, 0, 0, 0], [.25, .75, .75, .25, 0], [0, 0, .25, .75, .75], [0, 0, 0, 0, .25]])


_CONSTS = {}


def _const(name, values, device, dtype):
    """Small constant matrices on the device, uploaded once (a host-to-device copy has no place inside a hipGraph capture)."""
    key = (name, str(device), dtype)
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(values, dtype=dtype, device=device)
    return t


def fold_weights(w):
    """OIHW 5x5 -> [O][I][py][px][ty][tx]: the 4x4 filter of every output parity, W4 = A_py w A_px^T (float64)."""
    A = _const("FOLD_A", FOLD_A, w.device, torch.float64)
    return torch.einsum("ptk,qsl,oikl->oipqts", A, A, w.double())


# ---- stride-2 5x5 convolution == stride-1 3x3 convolution of the space-to-depth input (4*Cin channels): tap ky of the 5x5
# filter reads input row 2i + ky - 2 = 2(i + dy) + a with (dy, a) = S2D_TAP[ky]; the slice (dy = 1, a = 1) does not exist (zero).
S2D_TAP = [(-1, 0), (-1, 1), (0, 0), (0, 1), (1, 0)]


def s2d_weights(w):
    """OIHW [O][I][5][5] -> [O][4*I][3][3] with input channel (a*2 + b)*I + i (pure torch; tests/test_boundary_cpu.py)."""
    O, I = w.shape[0], w.shape[1]
    out = w.new_zeros(O, 4, I, 3, 3)
    for ky, (dy, a) in enumerate(S2D_TAP):
        for kx, (dx, b) in enumerate(S2D_TAP):
            out[:, a * 2 + b, :, dy + 1, dx + 1] = w[:, :, ky, kx]
    return out.reshape(O, 4 * I, 3, 3)


_S2D_ADJ_IDX = {}


def s2d_weights_adjoint(g3, I):
    """Gradient w.r.t. the [O][4*I][3][3] weights -> gradient w.r.t. the 5x5 weights (reads the 25 populated slices): tap (ky, kx) of the 5x5
    filter is entry (parity group a*2 + b, dy + 1, dx + 1) of the 3x3 one.  ONE transposing copy + ONE gather (round 6: 25 strided slice copies
    + a fill per encoder were 156 launches of ~5 us at the very end of a training step, where nothing runs beside them)."""
    O = g3.shape[0]
    idx = _S2D_ADJ_IDX.get(g3.device)
    if idx is None:
        idx = _S2D_ADJ_IDX[g3.device] = torch.tensor([(a * 2 + b) * 9 + (dy + 1) * 3 + (dx + 1) for (dy, a) in S2D_TAP for (dx, b) in S2D_TAP],
                                                     dtype=torch.int64, device=g3.device)
    t = g3.view(O, 4, I, 9).permute(0, 2, 1, 3).reshape(O, I, 36)           # [O][I][group * 9 + 3 * row + column]
    return t.index_select(2, idx).view(O, I, 5, 5)


# Packed weights are cached per parameter version (an optimizer step bumps it).  A hipGraph replays kernels, not Python: when a
# training step is captured, every pack kernel has to be IN the graph (the weights change between replays), so the capture
# starts a new epoch and all packs are re-run — and recorded — once.
_PACK_EPOCH = 0


def invalidate_packs():
    global _PACK_EPOCH
    _PACK_EPOCH += 1


class PackRef:
    """Handle on the packed weights of a ConvParam; the launch picks the layout (direct / Winograd) that fits it."""
    __slots__ = ("cp", "transposed")

    def __init__(self, cp, transposed):
        self.cp, self.transposed = cp, transposed


class ConvParam:
    """Kernel-side state of one convolution: packed weights (forward / backward-data layouts, re-packed when the
    nn.Parameter version changes) and the weight/bias gradient workspaces.  ``weights`` may hold several OIHW
    parameters that are fused along O into one launch (ConvGRU update|reset gates)."""

    def __init__(self, weights, biases, gates=1):
        self.weights, self.biases, self.gates = list(weights), list(biases), gates
        w0 = self.weights[0]
        self.Cout = sum(w.shape[0] for w in self.weights)
        self.Cin, self.k = w0.shape[1], w0.shape[2]
        self.CinWs = (self.Cin + 3) // 4 * 4
        self._bias = self._ws = self._bws = None
        self._ws_fold = None             # (dW4 [64][CinWs][Cout], dWrows, dWcols) of the folded upsample-conv backward-weights
        self._fold_used = self._ws_used = False
        self._vbias = None
        self._packs = {}
        self._dirty = False
        self._defer = []                 # queued backward-weights launches of a recurrent cell (_wgrad_cell)

    def flush_deferred(self):
        """Launch the queued cell updates as ONE multi-segment backward-weights launch (ramnet_wgrad_desc.segs)."""
        q, self._defer = self._defer, []
        if not q:
            return
        _, _, x0, taps, dout, dw, Cout, kw = q[0]
        if len(q) == 1:
            return wgrad_side(q[0][1], x0, taps, dout, dw, Cout, **kw)
        segs = (H.WgradSeg * len(q))()
        keep = []
        for i, (_, tensors, xi, _, di, _, _, kwi) in enumerate(q):
            segs[i].x0, segs[i].x1, segs[i].xm = _p(xi), _p(kwi.get("x1")), _p(kwi.get("xm"), kwi.get("xm_off", 0))
            segs[i].dout, segs[i].gmask = _p(di), _p(kwi.get("gmask"))
            keep += list(tensors)
        wgrad_side(keep, x0, taps, dout, dw, Cout, segs=segs, **kw)

    def splitk_ws(self, n, device):
        """Workspace of this layer's split-reduction launches (ramnet_conv_desc.splitk_ws): zero once — the kernel leaves the arrival
        counters at zero —, one buffer per size; the launches of ONE layer are ordered on one stream everywhere in this package
        (per-scale update chains, decoder stream and backward-weights stream run different layers).  First use inside a graph capture
        would record the zero-fill into the graph: the streaming runtimes warm up eagerly before they capture."""
        pool = self.__dict__.setdefault("_splitk", {})       # (derived parameter views — S2DConvParam — build themselves)
        ws = pool.get(n)
        if ws is None or ws.device != device:
            if torch.cuda.is_current_stream_capturing():
                # (the zero-fill would be recorded into the graph and the buffer would come out of the graph's private pool while eager
                #  launches of the same layer share it: ADVICE r4)
                raise RuntimeError("split-reduction workspace of a layer first used inside a graph capture: run the launch eagerly once "
                                   "before capturing (graph._warm)")
            ws = pool[n] = torch.zeros(n, device=device, dtype=torch.float32)
        return ws

    def _versions(self, ts):
        return (_PACK_EPOCH,) + tuple((t._version, t.data_ptr()) for t in ts)

    def _cat_w(self):
        w = self.weights[0] if len(self.weights) == 1 else torch.cat([w.detach() for w in self.weights], 0)
        return w.detach().contiguous()

    def _pack(self, transposed, wino):
        L, g = H.lib(), (self.gates if not transposed else 1)
        w = self._cat_w()
        if wino == "head":
            out = torch.empty(L.ramnet_packed_weight_elems_head(self.Cin), device=w.device, dtype=torch.float32)
            H.check(L.ramnet_pack_weight_head(_p(w), _p(out), self.Cout, self.Cin, _st()), "ramnet_pack_weight_head")
            return out
        if wino == "2x4":
            out = torch.empty(L.ramnet_packed_weight_elems_wino2x4(self.Cout, self.Cin, transposed), device=w.device, dtype=torch.float32)
            H.check(L.ramnet_pack_weight_wino2x4(_p(w), _p(out), self.Cout, self.Cin, transposed, _st()), "ramnet_pack_weight_wino2x4")
            return out
        if wino == "2x4s":      # three bf16 planes of the same Winograd-domain weights (csrc/conv_wino6s.hip); the size is in 4-byte units
            out = torch.empty(L.ramnet_packed_weight_elems_wino2x4_split(self.Cout, self.Cin, transposed), device=w.device, dtype=torch.float32)
            H.check(L.ramnet_pack_weight_wino2x4_split(_p(w), _p(out), self.Cout, self.Cin, transposed, _st()), "ramnet_pack_weight_wino2x4_split")
            return out
        if wino:
            n = L.ramnet_packed_weight_elems_wino(self.Cout, self.Cin, transposed, g)
            out = torch.empty(n, device=w.device, dtype=torch.float32)
            H.check(L.ramnet_pack_weight_wino(_p(w), _p(out), self.Cout, self.Cin, transposed, g, _st()), "ramnet_pack_weight_wino")
            return out
        n = L.ramnet_packed_weight_elems(self.Cout, self.Cin, self.k, self.k, transposed, g)
        out = torch.empty(n, device=w.device, dtype=torch.float32)
        H.check(L.ramnet_pack_weight(_p(w), _p(out), self.Cout, self.Cin, self.k, self.k, transposed, g, _st()), "ramnet_pack_weight")
        return out

    def pack(self, transposed, wino=False):
        """Packed weights for the forward (transposed=0) / backward-data (1) launch, re-packed when a parameter changes."""
        v = self._versions(self.weights)
        wino = wino if wino in ("head", "2x4", "2x4s") else bool(wino)
        key = (transposed, wino)
        hit = self._packs.get(key)
        if hit is None or hit[0] != v:
            hit = self._packs[key] = (v, self._pack(transposed, wino))
        return hit[1]

    def fwd(self):
        return PackRef(self, 0)

    # -- folded upsample-conv (5x5 after bilinear x2): effective 4x4 kernels of the four output parities + border matrices
    _FOLD_A = FOLD_A

    def pack_fold(self):
        """[64 = (py, px, ty, tx)] tap slices of fold_weights() in the direct kernel's layout."""
        v = (self._versions(self.weights), "fold")
        hit = self._packs.get("fold")
        if hit is None or hit[0] != v:
            w4 = fold_weights(self._cat_w()).reshape(self.Cout, self.Cin, 8, 8).float().contiguous()
            L = H.lib()
            out = torch.empty(L.ramnet_packed_weight_elems(self.Cout, self.Cin, 8, 8, 0, 1), device=w4.device)
            H.check(L.ramnet_pack_weight(_p(w4), _p(out), self.Cout, self.Cin, 8, 8, 0, 1, _st()), "ramnet_pack_weight")
            hit = self._packs["fold"] = (v, out)
        return hit[1]

    def pack_fold_wino(self):
        """Winograd-domain weights of the folded upsample-conv (conv_wino24_kernel), cached per parameter version."""
        v = (self._versions(self.weights), "fold24")
        hit = self._packs.get("fold24")
        if hit is None or hit[0] != v:
            L, w = H.lib(), self._cat_w()
            out = torch.empty(L.ramnet_packed_weight_elems_fold_wino(self.Cout, self.Cin), device=w.device, dtype=torch.float32)
            H.check(L.ramnet_pack_weight_fold_wino(_p(w), _p(out), self.Cout, self.Cin, _st()), "ramnet_pack_weight_fold_wino")
            hit = self._packs["fold24"] = (v, out)
        return hit[1]

    def pack_fold_wino_dgrad(self):
        v = (self._versions(self.weights), "fold24d")
        hit = self._packs.get("fold24d")
        if hit is None or hit[0] != v:
            L, w = H.lib(), self._cat_w()
            out = torch.empty(L.ramnet_packed_weight_elems_fold_wino(self.Cout, self.Cin), device=w.device, dtype=torch.float32)
            H.check(L.ramnet_pack_weight_fold_wino_dgrad(_p(w), _p(out), self.Cout, self.Cin, _st()), "ramnet_pack_weight_fold_wino_dgrad")
            hit = self._packs["fold24d"] = (v, out)
        return hit[1]

    def _border_pack(self):
        """(rows, cols, rows^T, cols^T) of the border matrices (tests/torch_restatements.border_matrices is the plain-torch statement; one launch of
        ramnet_pack_border_weights), cached per parameter version."""
        v = (self._versions(self.weights), "border")
        hit = self._packs.get("border")
        if hit is None or hit[0] != v:
            w = self._cat_w()
            rows, cols = (torch.empty(2, 5 * self.Cin, 2 * self.Cout, device=w.device) for _ in range(2))
            rows_t, cols_t = (torch.empty(2, 2 * self.Cout, 5 * self.Cin, device=w.device) for _ in range(2))
            H.check(H.lib().ramnet_pack_border_weights(_p(w), _p(rows), _p(cols), _p(rows_t), _p(cols_t), self.Cout, self.Cin, _st()),
                    "ramnet_pack_border_weights")
            hit = self._packs["border"] = (v, (rows, cols, rows_t, cols_t))
        return hit[1]

    def border_weights(self):
        """[2 sides][5*Cin][2*Cout] row / column border matrices of the current weights."""
        return self._border_pack()[:2]

    def border_weights_t(self):
        """border_weights() transposed to [2 sides][2*Cout][5*Cin] (the backward-data operand)."""
        return self._border_pack()[2:]

    def bwd(self):
        return PackRef(self, 1)

    def s2d(self):
        """3x3 view over the space-to-depth input of this 5x5 stride-2 convolution (S2DConvParam), created once."""
        if getattr(self, "_s2d", None) is None:
            self._s2d = S2DConvParam(self)
        return self._s2d

    def bias(self):
        if self.biases[0] is None:          # norm 'BN': the convolution has no bias (submodules.py:13)
            return None
        if len(self.biases) == 1:
            return self.biases[0].detach()
        v = self._versions(self.biases)
        if self._bias is None or v != self._vbias:
            self._bias, self._vbias = torch.cat([b.detach() for b in self.biases]).contiguous(), v
        return self._bias

    def grad_ws(self, wino_ok=False):
        """([tap][CinWs][Cout] weight-gradient workspace, [Cout] bias-gradient workspace), zero at pass start.
        wino_ok: the caller's launch is a plain 3x3 stride-1 convolution (no upsampling loader), i.e. eligible for the
        Winograd backward-weights kernel, whose workspace holds the transformed-domain gradient [16][CinWs][Cout]."""
        if self._ws is None:
            dev = self.weights[0].device
            slots = 24 if self.k == 3 else self.k * self.k       # 3x3: room for the Winograd-domain gradient dU (F(2x4): 24, F(2x2): 16 positions)
            # 3x3 layers: one slab per tile split of the Winograd backward-weights launch (joined by plain read-modify-write: no atomics,
            # bit-reproducible sums; ramnet_wgrad_desc.dw_slabs) — folded into slab 0 by finalize(); other layers: one slab
            self._slabs = H.lib().ramnet_wgrad_wino_slabs(self.CinWs, self.Cout) if self.k == 3 else 1
            # (F(2x4): blocked layout padded to 32 x 32-channel blocks, half as many slabs as F(2x2): never larger than the dense 24-slot slabs
            # of F(2x2)'s count unless the channel counts are ragged — take the maximum)
            n6 = H.lib().ramnet_wgrad_wino2x4_ws_floats(self.CinWs, self.Cout) * H.lib().ramnet_wgrad_wino2x4_slabs(self.CinWs, self.Cout) if self.k == 3 else 0
            self._ws = torch.zeros(max(self._slabs * slots * self.CinWs * self.Cout, n6), device=dev)
            self._bws = torch.zeros(self._slabs * self.Cout, device=dev)
            self._ws.slabs = self._slabs
        _Engine.enter()         # (a new pass after an aborted one resets _dirty first)
        if not self._dirty:     # one algorithm per backward pass: every launch of the pass accumulates into the same layout
            self._ws.wino = bool(wino_ok and _WINOGRAD and self.k == 3
                                 and self.CinWs >= _WINO_MIN_CIN)
            # F(2x4,3x3) for the plain 3x3 layers (not the space-to-depth views): slabs of [24][Cin][Cout], half as many as F(2x2)'s
            self._ws.wino6 = bool(self._ws.wino and (_WGRAD_2X4 == "force" or (_WGRAD_2X4 == "auto" and _USE_SIDE))
                                  and (type(self) is ConvParam or (_S2D_2X4 and self.Cin >= 128)) and self.CinWs >= 64)
            # ... and with split operands on, those layers' backward-weights in direct form on the bf16 matrix pipe: slabs of [9][Cin][Cout]
            self._ws.wg_dsplit = bool(self._ws.wino6 and _SPLIT_OPERANDS and _SPLIT_WGRAD)
            if self._ws.wg_dsplit:
                ns, nf = H.lib().ramnet_wgrad_dsplit_slabs(self.CinWs, self.Cout), H.lib().ramnet_wgrad_dsplit_ws_floats(self.CinWs, self.Cout)
                if self._ws.numel() < ns * nf or self._bws.numel() < ns * self.Cout:       # (between passes the workspaces hold zeros: grow them)
                    old = self._ws
                    self._ws = torch.zeros(max(ns * nf, old.numel()), device=old.device)
                    self._ws.wg_dsplit = True
                    self._bws = torch.zeros(max(ns, self._slabs) * self.Cout, device=old.device)
                self._ws.wino = self._ws.wino6 = False
                self._ws.slabs = ns
            elif self._ws.wino6:
                self._ws.wino = False
                self._ws.slabs = min(self._slabs, H.lib().ramnet_wgrad_wino2x4_slabs(self.CinWs, self.Cout))
            else:
                self._ws.slabs = self._slabs
            # head layers: the launch may run the head kernel (same [tap][CinWs][Cout] layout as the direct kernel)
            self._ws.head_cin = self.Cin if (self.k == 5 and self.gates == 1 and len(self.weights) == 1
                                             and H.lib().ramnet_head_supported(self.Cin, self.Cout)) else 0
        _Engine.mark(self)
        self._ws_used = True
        return self._ws, self._bws

    def grad_ws_fold(self):
        """Workspaces of the folded upsample-conv backward-weights: dW4 [64 = (py,px,ty,tx)][CinWs][Cout] of the four parity
        convolutions, the gradients of the two border-GEMM weight stacks, and the shared bias workspace."""
        assert len(self.weights) == 1 and self.k == 5
        dev = self.weights[0].device
        if self._bws is None:
            self._bws = torch.zeros(self.Cout, device=dev)
        if self._ws_fold is None:
            self._ws_fold = (torch.zeros(64 * self.CinWs * self.Cout, device=dev),
                             torch.zeros(2, 5 * self.Cin, 2 * self.Cout, device=dev),
                             torch.zeros(2, 5 * self.Cin, 2 * self.Cout, device=dev))
        _Engine.mark(self)
        self._fold_used = True
        return self._ws_fold + (self._bws,)

    def grad_ws_fold24(self):
        """dU [4 parity classes][25 positions][Cin][Cout]: Winograd-domain gradient of the four 4x4 parity filters."""
        if getattr(self, "_ws_fold24", None) is None:
            self._ws_fold24 = torch.zeros(4 * 25 * self.CinWs * self.Cout, device=self.weights[0].device)
        self._fold24_used = True
        return self._ws_fold24

    def _finalize_fold(self):
        """dW5 += sum_parities A_py^T dW4 A_px  -  (border GEMM gradients routed back to the taps they summed), dW4 = the direct parity
        launches' workspace + G^T dU G of the Winograd-domain launches: one launch (ramnet_fold_unpack_wgrad), which also zeroes the
        workspaces for the next pass."""
        w4, wr, wc = self._ws_fold
        g = ensure_grad(self.weights[0])
        dU = self._ws_fold24 if getattr(self, "_fold24_used", False) else None
        H.check(H.lib().ramnet_fold_unpack_wgrad(_p(w4), _p(dU), _p(wr), _p(wc), _p(g), self.Cout, self.Cin, self.CinWs, _st()),
                "ramnet_fold_unpack_wgrad")
        self._fold24_used = False
        self._fold_used = False

    def discard(self):
        """Zero every gradient workspace without folding it (aborted backward pass)."""
        for t in (self._ws, self._bws, getattr(self, "_ws_fold24", None)) + tuple(self._ws_fold or ()):
            if t is not None:
                t.zero_()
        self._dirty = self._ws_used = self._fold_used = self._fold24_used = False

    def _join_slabs(self):
        """Winograd backward-weights slabs -> slab 0 (fixed order), weights and bias."""
        w6, ds = getattr(self._ws, "wino6", False), getattr(self._ws, "wg_dsplit", False)
        ns = getattr(self._ws, "slabs", 1)
        if ns > 1 and (w6 or ds or getattr(self._ws, "wino", False)):
            n = (H.lib().ramnet_wgrad_wino2x4_ws_floats(self.CinWs, self.Cout) if w6 else
                 H.lib().ramnet_wgrad_dsplit_ws_floats(self.CinWs, self.Cout) if ds else 16 * self.CinWs * self.Cout)
            H.check(H.lib().ramnet_reduce_slabs(_p(self._ws), ns, n, _st()), "ramnet_reduce_slabs")
            H.check(H.lib().ramnet_reduce_slabs(_p(self._bws), ns, self.Cout, _st()), "ramnet_reduce_slabs")

    def _zero_ws(self):
        """Zero what the pass used of the gradient workspace for the next one: slab 0 only where the slabs were joined
        (ramnet_reduce_slabs leaves slabs 1.. zeroed), not the whole buffer sized for the largest layout."""
        w6, wn, ds = getattr(self._ws, "wino6", False), getattr(self._ws, "wino", False), getattr(self._ws, "wg_dsplit", False)
        ns = getattr(self._ws, "slabs", 1)
        if w6:
            n = H.lib().ramnet_wgrad_wino2x4_ws_floats(self.CinWs, self.Cout)
        elif wn:
            n = 16 * self.CinWs * self.Cout
        elif ds:
            n = H.lib().ramnet_wgrad_dsplit_ws_floats(self.CinWs, self.Cout)
        else:
            n, ns = self.k * self.k * self.CinWs * self.Cout, 1
        joined = ns > 1 and (w6 or wn or ds)
        _zero_later(self._ws[:n if joined else min(self._ws.numel(), n * max(ns, 1))])
        _zero_later(self._bws[:self.Cout if joined else self._bws.numel()])

    def finalize(self):
        if self._fold_used:
            self._finalize_fold()
        if self._ws_used:
            self._join_slabs()
        if not self._ws_used:
            if self.biases[0] is not None and self.biases[0].shape[0] == self.Cout:
                ensure_grad(self.biases[0]).add_(self._bws[:self.Cout])
            _zero_later(self._bws)
            self._dirty = False
            return
        off = 0
        for w, b in zip(self.weights, self.biases):
            n = w.shape[0]
            g = ensure_grad(w)
            if getattr(self._ws, "wino6", False):
                H.check(H.lib().ramnet_unpack_wgrad_wino2x4(_p(self._ws), _p(g), n, self.Cin, self.CinWs, self.Cout, off, _st()),
                        "ramnet_unpack_wgrad_wino2x4")
            elif getattr(self._ws, "wg_dsplit", False):
                H.check(H.lib().ramnet_unpack_wgrad_dsplit(_p(self._ws), _p(g), n, self.Cin, self.CinWs, self.Cout, off, _st()),
                        "ramnet_unpack_wgrad_dsplit")
            elif getattr(self._ws, "wino", False):
                H.check(H.lib().ramnet_unpack_wgrad_wino(_p(self._ws), _p(g), n, self.Cin, self.CinWs, self.Cout, off, _st()),
                        "ramnet_unpack_wgrad_wino")
            else:
                H.check(H.lib().ramnet_unpack_wgrad(_p(self._ws), _p(g), n, self.Cin, self.CinWs, self.Cout, off,
                                                    self.k, self.k, _st()), "ramnet_unpack_wgrad")
            if b is not None and b.shape[0] == n:      # (transposed conv: bias has Cout_t entries, handled by its op)
                ensure_grad(b).add_(self._bws[off:off + n])
            off += n
        self._zero_ws()
        self._dirty = self._ws_used = False


class S2DConvParam(ConvParam):
    """The 3x3 / 4*Cin view of a stride-2 5x5 convolution (s2d_weights): gives that layer the Winograd kernels for forward,
    backward-data and backward-weights.  Packs follow the parent's parameter versions; the weight gradient accumulates in the
    Winograd-domain workspace of THIS object and is mapped back into the parent's 5x5 `.grad` when the engine finishes."""

    def __init__(self, parent):
        assert len(parent.weights) == 1 and parent.k == 5 and parent.gates == 1
        self.parent = parent
        self.weights, self.biases, self.gates = parent.weights, parent.biases, 1
        self.Cout, self.Cin, self.k = parent.Cout, 4 * parent.Cin, 3
        self.CinWs = self.Cin
        self._bias = self._ws = self._bws = None
        self._vbias = None
        self._ws_fold = None
        self._fold_used = self._ws_used = False
        self._packs = {}
        self._dirty = False
        self._defer = []

    def _cat_w(self):
        return s2d_weights(self.parent.weights[0].detach()).contiguous()

    def finalize(self):
        w, b = self.parent.weights[0], self.parent.biases[0]
        g3 = torch.zeros(self.Cout, self.Cin, 3, 3, device=w.device)
        L = H.lib()
        self._join_slabs()
        if getattr(self._ws, "wino6", False):
            H.check(L.ramnet_unpack_wgrad_wino2x4(_p(self._ws), _p(g3), self.Cout, self.Cin, self.CinWs, self.Cout, 0, _st()), "ramnet_unpack_wgrad_wino2x4")
        elif getattr(self._ws, "wg_dsplit", False):
            H.check(L.ramnet_unpack_wgrad_dsplit(_p(self._ws), _p(g3), self.Cout, self.Cin, self.CinWs, self.Cout, 0, _st()), "ramnet_unpack_wgrad_dsplit")
        elif getattr(self._ws, "wino", False):
            H.check(L.ramnet_unpack_wgrad_wino(_p(self._ws), _p(g3), self.Cout, self.Cin, self.CinWs, self.Cout, 0, _st()), "ramnet_unpack_wgrad_wino")
        else:
            H.check(L.ramnet_unpack_wgrad(_p(self._ws), _p(g3), self.Cout, self.Cin, self.CinWs, self.Cout, 0, 3, 3, _st()), "ramnet_unpack_wgrad")
        ensure_grad(w).add_(s2d_weights_adjoint(g3, self.parent.Cin))
        if b is not None:
            ensure_grad(b).add_(self._bws[:self.Cout])
        self._zero_ws()
        self._dirty = self._ws_used = False


# ------------------------------------------------------------------------------------------------ operators
def pack_input(x, device, crop=None):
    """Model input NCHW (any device) -> NHWC with channels zero-padded to a multiple of 4 (model.py:177,200).
    crop: a CropParameters (full-frame mode, utils/inference_utils.py:287-314) — the frame is reflect-padded to crop.height_crop_size x
    crop.width_crop_size inside the same launch."""
    x = x.to(device=device, dtype=torch.float32).contiguous()
    B, Cc, Hh, W = x.shape
    cp = (Cc + 3) // 4 * 4
    if crop is not None and (crop.height_crop_size, crop.width_crop_size) != (Hh, W):
        assert (crop.height, crop.width) == (Hh, W), "full-frame mode: CropParameters built for %dx%d, input is %dx%d" % (crop.height, crop.width, Hh, W)
        out = torch.empty(B, crop.height_crop_size, crop.width_crop_size, cp, device=device)
        H.check(H.lib().ramnet_reflect_pad(_p(x), _p(out), B, Cc, Hh, W, cp, crop.padding_top, crop.padding_left, crop.height_crop_size,
                                           crop.width_crop_size, 1, _st()), "ramnet_reflect_pad")
        return out
    out = torch.empty(B, Hh, W, cp, device=device)
    H.check(H.lib().ramnet_nchw_to_nhwc_pad(_p(x), _p(out), B, Cc, Hh, W, cp, _st()), "ramnet_nchw_to_nhwc_pad")
    return out


class CropParameters:
    """utils/inference_utils.py:278-314: the smallest size >= (height, width) divisible by 2^num_encoders, the reflection padding that
    centres the frame in it (top / left get the ceil of half the excess) and the window that crops a network output back.  Same
    attribute names as the reference's class; `pad` is the HIP reflect-pad launch instead of torch.nn.ReflectionPad2d."""

    def __init__(self, width, height, num_encoders):
        from math import ceil, floor
        self.height, self.width, self.num_encoders = height, width, num_encoders
        f = 2 ** num_encoders
        self.width_crop_size = int(f * ceil(width / f))
        self.height_crop_size = int(f * ceil(height / f))
        self.padding_top = ceil(0.5 * (self.height_crop_size - height))
        self.padding_bottom = floor(0.5 * (self.height_crop_size - height))
        self.padding_left = ceil(0.5 * (self.width_crop_size - width))
        self.padding_right = floor(0.5 * (self.width_crop_size - width))
        self.cx, self.cy = floor(self.width_crop_size / 2), floor(self.height_crop_size / 2)
        self.ix0, self.ix1 = self.cx - floor(width / 2), self.cx + ceil(width / 2)
        self.iy0, self.iy1 = self.cy - floor(height / 2), self.cy + ceil(height / 2)

    @property
    def identity(self):
        return (self.height_crop_size, self.width_crop_size) == (self.height, self.width)

    def pad(self, x):
        """[B, C, height, width] (any device) -> reflect-padded [B, C, height_crop_size, width_crop_size] on the current device."""
        x = x.to(device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32).contiguous()
        B, Cc, Hh, W = x.shape
        assert (Hh, W) == (self.height, self.width)
        out = torch.empty(B, Cc, self.height_crop_size, self.width_crop_size, device=x.device)
        H.check(H.lib().ramnet_reflect_pad(_p(x), _p(out), B, Cc, Hh, W, 0, self.padding_top, self.padding_left, self.height_crop_size,
                                           self.width_crop_size, 0, _st()), "ramnet_reflect_pad")
        return out

    def crop(self, y):
        """Network output [..., height_crop_size, width_crop_size] -> the original frame's window (a view)."""
        return y[..., self.iy0:self.iy1, self.ix0:self.ix1]


def gemm(a, b, out, trans_a=False, accumulate=False):
    """out[i] (=, +=) a[i] @ b[i]  (trans_a: a[i]^T @ b[i]) for the entries i of a batch, in ONE launch of the library's own MFMA
    kernel (ramnet_gemm); a, b, out: [batch][rows][cols] row-major.  Plain products are bit-reproducible (one wave per 32 x 32
    block, no atomics); accumulate=True splits the reduction and joins by atomics (backward pass)."""
    nb, M, N = out.shape
    K = a.shape[1] if trans_a else a.shape[2]
    H.check(H.lib().ramnet_gemm(_p(a), _p(b), _p(out), M, N, K, a.stride(1), b.stride(1), out.stride(1), int(trans_a), int(accumulate),
                                nb, a.stride(0), b.stride(0), out.stride(0), _st()), "ramnet_gemm")


def gemm2(a1, b1, out1, a2, b2, out2, trans_a=False, accumulate=False):
    """gemm(a1, b1, out1) and gemm(a2, b2, out2) in ONE launch (ramnet_gemm2): the row and the column border of a decoder layer — same
    N, leading dimensions and mode; the two problems differ in M (forward, backward-data) or in the reduction length (weight gradient)."""
    nb1, M1, N = out1.shape
    nb2, M2, N2 = out2.shape
    K1 = a1.shape[1] if trans_a else a1.shape[2]
    K2 = a2.shape[1] if trans_a else a2.shape[2]
    assert N == N2 and a1.stride(1) == a2.stride(1) and b1.stride(1) == b2.stride(1) and out1.stride(1) == out2.stride(1)
    H.check(H.lib().ramnet_gemm2(_p(a1), _p(b1), _p(out1), M1, a1.stride(0), b1.stride(0), out1.stride(0), nb1, _p(a2), _p(b2), _p(out2), M2,
                                 a2.stride(0), b2.stride(0), out2.stride(0), nb2, N, K1, K2, a1.stride(1), b1.stride(1), out1.stride(1),
                                 int(trans_a), int(accumulate), _st()), "ramnet_gemm2")


def _folded_upsample_conv(x, skip, cp, y, epi):
    """y = act(conv5x5_zero_padded(up2x(x + skip)) + b) without ever forming the upsampled image:
    (1) ONE multi-class launch: each output parity is a 4x4 convolution of the replicate-padded low-res sum — 16 taps instead
        of 25, plain loads — which equals the 5x5 convolution of the REPLICATE-extended upsample;
    (2) the true layer zero-pads instead, so the outermost two rows / columns lose the taps that fall outside: those taps see a
        constant line (the clamped border row / column of the upsample), i.e. four small plain GEMMs
        [border pixels x 5*Cin] x [5*Cin x 2*Cout] (ramnet_gemm, csrc/gemm_skinny.hip; 1-3 % of the layer's FLOP), whose results the
        epilogue of (1) adds to the pre-activation of the frame pixels (ramnet_conv_desc.frame)."""
    L = H.lib()
    B, Hh, W, Cc = x.shape
    dev = x.device
    if Cc != cp.Cin:
        raise RuntimeError("folded upsample-conv needs un-padded input channels")
    H2, W2 = 2 * Hh, 2 * W
    xpad = torch.empty(B, Hh + 4, W + 4, Cc, device=dev)
    # inference with branch streams: the border path (im2col -> two GEMMs) does not depend on the padded sum and runs beside it
    side = branch_stream(dev, 8) if (_USE_BRANCH and not torch.is_grad_enabled()) else None
    main = torch.cuda.current_stream()
    # every lazily packed operand the side stream reads is packed (on main, on a cache miss: first call, after an optimizer step or
    # load_state_dict) BEFORE the side stream waits on main — the wait then orders gemm2 behind the kernels that write the border weights
    w_rows, w_cols = cp.border_weights()                                          # [2 sides][5*Cin][2*Cout]
    if side is not None:
        side.wait_stream(main)
    if side is not None:
        H.check(L.ramnet_pad2_sum(_p(x), _p(skip), _p(xpad), B, Hh, W, Cc, _st()), "ramnet_pad2_sum")
    with torch.cuda.stream(side if side is not None else main):
        a_rows = torch.empty(2, B * W2, 5 * Cc, device=dev)
        a_cols = torch.empty(2, B * H2, 5 * Cc, device=dev)
        if side is not None:
            H.check(L.ramnet_up2x_border_im2col(_p(x), _p(skip), _p(a_rows), _p(a_cols), B, Hh, W, Cc, _st()), "ramnet_up2x_border_im2col")
        else:       # the padded sum and the unrolled border lines in ONE launch
            H.check(L.ramnet_pad2_sum_im2col(_p(x), _p(skip), _p(xpad), _p(a_rows), _p(a_cols), B, Hh, W, Cc, _st()), "ramnet_pad2_sum_im2col")
        g_rows = torch.empty(2, B * W2, 2 * cp.Cout, device=dev)
        g_cols = torch.empty(2, B * H2, 2 * cp.Cout, device=dev)
        gemm2(a_rows, w_rows, g_rows, a_cols, w_cols, g_cols)      # the two sides of both borders in one launch
    if side is not None:
        main.wait_stream(side)
        if not torch.cuda.is_current_stream_capturing():
            for t in (x, skip):
                if t is not None:
                    t.record_stream(side)
            g_rows.record_stream(main), g_cols.record_stream(main)
    desc_kw = dict(bias=cp.bias(), epi=epi, frame=2, e0=g_cols.view(2 * B, H2, 1, 2 * cp.Cout), e1=g_rows.view(2 * B, W2, 1, 2 * cp.Cout))
    if _FOLD_WINO and _fold_wino_ok(Cc, cp.Cout):   # Winograd F(2x2,4x4) over the four parities
        conv_launch(xpad, Taps.get("fold", 4, 0, 0, 0), cp.pack_fold_wino(), y, cp.Cout, Ho=Hh, Wo=W, wino24=True, ws_owner=cp, **desc_kw)
        return xpad
    conv_launch_multi(xpad, cp.pack_fold(), y, cp.Cout,
                      [(Taps.get("fold", 4, 0, py, px), Hh, W, (2, 2, py, px)) for py in range(2) for px in range(2)], **desc_kw)
    return xpad


# Stride-2 5x5 layers (the encoders) as 3x3 stride-1 convolutions of the space-to-depth input on the Winograd kernels
# (DESIGN 3.1d).  RAMNET_S2D=0 / set_space_to_depth(False) keeps the direct stride-2 kernels.
_S2D = True


def set_space_to_depth(on):
    global _S2D
    _S2D = bool(on)


def get_space_to_depth():
    return _S2D


def _s2d_eligible(x, cp, k, stride, up):
    return bool(_S2D and _WINOGRAD and stride == 2 and k == 5 and not up and x.shape[1] % 2 == 0
                and x.shape[2] % 2 == 0 and x.shape[3] == cp.Cin and cp.Cin % 8 == 0 and cp.gates == 1 and len(cp.weights) == 1)


# The space-to-depth view is read / written in place by the Winograd kernels (RAMNET_IN_S2D loader, out_s2d epilogue) when
# the channel count is a power of two >= 32; ops.set_space_to_depth_fused(False) materialises it with ramnet_space_to_depth2 instead.
_S2D_FUSED = True
# ... and skip the Winograd positions that the zero slices of that view annihilate (ramnet_conv_desc.s2d_5x5)
_S2D_SPARSE = True
# ... or run the view on the F(2x4,3x3) kernel (dense) where the library's size heuristics select it (the training batch)
_S2D_2X4 = _os.environ.get("RAMNET_S2D_2X4", "1") != "0"      # (environment: A/B runs of bench.py)


def set_space_to_depth_2x4(on):
    global _S2D_2X4
    _S2D_2X4 = bool(on)


def set_space_to_depth_fused(on):
    global _S2D_FUSED
    _S2D_FUSED = bool(on)


def _s2d_fused(C):
    return bool(_S2D_FUSED and C >= 32 and C & (C - 1) == 0)


def _space_to_depth(x, inverse=False):
    """[B,H,W,C] -> [B,H/2,W/2,4C] (channel = pixel parity major), or back."""
    B, Hh, W, Cc = x.shape
    out = torch.empty((B, 2 * Hh, 2 * W, Cc // 4) if inverse else (B, Hh // 2, W // 2, 4 * Cc), device=x.device)
    full = out if inverse else x
    H.check(H.lib().ramnet_space_to_depth2(_p(x), _p(out), B, full.shape[1], full.shape[2], full.shape[3], int(inverse), _st()),
            "ramnet_space_to_depth2")
    return out


def _fold_eligible(x, cp, k, stride, up):
    return bool(up and k == 5 and stride == 1 and _FOLD_UP and x.shape[1] >= 4 and x.shape[2] >= 4
                and x.shape[3] == cp.Cin)


def _folded_upsample_wgrad(x, skip, dy, y, cp, xpad=None):
    """Backward-weights of the folded upsample-conv: each output parity is a 16-tap convolution of pad2(x + skip), so its weight
    gradient is a 16-tap backward-weights launch on (pad2(x + skip), the parity sub-grid of dy [* ReLU mask]) — 64 instead of
    100 tap-pixels per low-res pixel, plain loads — and the border GEMMs contribute A^T dy_frame; ConvParam._finalize_fold maps
    both back to the 5x5 weights when the autograd engine finishes."""
    L = H.lib()
    B, Hh, W, Cc = x.shape
    H2, W2 = 2 * Hh, 2 * W
    dev = x.device
    w4, wr, wc, bws = cp.grad_ws_fold()
    # The WHOLE backward-weights path of the layer — the recomputed padded sum, the Winograd launch, the unrolled border lines, the
    # frame of dy and the two border-GEMM gradients (~0.3 ms of small, latency-bound launches per layer and time step) — runs on the
    # side stream: nothing of it is needed before the end of the backward pass (round 3; before, only the Winograd launch did).
    with side_work([x, skip, dy, y, xpad], dev):
        a_rows = torch.empty(2, B * W2, 5 * Cc, device=dev)
        a_cols = torch.empty(2, B * H2, 5 * Cc, device=dev)
        if xpad is None:            # not kept by forward (RAMNET_SAVE_XPAD=0): recompute pad2(x + skip), together with the border lines
            xpad = torch.empty(B, Hh + 4, W + 4, Cc, device=dev)
            H.check(L.ramnet_pad2_sum_im2col(_p(x), _p(skip), _p(xpad), _p(a_rows), _p(a_cols), B, Hh, W, Cc, _st()), "ramnet_pad2_sum_im2col")
        else:
            H.check(L.ramnet_up2x_border_im2col(_p(x), _p(skip), _p(a_rows), _p(a_cols), B, Hh, W, Cc, _st()), "ramnet_up2x_border_im2col")
        if _FOLD_WINO_WGRAD and Cc == cp.CinWs and ((Cc % 32 == 0 and cp.Cout % 64 == 0) or
                                                                                 (Cc % 64 == 0 and cp.Cout % 32 == 0)):
            # one launch, all four parities, in the Winograd F(2x2,4x4) domain (csrc/conv_wgrad_wino24.hip)
            wgrad_launch(xpad, Taps.get("fold", 4, 0, 0, 0), dy, cp.grad_ws_fold24(), cp.Cout, gmask=y, dbias=bws, Ho=Hh, Wo=W,
                         gview=(0, 0, 0, 0, H2, W2), wino24=True)
        else:
            for py in range(2):
                for px in range(2):
                    wgrad_launch(xpad, Taps.get("fold", 4, 0, py, px), dy, w4, cp.Cout, gmask=y, dbias=bws, Ho=Hh, Wo=W,
                                 gview=(2, 2, py, px, H2, W2), dw_off=(py * 2 + px) * 16 * cp.CinWs * cp.Cout)
        g_rows = torch.empty(2, B * W2, 2 * cp.Cout, device=dev)
        g_cols = torch.empty(2, B * H2, 2 * cp.Cout, device=dev)
        H.check(L.ramnet_frame_gather(_p(dy), _p(y), _p(g_rows), _p(g_cols), B, H2, W2, cp.Cout, _st()), "ramnet_frame_gather")
        gemm2(a_rows, g_rows, wr, a_cols, g_cols, wc, trans_a=True, accumulate=True)      # w[s] += a[s]^T g[s]: the border GEMMs' weight gradient


# Backward-data of the folded upsample-conv: the adjoint of (four parity convolutions of the replicate-padded input + border GEMMs),
# i.e. conv_wino24_kernel over the parity sub-grids of dy * mask with the flipped filters -> gradient of the padded tensor ->
# replicate-padding adjoint, plus the border GEMMs' adjoint scattered through the bilinear taps.  RAMNET_FOLD_DGRAD=0: direct 5x5 kernel on the full-resolution grid + bilinear adjoint.
_FOLD_DGRAD = True


def set_fold_dgrad(on):
    global _FOLD_DGRAD
    _FOLD_DGRAD = bool(on)


def _fold_dgrad_ok(B, H2, W2, cp):
    return bool(_FOLD_DGRAD and _FOLD_UP and cp.Cout % 16 == 0 and cp.Cin % 64 == 0
                and B * H2 * W2 * cp.Cout * 4 < 2 ** 30)


def _folded_upsample_dgrad(x, dy, y, cp):
    L = H.lib()
    B, Hh, W, Cc = x.shape
    H2, W2 = 2 * Hh, 2 * W
    dev = x.device
    g = dy
    if y is not None:
        g = torch.empty_like(dy)
        H.check(L.ramnet_relu_bwd(_p(dy), _p(y), _p(g), dy.numel(), _st()), "ramnet_relu_bwd")
    dxpad = torch.empty(B, Hh + 4, W + 4, Cc, device=dev)
    conv_launch(g, Taps.get("fold", 4, 0, 0, 0), cp.pack_fold_wino_dgrad(), dxpad, Cc, in_mode=H.IN_PARITY4, wino24=True)
    dx = torch.empty(B, Hh, W, Cc, device=dev)
    H.check(L.ramnet_unpad2_fold(_p(dxpad), _p(dx), B, Hh, W, Cc, _st()), "ramnet_unpad2_fold")
    g_rows = torch.empty(2, B * W2, 2 * cp.Cout, device=dev)
    g_cols = torch.empty(2, B * H2, 2 * cp.Cout, device=dev)
    H.check(L.ramnet_frame_gather(_p(g), None, _p(g_rows), _p(g_cols), B, H2, W2, cp.Cout, _st()), "ramnet_frame_gather")
    wt_rows, wt_cols = cp.border_weights_t()                                      # [2 sides][2*Cout][5*Cin]
    d_rows = torch.empty(2, B * W2, 5 * Cc, device=dev)       # plain product (the reduction is 2 * Cout <= 256 long: nothing to split, no zero-fill)
    d_cols = torch.empty(2, B * H2, 5 * Cc, device=dev)
    gemm2(g_rows, wt_rows, d_rows, g_cols, wt_cols, d_cols)
    H.check(L.ramnet_up2x_border_col2im(_p(d_rows), _p(d_cols), _p(dx), B, Hh, W, Cc, _st()), "ramnet_up2x_border_col2im")
    return dx


class ConvAct(Function):
    """ConvLayer / UpsampleConvLayer (submodules.py:8-35, 69-97): [bilinear x2 of (x [+ skip])] -> KxK conv -> bias -> [ReLU]."""

    @staticmethod
    def forward(ctx, x, skip, w, b, cp, stride, relu, up):
        x, skip = dense(x), dense(skip)
        if skip is not None and skip.shape != x.shape:
            raise RuntimeError("skip connection %s does not match the decoder feature map %s (NHWC)" % (tuple(skip.shape), tuple(x.shape)))
        if x.shape[3] != (cp.Cin + 3) // 4 * 4:
            raise RuntimeError("conv expects %d input channels, got %d" % (cp.Cin, x.shape[3]))
        B, Hh, W, _ = x.shape
        k, pad = cp.k, cp.k // 2
        Hin, Win = (2 * Hh, 2 * W) if up else (Hh, W)
        Ho, Wo = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
        y = torch.empty(B, Ho, Wo, cp.Cout, device=x.device)
        mode = (H.IN_UP2X_SKIP if skip is not None else H.IN_UP2X) if up else H.IN_PLAIN
        epi = H.EPI_RELU if relu else H.EPI_LINEAR
        xpad = None
        ctx.s2d = _s2d_eligible(x, cp, k, stride, up)
        ctx.s2d_fused = ctx.s2d and _s2d_fused(x.shape[3])
        if ctx.s2d_fused:   # ... reading the four parities straight from x
            conv_launch(x, Taps.get("conv_s2d", 3, 1), cp.s2d().fwd(), y, cp.Cout, in_mode=H.IN_S2D, Hin=Hh // 2, Win=W // 2,
                        bias=cp.bias(), epi=epi)
        elif ctx.s2d:       # 5x5 stride 2 == 3x3 stride 1 over the four input parities: Winograd kernels
            x = _space_to_depth(x)
            conv_launch(x, Taps.get("conv_s2d", 3, 1), cp.s2d().fwd(), y, cp.Cout, bias=cp.bias(), epi=epi)
        elif _fold_eligible(x, cp, k, stride, up):
            xpad = _folded_upsample_conv(x, skip, cp, y, epi)      # pad2(x + skip): kept for backward-weights
        else:
            conv_launch(x, Taps.get("conv", k, pad), cp.fwd(), y, cp.Cout, stride=stride, x1=skip, in_mode=mode,
                        Hin=Hin, Win=Win, bias=cp.bias(), epi=epi)
        ctx.cp, ctx.stride, ctx.relu, ctx.up, ctx.mode = cp, stride, relu, up, mode
        # premask_ok: a caller that routes EVERY gradient of y through one fan-in (ops.TimeSplit / TimeFan) may have that fan-in apply the ReLU
        # mask and set premasked (premask_relu_feature): backward then reads dy as a plain operand
        ctx.premask_ok, ctx.premasked = bool(relu), False
        ctx.save_for_backward(x, skip, y, xpad if _SAVE_XPAD else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, skip, y, xpad = ctx.saved_tensors
        cp, stride, relu, up, mode = ctx.cp, ctx.stride, ctx.relu, ctx.up, ctx.mode
        relu = relu and not ctx.premasked           # (dy == dy * (y > 0) already)
        dy = dense(dy)
        if ctx.s2d_fused:   # x is the full-resolution input; the kernels address its space-to-depth view
            sp = cp.s2d()
            ws, bws = sp.grad_ws(wino_ok=True)
            Hl, Wl = x.shape[1] // 2, x.shape[2] // 2
            wgrad_side([x, dy, y], x, Taps.get("conv_s2d", 3, 1), dy, ws, cp.Cout, in_mode=H.IN_S2D, Hin=Hl, Win=Wl,
                       gmask=y if relu else None, dbias=bws)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                conv_launch(dy, Taps.get("dgrad1_s2d", 3, 1), sp.bwd(), dx, sp.Cin, xm=y if relu else None,
                            in_mode=H.IN_RELUMASK if relu else H.IN_PLAIN, Ho=Hl, Wo=Wl, out_s2d=x.shape[3])
            return dx, None, None, None, None, None, None, None
        if ctx.s2d:         # x is the space-to-depth input saved by forward
            sp = cp.s2d()
            ws, bws = sp.grad_ws(wino_ok=True)
            wgrad_side([x, dy, y], x, Taps.get("conv_s2d", 3, 1), dy, ws, cp.Cout, gmask=y if relu else None, dbias=bws)
            dx = None
            if ctx.needs_input_grad[0]:
                gs = torch.empty_like(x)
                conv_launch(dy, Taps.get("dgrad1_s2d", 3, 1), sp.bwd(), gs, sp.Cin, xm=y if relu else None,
                            in_mode=H.IN_RELUMASK if relu else H.IN_PLAIN)
                dx = _space_to_depth(gs, inverse=True)
            return dx, None, None, None, None, None, None, None
        B, Hh, W, _ = x.shape
        k, pad = cp.k, cp.k // 2
        Hin, Win = (2 * Hh, 2 * W) if up else (Hh, W)
        if _fold_eligible(x, cp, k, stride, up):
            dy = dy.contiguous()        # (no-op on the path: the folded kernels' point-wise helpers index dy densely)
            _folded_upsample_wgrad(x, skip, dy, y if relu else None, cp, xpad)
        else:
            ws, bws = cp.grad_ws(wino_ok=(stride == 1 and k == 3 and pad == 1 and mode == H.IN_PLAIN))
            wgrad_side([x, skip, dy, y], x, Taps.get("conv", k, pad), dy, ws, cp.Cout, stride=stride, x1=skip, in_mode=mode,
                       Hin=Hin, Win=Win, gmask=y if relu else None, dbias=bws)
        dx = dskip = None
        if (ctx.needs_input_grad[0] or (skip is not None and ctx.needs_input_grad[1])) and _fold_eligible(x, cp, k, stride, up) \
                and _fold_dgrad_ok(B, Hin, Win, cp):
            dx = _folded_upsample_dgrad(x, dy, y if relu else None, cp)
            return dx, (dx if skip is not None else None), None, None, None, None, None, None
        if ctx.needs_input_grad[0] or (skip is not None and ctx.needs_input_grad[1]):
            gin = torch.empty(B, Hin, Win, cp.Cin, device=x.device)
            gmode = H.IN_RELUMASK if relu else H.IN_PLAIN
            if stride == 1:
                conv_launch(dy, Taps.get("dgrad1", k, pad), cp.bwd(), gin, cp.Cin, xm=y if relu else None, in_mode=gmode)
            else:
                conv_launch_multi(dy, cp.bwd(), gin, cp.Cin,
                                  [(Taps.get("dgrad2", k, pad, py, px), (Hin - py + 1) // 2, (Win - px + 1) // 2, (2, 2, py, px))
                                   for py in range(2) for px in range(2)], xm=y if relu else None, in_mode=gmode)
            if up:
                dx = torch.empty(B, Hh, W, cp.Cin, device=x.device)
                H.check(H.lib().ramnet_upsample2x_bwd(_p(gin), _p(dx), B, Hh, W, cp.Cin, _st()), "ramnet_upsample2x_bwd")
            else:
                dx = gin
            dskip = dx if skip is not None else None
        return dx, dskip, None, None, None, None, None, None


class TConvAct(Function):
    """TransposedConvLayer (submodules.py:38-66): ConvTranspose2d k5 s2 p2 output_padding 1 -> bias -> ReLU.
    The ConvTranspose weight [Cin, Cout, 5, 5] IS the OIHW weight of the stride-2 conv whose backward-data this op
    computes, so forward = 4 sub-pixel launches of the backward-data form, backward-data = a plain stride-2 conv."""

    @staticmethod
    def forward(ctx, x, w, b, cp, relu=True):
        x = dense(x)
        B, Hh, W, _ = x.shape
        Ct = cp.Cin                                    # ConvParam sees (O=Cin_t, I=Cout_t): produced channels = cp.Cin
        y = torch.empty(B, 2 * Hh, 2 * W, Ct, device=x.device)
        conv_launch_multi(x, cp.bwd(), y, Ct, [(Taps.get("dgrad2", 5, 2, py, px), Hh, W, (2, 2, py, px))
                                               for py in range(2) for px in range(2)], bias=cp.bias(),
                          epi=H.EPI_RELU if relu else H.EPI_LINEAR)
        ctx.cp, ctx.relu = cp, relu
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        cp = ctx.cp
        dy = dense(dy)
        B, Hh, W, Cx = x.shape
        taps = Taps.get("conv", 5, 2)
        ws, _ = cp.grad_ws()
        mask, mode = (y, H.IN_RELUMASK) if ctx.relu else (None, H.IN_PLAIN)
        # weight gradient of the stride-2 conv (input = masked dy, output gradient = x); bias gradient = sum of masked dy
        wgrad_launch(dy, taps, x, ws, Cx, stride=2, xm=mask, in_mode=mode)
        if cp.biases[0] is not None:
            dyc = dy.contiguous()
            H.check(H.lib().ramnet_bias_grad(_p(dyc), _p(mask), _p(ensure_grad(cp.biases[0])), B * 4 * Hh * W, cp.Cin, _st()), "bias_grad")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(B, Hh, W, Cx, device=x.device)
            conv_launch(dy, taps, cp.fwd(), dx, Cx, stride=2, xm=mask, in_mode=mode)
        return dx, None, None, None, None


class ResConv(Function):
    """Second half of ResidualBlock (submodules.py:205-214): relu(conv3x3(t) + b + residual)."""

    @staticmethod
    def forward(ctx, t, res, w, b, cp):
        t, res = dense(t), dense(res)
        assert t.shape == res.shape
        y = torch.empty_like(res, memory_format=torch.contiguous_format)
        conv_launch(t, Taps.get("conv", 3, 1), cp.fwd(), y, cp.Cout, bias=cp.bias(), epi=H.EPI_RES_RELU, e0=res)
        ctx.cp = cp
        ctx.save_for_backward(t, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        t, y = ctx.saved_tensors
        cp = ctx.cp
        dy = dense(dy).contiguous()
        dpre = torch.empty_like(y)
        H.check(H.lib().ramnet_relu_bwd(_p(dy), _p(y), _p(dpre), y.numel(), _st()), "ramnet_relu_bwd")
        ws, bws = cp.grad_ws(wino_ok=True)
        wgrad_side([t, dpre], t, Taps.get("conv", 3, 1), dpre, ws, cp.Cout, dbias=bws)
        dt = torch.empty_like(t, memory_format=torch.contiguous_format)
        conv_launch(dpre, Taps.get("dgrad1", 3, 1), cp.bwd(), dt, cp.Cin)
        dres = dpre
        if _USE_SIDE:       # autograd may accumulate IN PLACE into the tensor returned for the residual input while the side
            dres = torch.empty_like(y)      # stream still reads dpre: hand it a buffer of its own
            H.check(H.lib().ramnet_relu_bwd(_p(dy), _p(y), _p(dres), y.numel(), _st()), "ramnet_relu_bwd")
        return dt, dres, None, None, None


def _norm_partial(a, y, act, b, groups):
    """Per (group, slab, channel) partial sums (sum a', sum a' * b) in fp64, a' = a or a * act'(y) (csrc/norm.hip)."""
    B, Hh, W, Cc = b.shape
    npix = B * Hh * W // groups
    L = H.lib()
    nslab = L.ramnet_norm_slabs(groups, npix, Cc)
    part = torch.empty(groups, nslab, Cc, 2, device=b.device, dtype=torch.float64)
    H.check(L.ramnet_norm_partial(_p(a), Cc, _p(y), Cc, act, _p(b), Cc, groups, npix, Cc, nslab, _p(part), _st()), "ramnet_norm_partial")
    return part, nslab, npix


def norm_stats(x, groups):
    """(mean, biased variance) of an NHWC tensor per (group, channel), fp64 [groups][C]: group = the whole batch (1: BatchNorm) or one
    image (B: InstanceNorm)."""
    part, _, npix = _norm_partial(x, None, 0, x, groups)
    s = part.sum(1)
    mean = s[..., 0] / npix
    return mean, (s[..., 1] / npix - mean * mean).clamp_(min=0.0)


class NormAct(Function):
    """BatchNorm2d / InstanceNorm2d [+ residual] [+ ReLU | sigmoid] behind a convolution (submodules.py:29-33, 60-64, 92-96,
    203-214): y = act(x * scale + shift [+ res]) with scale = gamma * rstd, shift = beta - mean * scale [groups][C] from norm_act():
    the statistics of x itself (batch_stats: their dependence on x is part of the gradient) or the layer's running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, mean, rstd, scale, shift, batch_stats, act):
        x, res = dense(x).contiguous(), (dense(res).contiguous() if res is not None else None)
        B, Hh, W, Cc = x.shape
        groups = mean.shape[0]
        y = torch.empty_like(x)
        H.check(H.lib().ramnet_norm_apply(_p(x), Cc, _p(scale), _p(shift), _p(res), Cc, act, _p(y), Cc, groups, B * Hh * W // groups, Cc,
                                          _st()), "ramnet_norm_apply")
        ctx.batch_stats, ctx.act, ctx.has_res = batch_stats, act, res is not None
        ctx.save_for_backward(x, y, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        dy = dense(dy).contiguous()
        Cc = x.shape[3]
        groups, act, L = mean.shape[0], ctx.act, H.lib()
        part, nslab, npix = _norm_partial(dy, y if act else None, act, x, groups)
        c = torch.empty(3, groups, Cc, device=x.device)
        dgb = torch.empty(2, Cc, device=x.device) if gamma is not None else None
        H.check(L.ramnet_norm_finalize_bwd(_p(part), groups, nslab, Cc, npix, _p(mean), _p(rstd), _p(gamma.detach()) if gamma is not None else None,
                                           int(ctx.batch_stats), _p(c[0]), _p(c[1]), _p(c[2]), _p(dgb[0]) if dgb is not None else None,
                                           _p(dgb[1]) if dgb is not None else None, _st()), "ramnet_norm_finalize_bwd")
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        H.check(L.ramnet_norm_bwd(_p(dy), Cc, _p(y) if act else None, Cc, act, _p(x), Cc, _p(c[0]), _p(c[1]), _p(c[2]), _p(dx), Cc, _p(dres), Cc,
                                  groups, npix, Cc, _st()), "ramnet_norm_bwd")
        return dx, (dgb[0] if dgb is not None else None), (dgb[1] if dgb is not None else None), dres, None, None, None, None, None, None


_ACT_CODE = {None: 0, "relu": 1, "sigmoid": 2}


def norm_act(x, layer, act=None, res=None):
    """`layer` (nn.BatchNorm2d | nn.InstanceNorm2d: the parameter / buffer holder with the reference's state_dict keys) applied to the
    NHWC tensor x, then [+ res] and the activation: three launches (partial sums, finalize, apply).  Statistics and running-buffer
    updates follow torch: BatchNorm uses the batch statistics in training mode (running = (1 - m) running + m (mean, UNBIASED
    variance), num_batches_tracked += 1) and the running ones in eval mode; InstanceNorm uses per-image statistics unless it tracks
    running statistics AND is in eval mode, and in training mode feeds the batch mean of its per-image (mean, unbiased variance)
    into the running buffers."""
    x = dense(x).contiguous()
    B, Hh, W, Cc = x.shape
    inst = isinstance(layer, torch.nn.InstanceNorm2d)
    tracked = layer.running_mean is not None
    use_input = layer.training or not tracked
    groups = (B if inst else 1) if use_input else 1
    npix = B * Hh * W // groups
    update = bool(tracked and layer.training)
    if update and npix < 2:
        raise ValueError("Expected more than 1 value per channel when training, got input size %s" % ([B, Cc, Hh, W],))
    if update and layer.momentum is None:
        raise NotImplementedError("cumulative moving average (momentum=None): the reference builds its norm layers with momentum 0.1")
    dev = x.device
    mr = torch.empty(2, groups, Cc, device=dev, dtype=torch.float64)
    ss = torch.empty(2, groups, Cc, device=dev)
    part, nslab = None, 0
    if use_input:
        part, nslab, _ = _norm_partial(x, None, 0, x, groups)
    g, b = layer.weight, layer.bias
    H.check(H.lib().ramnet_norm_finalize(_p(part), groups, nslab, Cc, npix, float(layer.eps), _p(g.detach()) if g is not None else None,
                                         _p(b.detach()) if b is not None else None, _p(layer.running_mean), _p(layer.running_var),
                                         float(layer.momentum or 0.0), int(update), int(not use_input),
                                         _p(layer.num_batches_tracked) if (update and not inst) else None,
                                         _p(mr[0]), _p(mr[1]), _p(ss[0]), _p(ss[1]), _st()), "ramnet_norm_finalize")
    return NormAct.apply(x, g, b, res, mr[0], mr[1], ss[0], ss[1], use_input, _ACT_CODE[act])


class GRUCell(Function):
    """ConvGRU (submodules.py:436-454) as two fused launches: [u|r] = sigmoid(W_ur*[x,h]) and
    h' = h(1-u) + tanh(W_o*[x, h.r]) u (concat, h.r, tanh and the blend never touch HBM separately)."""

    @staticmethod
    def forward(ctx, x, h, wu, bu, wr, br, wo, bo, cp_ur, cp_o, out=None):
        """out: optional NHWC buffer that receives h' (the streaming runtimes write the new state straight into their static
        state buffers instead of copying it there)."""
        x, h = dense(x), dense(h)
        if x.shape != h.shape:       # the reference fails in torch.cat here (e.g. H, W not divisible by 2**num_encoders)
            raise RuntimeError("Sizes of tensors must match except in dimension 1. Expected %s but got %s (input vs state; "
                               "NHWC)" % (tuple(x.shape), tuple(h.shape)))
        B, Hh, W, Cc = x.shape
        taps = Taps.get("conv", 3, 1)
        ur = torch.empty(B, Hh, W, 2 * Cc, device=x.device)
        hr = torch.empty(B, Hh, W, Cc, device=x.device) if (_GRU_HR and Cc % 4 == 0) else None
        if hr is not None:
            conv_launch(x, taps, cp_ur.fwd(), ur, 2 * Cc, x1=h, in_mode=H.IN_CAT, C1=Cc, bias=cp_ur.bias(), epi=H.EPI_SIGMOID_HR, e1=h, o1=hr)
        else:
            conv_launch(x, taps, cp_ur.fwd(), ur, 2 * Cc, x1=h, in_mode=H.IN_CAT, C1=Cc, bias=cp_ur.bias(), epi=H.EPI_SIGMOID)
        hn = torch.empty(B, Hh, W, Cc, device=x.device) if out is None else out
        if out is not None:
            if tuple(out.shape) != (B, Hh, W, Cc) or not out.is_contiguous() or out.dtype != torch.float32:
                raise RuntimeError("GRUCell: `out` must be a contiguous fp32 NHWC buffer of the state's shape")
            # the library writes `out` behind autograd's back: declared as an in-place modification, so that its version counter moves and
            # a cell that saved the slot's previous content for its backward raises instead of differentiating the overwritten state
            ctx.mark_dirty(out)
        need = any(ctx.needs_input_grad)
        o = torch.empty_like(hn) if need else None
        if hr is not None:
            conv_launch(x, taps, cp_o.fwd(), hn, Cc, x1=hr, in_mode=H.IN_CAT, C1=Cc, bias=cp_o.bias(), epi=H.EPI_GRU_BLEND, e0=ur, e1=h, o1=o)
        else:
            conv_launch(x, taps, cp_o.fwd(), hn, Cc, x1=h, xm=ur, xm_off=Cc, in_mode=H.IN_CAT_MUL, C1=Cc, bias=cp_o.bias(),
                        epi=H.EPI_GRU_BLEND, e0=ur, e1=h, o1=o)
        ctx.cps = (cp_ur, cp_o)
        ctx.has_hr = hr is not None
        if need:
            ctx.save_for_backward(*((x, h, ur, o, hr) if hr is not None else (x, h, ur, o)))
        return hn

    @staticmethod
    def backward(ctx, dhn):
        if ctx.has_hr:
            x, h, ur, o, hr = ctx.saved_tensors
        else:
            (x, h, ur, o), hr = ctx.saved_tensors, None
        cp_ur, cp_o = ctx.cps
        B, Hh, W, Cc = x.shape
        npix = B * Hh * W
        dhn = dense(dhn)                  # the [.., C:] half of the next update's [dx | dh] is read in place (ld = 2C)
        L = H.lib()
        dpo = torch.empty_like(o)
        dpur = torch.empty_like(ur)
        dxh = torch.empty(B, Hh, W, 2 * Cc, device=x.device)       # [dx | dh]
        taps, tapsd = Taps.get("conv", 3, 1), Taps.get("dgrad1", 3, 1)
        # stage B (dpr = d(h.r) h r (1-r), dh = dh'(1-u) + d(h.r) r) in the epilogue of the launch that produces d(h.r): stage A leaves
        # dh'(1-u) in the [.., C:] half of dxh (RAMNET_EPI_GRU_BWD; a 64-channel output block must lie in one half)
        fused = _GRU_BWD_FUSED and Cc % 64 == 0
        if fused:
            H.check(L.ramnet_gru_bwd_a2(_p(dhn), _p(ur), _p(o), _p(h), _p(dpo), _p(dpur), _p(dxh, Cc), npix, Cc, ld(dhn), 2 * Cc, _st()), "gru_bwd_a2")
        else:
            dhd = torch.empty_like(o)
            H.check(L.ramnet_gru_bwd_a(_p(dhn), _p(ur), _p(o), _p(h), _p(dpo), _p(dpur), _p(dhd), npix, Cc, ld(dhn), _st()), "gru_bwd_a")
        ws, bws = cp_o.grad_ws(wino_ok=Cc % 32 == 0)
        if hr is not None:
            _wgrad_cell(cp_o, ws, [x, hr, dpo], x, taps, dpo, Cc, x1=hr, in_mode=H.IN_CAT, C1=Cc, dbias=bws)
        else:
            _wgrad_cell(cp_o, ws, [x, h, ur, dpo], x, taps, dpo, Cc, x1=h, xm=ur, xm_off=Cc, in_mode=H.IN_CAT_MUL, C1=Cc, dbias=bws)
        if fused:
            conv_launch(dpo, tapsd, cp_o.bwd(), dxh, 2 * Cc, epi=H.EPI_GRU_BWD, e0=ur, e1=h, o1=dpur)
        else:
            conv_launch(dpo, tapsd, cp_o.bwd(), dxh, 2 * Cc)
            H.check(L.ramnet_gru_bwd_b(_p(dxh), _p(ur), _p(h), _p(dpur), _p(dhd), npix, Cc, _st()), "gru_bwd_b")
        ws, bws = cp_ur.grad_ws(wino_ok=Cc % 32 == 0)
        _wgrad_cell(cp_ur, ws, [x, h, dpur], x, taps, dpur, 2 * Cc, x1=h, in_mode=H.IN_CAT, C1=Cc, dbias=bws)
        conv_launch(dpur, tapsd, cp_ur.bwd(), dxh, 2 * Cc, beta=1.0)
        return dxh[..., :Cc], dxh[..., Cc:], None, None, None, None, None, None, None, None, None


_TIME_BATCH = True


_TIME_BATCH_MAX_DECODES = 0      # decodes per chain of the time-batched forward (0: a whole group)


def set_time_batching(on, max_decodes=None):
    """ERGB2DepthRecurrent.forward batches over TIME what does not depend on the order of the state updates (the event encoders of a
    package as ONE chain at batch K x B, the decodes of a package in groups: model/model.py) — on by default, off for A/B runs and the
    pass-by-pass path's tests.  max_decodes: split a decode group into chains of at most that many measurements (0 = no split)."""
    global _TIME_BATCH, _TIME_BATCH_MAX_DECODES
    _TIME_BATCH = bool(on)
    if max_decodes is not None:
        _TIME_BATCH_MAX_DECODES = int(max_decodes)


def time_batch_max_decodes():
    return _TIME_BATCH_MAX_DECODES


def time_batching():
    return _TIME_BATCH


# A time-batched feature that is a ReLU output gets its ReLU mask where its gradient is summed (TimeSplit / TimeFan backward: one more
# operand in an HBM-bound launch) instead of in the loaders of its layer's backward-data and backward-weights launches (two operands per
# staged slot on the matrix kernels: RAMNET_IN_RELUMASK / gmask).  The same values reach the same products: bit-identical gradients.
_RELU_PREMASK = True


def set_relu_premask(on):
    global _RELU_PREMASK
    _RELU_PREMASK = bool(on)


def get_relu_premask():
    return _RELU_PREMASK


def premask_relu_feature(y):
    """y = the output of a ConvAct with ReLU whose ONLY consumer is the TimeSplit / TimeFan the caller builds next (model/model.py): tell the
    layer's backward that the gradient it receives is already masked.  Returns whether the fan-in should apply the mask."""
    fn = getattr(y, "grad_fn", None)
    if not (_RELU_PREMASK and fn is not None and getattr(fn, "premask_ok", False)):
        return False
    fn.premasked = True
    return True


def _cat_batch_add(grads, base, shape, device, mask=None):
    """cat(grads, 0) [+ base] as ONE launch (ramnet_cat_batch_add): the slices' gradients are channel slices of wider tensors (the
    [dx | dh] of a ConvGRU backward: ld = 2C), `base` the gradient that arrived through the batched consumer."""
    n = len(grads)
    ok = (len(shape) == 4 and n <= 8 and shape[3] % 4 == 0 and all(g is not None and g.is_cuda and g.dim() == 4 and g.dtype == torch.float32 for g in grads)
          and (base is None or base.dtype == torch.float32))
    if ok:
        gs = [dense(g) for g in grads]
        ok = all(ld(g) == ld(gs[0]) and g.stride(1) == g.shape[2] * g.stride(2) and g.stride(0) == g.shape[1] * g.stride(1)
                 and tuple(g.shape) == (shape[0] // n,) + tuple(shape[1:]) for g in gs)
    if not ok:
        filled = [g if g is not None else torch.zeros((shape[0] // n,) + tuple(shape[1:]), device=device) for g in grads]
        out = torch.cat(filled, 0)
        out = out if base is None else out + base
        return out if mask is None else out * (mask > 0).to(out.dtype)
    out = torch.empty(shape, device=device)
    arr = (C.c_void_p * n)(*[g.data_ptr() for g in gs])
    b = None if base is None else base.contiguous()
    npix = shape[1] * shape[2] * (shape[0] // n)
    if mask is not None:
        assert tuple(mask.shape) == tuple(shape) and mask.is_contiguous() and mask.dtype == torch.float32
        H.check(H.lib().ramnet_cat_batch_add_masked(arr, n, npix, shape[3], ld(gs[0]), _p(b), _p(mask), _p(out), _st()), "ramnet_cat_batch_add_masked")
    else:
        H.check(H.lib().ramnet_cat_batch_add(arr, n, npix, shape[3], ld(gs[0]), _p(b), _p(out), _st()), "ramnet_cat_batch_add")
    return out


def arena_slots(n, shape, device):
    """A [n, *shape] fp32 buffer and n tensors over its consecutive slots that are NOT autograd views of it (each has its own version counter):
    a cell that writes slot k (GRUCell / LSTMCell `out=`, declared with mark_dirty) neither rebases the history of the other slots nor
    invalidates what other cells saved of them, and overwriting a slot whose old content a backward still needs raises."""
    buf = torch.empty((n,) + tuple(shape), device=device, dtype=torch.float32)
    numel = buf[0].numel()
    st = buf.untyped_storage()
    slots = [torch.empty(0, device=device, dtype=torch.float32).set_(st, buf.storage_offset() + k * numel, tuple(shape)) for k in range(n)]
    return buf, slots


class TimeSplit(Function):
    """[n * B, ...] -> n views [B, ...] (the features of n measurements that went through a layer chain as one batch); backward
    concatenates the n gradients (ONE launch) instead of autograd's n zero-fills + n slice copies."""

    @staticmethod
    def forward(ctx, x, n, premask=False):
        B = x.shape[0] // n
        ctx.meta = (n, tuple(x.shape), x.device)
        ctx.set_materialize_grads(False)
        ctx.premask = bool(premask) and x.is_contiguous()
        if ctx.premask:
            ctx.save_for_backward(x)          # (the layer that made x keeps it for its own backward anyway: no extra memory)
        return tuple(x[k * B:(k + 1) * B] for k in range(n))

    @staticmethod
    def backward(ctx, *grads):
        n, shape, dev = ctx.meta
        mask = ctx.saved_tensors[0] if ctx.premask else None
        if all(g is None for g in grads):
            return None, None, None
        return _cat_batch_add(list(grads), None, shape, dev, mask), None, None


class TimeFan(Function):
    """TimeSplit for a feature that ALSO feeds the next layer of the batched chain: returns (x itself, n views).  Backward gets the
    gradient of the batched consumer and the n slice gradients in ONE call and forms their sum in one launch — autograd's fan-in add
    (and the concatenation) as library work."""

    @staticmethod
    def forward(ctx, x, n, premask=False):
        B = x.shape[0] // n
        ctx.meta = (n, tuple(x.shape), x.device)
        ctx.set_materialize_grads(False)
        ctx.premask = bool(premask) and x.is_contiguous()
        if ctx.premask:
            ctx.save_for_backward(x)
        return (x.view_as(x),) + tuple(x[k * B:(k + 1) * B] for k in range(n))

    @staticmethod
    def backward(ctx, g_all, *grads):
        n, shape, dev = ctx.meta
        mask = ctx.saved_tensors[0] if ctx.premask else None
        if all(g is None for g in grads):
            if mask is not None and g_all is not None:      # (the promise to the layer holds on every path)
                if g_all.is_cuda:
                    g_all = dense(g_all)
                    out = torch.empty_like(g_all)
                    H.check(H.lib().ramnet_relu_bwd(_p(g_all), _p(mask), _p(out), g_all.numel(), _st()), "ramnet_relu_bwd")
                    g_all = out
                else:                                       # (the autograd plumbing is exercised on CPU tensors by the tests)
                    g_all = g_all * (mask > 0).to(g_all.dtype)
            return g_all, None, None
        return _cat_batch_add(list(grads), g_all, shape, dev, mask), None, None


class TimeJoin(Function):
    """n state tensors [B, h, w, C] that already ARE consecutive slots of one buffer (the cells wrote them there: GRUCell `out=`)
    -> the buffer's [n * B, h, w, C] view, without a copy; backward hands each cell its slice of the batched gradient."""

    @staticmethod
    def forward(ctx, joined, *parts):
        B = parts[0].shape[0]
        step = parts[0].numel() * parts[0].element_size()
        for i, t in enumerate(parts):
            if t.data_ptr() != joined.data_ptr() + i * step or not t.is_contiguous():
                raise RuntimeError("TimeJoin: part %d is not slot %d of the joined buffer" % (i, i))
        ctx.meta = (len(parts), B)
        return joined.view(joined.shape)

    @staticmethod
    def backward(ctx, g):
        n, B = ctx.meta
        g = dense(g)
        return (None,) + tuple(g[i * B:(i + 1) * B] for i in range(n))


class LSTMCell(Function):
    """ConvLSTM (submodules.py:318-358): one launch; the gate non-linearities and the cell update are the epilogue."""

    @staticmethod
    def forward(ctx, x, h, c, w, b, cp, out_h=None, out_c=None):
        x, h, c = dense(x), dense(h), dense(c)
        if x.shape != h.shape or x.shape != c.shape:
            raise RuntimeError("Sizes of tensors must match except in dimension 1. Expected %s but got %s / %s (input vs "
                               "hidden / cell state; NHWC)" % (tuple(x.shape), tuple(h.shape), tuple(c.shape)))
        B, Hh, W, Cc = x.shape
        hn = torch.empty(B, Hh, W, Cc, device=x.device) if out_h is None else out_h
        cn = torch.empty_like(hn) if out_c is None else out_c
        for t in (out_h, out_c):
            if t is not None and (tuple(t.shape) != (B, Hh, W, Cc) or not t.is_contiguous() or t.dtype != torch.float32):
                raise RuntimeError("LSTMCell: `out_h` / `out_c` must be contiguous fp32 NHWC buffers of the state's shape")
        dirty = [t for t in (out_h, out_c) if t is not None]
        if dirty:
            ctx.mark_dirty(*dirty)          # (as in GRUCell: written by the library, declared to autograd)
        need = any(ctx.needs_input_grad)
        gates = torch.empty(B, Hh, W, 4 * Cc, device=x.device) if need else None
        conv_launch(x, Taps.get("conv", 3, 1), cp.fwd(), hn, Cc, x1=h, in_mode=H.IN_CAT, C1=Cc, bias=cp.bias(),
                    epi=H.EPI_LSTM, e1=c, o1=cn, o2=gates)
        ctx.cp = cp
        if need:
            ctx.save_for_backward(x, h, c, cn, gates)
        return hn, cn

    @staticmethod
    def backward(ctx, dhn, dcn):
        x, h, c, cn, gates = ctx.saved_tensors
        cp = ctx.cp
        B, Hh, W, Cc = x.shape
        npix = B * Hh * W
        dhn = None if dhn is None else dense(dhn).contiguous()
        dcn = None if dcn is None else dense(dcn).contiguous()
        dpre = torch.empty_like(gates)
        dc = torch.empty_like(cn)
        H.check(H.lib().ramnet_lstm_bwd(_p(gates), _p(c), _p(cn), _p(dhn), _p(dcn), _p(dpre), _p(dc), npix, Cc, _st()), "lstm_bwd")
        ws, bws = cp.grad_ws(wino_ok=x.shape[3] % 32 == 0)
        wgrad_side([x, h, dpre], x, Taps.get("conv", 3, 1), dpre, ws, 4 * Cc, x1=h, in_mode=H.IN_CAT, C1=Cc, dbias=bws)
        dxh = torch.empty(B, Hh, W, 2 * Cc, device=x.device)
        conv_launch(dpre, Taps.get("dgrad1", 3, 1), cp.bwd(), dxh, 2 * Cc)
        return dxh[..., :Cc], dxh[..., Cc:], dc, None, None, None, None, None


class PredSigmoid(Function):
    """pred = sigmoid(conv1x1(x) + b) (statenet.py:116-117, 313); returns NCHW [B,1,H,W]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = dense(x)
        B, Hh, W, Cc = x.shape
        y = torch.empty(B, 1, Hh, W, device=x.device)
        H.check(H.lib().ramnet_pred_sigmoid_fwd(_p(x), ld(x), Cc, _p(w.detach()), _p(b.detach()), _p(y), B * Hh * W, _st()), "pred_fwd")
        ctx.save_for_backward(x, w, b, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, y = ctx.saved_tensors
        B, Hh, W, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty(B, Hh, W, Cc, device=x.device) if ctx.needs_input_grad[0] else None
        H.check(H.lib().ramnet_pred_sigmoid_bwd(_p(x), ld(x), Cc, _p(w.detach()), _p(y), _p(dy), _p(dx), Cc,
                                                _p(ensure_grad(w)), _p(ensure_grad(b)), B * Hh * W, _st()), "pred_bwd")
        return dx, None, None


_SI_FUSION = True


def set_si_fusion(on):
    """Scale-invariant loss of the supervised predictions inside the prediction layer's launches (PredSigmoidSI; trainer.sequence_loss
    asks the model for it): on by default, off for A/B runs and the separate-launch path's tests."""
    global _SI_FUSION
    _SI_FUSION = bool(on)


def si_fusion():
    return _SI_FUSION


class PredSigmoidSI(Function):
    """PredSigmoid + scale_invariant_loss (model/loss.py:6-9) of the `len(targets)` equal batch segments of its output against their
    target maps, statistics formed in the forward launch and the loss gradient in the backward launch (ramnet_pred_sigmoid_si_fwd / _bwd):
    returns (pred NCHW [B,1,H,W], loss_0, ..., loss_{n-1}).  pred stays differentiable for other consumers (a dense gradient is added)."""

    @staticmethod
    def forward(ctx, x, w, b, weight, n_lambda, mask_x, *targets):
        """mask_x: x is a ReLU output whose every gradient comes from here (premask_relu_feature said yes): dx leaves with that mask applied."""
        x = dense(x)
        B, Hh, W, Cc = x.shape
        n = len(targets)
        assert 1 <= n <= 8 and B % n == 0
        seg_pix = (B // n) * Hh * W
        tg = [t.contiguous() for t in targets]
        for t in tg:
            assert t.is_cuda and t.dtype == torch.float32 and t.numel() == seg_pix, "PredSigmoidSI: one [B/n, 1, H, W] fp32 target per segment"
        L = H.lib()
        y = torch.empty(B, 1, Hh, W, device=x.device)
        scratch = torch.zeros(L.ramnet_pred_si_scratch_doubles(seg_pix, n), device=x.device, dtype=torch.float64)
        stats = torch.empty(n, 4, device=x.device, dtype=torch.float64)
        loss = torch.empty(n, device=x.device)
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in tg])
        H.check(L.ramnet_pred_sigmoid_si_fwd(_p(x), ld(x), Cc, _p(w.detach()), _p(b.detach()), _p(y), seg_pix, n, arr, weight, n_lambda,
                                             _p(scratch), _p(stats), _p(loss), _st()), "pred_si_fwd")
        ctx.save_for_backward(x, w, b, y, stats, scratch, *tg)      # (scratch: the backward launch joins its partial sums through it)
        ctx.meta = (weight, n_lambda, n, seg_pix, bool(mask_x))
        ctx.set_materialize_grads(False)
        return (y,) + tuple(loss[i] for i in range(n))

    @staticmethod
    def backward(ctx, dy, *dloss):
        x, w, b, y, stats, scratch = ctx.saved_tensors[:6]
        tg = ctx.saved_tensors[6:]
        weight, n_lambda, n, seg_pix, mask_x = ctx.meta
        B, Hh, W, Cc = x.shape
        dy = dy.contiguous() if dy is not None else None
        if all(g is None for g in dloss):
            gs = torch.zeros(n, device=x.device)
        else:
            gs = torch.stack([g.float().reshape(()) if g is not None else torch.zeros((), device=x.device) for g in dloss])
        dx = torch.empty(B, Hh, W, Cc, device=x.device) if ctx.needs_input_grad[0] else None
        arr = (C.c_void_p * n)(*[t.data_ptr() for t in tg])
        # the fixed-order join of the weight / bias partial sums runs through the forward's scratch; its layout follows the library options
        # ("pred_si_cap" / "pred_si_bwd_cap"): had they changed since the forward pass, fall back to the atomic form instead of a wrong layout
        join = scratch.numel() == H.lib().ramnet_pred_si_scratch_doubles(seg_pix, n)
        H.check(H.lib().ramnet_pred_sigmoid_si_bwd(_p(x), ld(x), Cc, _p(w.detach()), _p(y), _p(dy), seg_pix, n, arr, _p(stats), _p(gs), weight,
                                                   n_lambda, _p(dx), Cc, _p(ensure_grad(w)), _p(ensure_grad(b)), _p(scratch) if join else None, int(mask_x), _st()), "pred_si_bwd")
        return (dx, None, None, None, None, None) + (None,) * n


class PredLinear(Function):
    """The prediction layer's 1x1 convolution to ONE channel without its sigmoid (a norm layer follows, submodules.py:29-33):
    NHWC [B,H,W,C] -> [B,H,W,1]; the bias is absent under norm 'BN'."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = dense(x)
        B, Hh, W, Cc = x.shape
        z = torch.empty(B, Hh, W, 1, device=x.device)
        H.check(H.lib().ramnet_pred_linear_fwd(_p(x), ld(x), Cc, _p(w.detach()), _p(b.detach()) if b is not None else None, _p(z),
                                               B * Hh * W, _st()), "pred_linear_fwd")
        ctx.save_for_backward(x, w, b)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, w, b = ctx.saved_tensors
        B, Hh, W, Cc = x.shape
        dz = dz.contiguous()
        dx = torch.empty(B, Hh, W, Cc, device=x.device) if ctx.needs_input_grad[0] else None
        H.check(H.lib().ramnet_pred_linear_bwd(_p(x), ld(x), Cc, _p(w.detach()), _p(dz), _p(dx), Cc, _p(ensure_grad(w)),
                                               _p(ensure_grad(b)) if b is not None else None, B * Hh * W, _st()), "pred_linear_bwd")
        return dx, None, None


class SILoss(Function):
    """scale_invariant_loss (model/loss.py:6-9): w * (mean(d^2) - lambda * mean(d)^2) over non-NaN d."""

    @staticmethod
    def forward(ctx, pred, target, weight, n_lambda):
        pred, target = pred.contiguous(), target.contiguous()
        stats = torch.empty(4, device=pred.device, dtype=torch.float64)      # sums + the reduction's arrival ticket
        loss = torch.empty((), device=pred.device)
        H.check(H.lib().ramnet_si_loss_fwd(_p(pred), _p(target), pred.numel(), weight, n_lambda, _p(stats), _p(loss), _st()), "si_fwd")
        ctx.save_for_backward(pred, target, stats)
        ctx.wl = (weight, n_lambda)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target, stats = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(pred)
        H.check(H.lib().ramnet_si_loss_bwd(_p(pred), _p(target), pred.numel(), ctx.wl[0], ctx.wl[1], _p(stats), _p(g), _p(d), _st()), "si_bwd")
        return d, None, None, None


def si_local_stats(pred, target, out_row):
    """(sum d, sum d^2, n) of ONE supervised map into out_row (4 doubles of a caller-owned table; [3] is the reduction's scratch):
    the per-rank half of the exact data-parallel SI loss.  Not differentiated — SILossFromStats carries the gradient."""
    pred, target = pred.contiguous(), target.contiguous()
    dummy = torch.empty((), device=pred.device)
    H.check(H.lib().ramnet_si_loss_fwd(_p(pred), _p(target), pred.numel(), 1.0, 1.0, _p(out_row), _p(dummy), _st()), "si_fwd")


class SILossFromStats(Function):
    """scale_invariant_loss (model/loss.py:6-9) of the GLOBAL batch from all-reduced statistics: loss = w (S2/n - lambda (S1/n)^2);
    d loss / d pred_i = w (2 d_i / n - 2 lambda S1 / n^2) with the global n, S1 — this rank's share of the single-process gradient on
    the concatenated batch.  `gain` = world size: the reducer AVERAGES gradients over ranks, and the shares have to ADD UP."""

    @staticmethod
    def forward(ctx, pred, target, stats, weight, n_lambda, gain):
        pred, target = pred.contiguous(), target.contiguous()
        loss = torch.empty((), device=pred.device)
        H.check(H.lib().ramnet_si_loss_from_stats(_p(stats), weight, n_lambda, _p(loss), _st()), "si_from_stats")
        ctx.save_for_backward(pred, target, stats)
        ctx.wl = (weight * gain, n_lambda)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target, stats = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(pred)
        H.check(H.lib().ramnet_si_loss_bwd(_p(pred), _p(target), pred.numel(), ctx.wl[0], ctx.wl[1], _p(stats), _p(g), _p(d), _st()), "si_bwd")
        return d, None, None, None, None, None


class SILogLoss(Function):
    """scale_invariant_log_loss (model/loss.py:12-15): mean(d^2) - lambda * mean(d)^2 over non-NaN d = log(pred) - log(target)."""

    @staticmethod
    def forward(ctx, pred, target, n_lambda):
        pred, target = pred.contiguous(), target.contiguous()
        stats = torch.empty(4, device=pred.device, dtype=torch.float64)
        loss = torch.empty((), device=pred.device)
        H.check(H.lib().ramnet_si_log_loss_fwd(_p(pred), _p(target), pred.numel(), n_lambda, _p(stats), _p(loss), _st()), "si_log_fwd")
        ctx.save_for_backward(pred, target, stats)
        ctx.n_lambda = n_lambda
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target, stats = ctx.saved_tensors
        g = g.contiguous().float()
        d = torch.empty_like(pred)
        H.check(H.lib().ramnet_si_log_loss_bwd(_p(pred), _p(target), pred.numel(), ctx.n_lambda, _p(stats), _p(g), _p(d), _st()), "si_log_bwd")
        return d, None, None


class MSELoss(Function):
    """mse_loss (model/loss.py:18-19) over the non-NaN target entries; half: both maps first through the bilinear x0.5 resize of
    lstm_trainer.py:173-181 (fused into the reduction: no half-resolution tensors exist)."""

    @staticmethod
    def forward(ctx, pred, target, half):
        pred, target = pred.contiguous(), target.contiguous()
        assert pred.dim() == 4 and pred.shape[1] == 1 and pred.shape == target.shape, "mse_loss: [B, 1, H, W] maps"
        B, _, Hh, W = pred.shape
        stats = torch.empty(4, device=pred.device, dtype=torch.float64)
        loss = torch.empty((), device=pred.device)
        H.check(H.lib().ramnet_mse_loss_fwd(_p(pred), _p(target), B, Hh, W, int(half), _p(stats), _p(loss), _st()), "mse_fwd")
        ctx.save_for_backward(pred, target, stats)
        ctx.half = int(half)
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target, stats = ctx.saved_tensors
        B, _, Hh, W = pred.shape
        g = g.contiguous().float()
        d = torch.empty_like(pred)
        H.check(H.lib().ramnet_mse_loss_bwd(_p(pred), _p(target), B, Hh, W, ctx.half, _p(stats), _p(g), _p(d), _st()), "mse_bwd")
        return d, None, None


def scale_invariant_log_loss(y_input, y_target, n_lambda=1.0):
    """Drop-in for model.loss.scale_invariant_log_loss (model/loss.py:12-15) on device tensors."""
    return SILogLoss.apply(y_input.float(), y_target.to(y_input.device).float(), float(n_lambda))


def mse_loss(y_input, y_target, downsampling_factor=1.0):
    """Drop-in for model.loss.mse_loss (model/loss.py:18-19); downsampling_factor 0.5 reproduces the trainer's half-resolution form
    (lstm_trainer.py:169-185: both maps through F.interpolate(scale_factor=0.5, 'bilinear', align_corners=False) first).  Other factors
    are refused: no shipped configuration sets one."""
    if downsampling_factor not in (1.0, 0.5):
        raise NotImplementedError("mse_loss: downsampling_factor %r (1.0 and the trainer's default 0.5 are built)" % (downsampling_factor,))
    return MSELoss.apply(y_input.float(), y_target.to(y_input.device).float(), downsampling_factor == 0.5)


class MSGLoss(Function):
    """multi_scale_grad_loss (model/loss.py:22-70): 4 average-pool scales x Sobel gradient, NaN-masked L1.
    The Sobel arithmetic lives in kornia 0.4.0 (absent here): parity is pinned to the restatement in the oracle only."""

    @staticmethod
    def forward(ctx, pred, target, num_scales):
        pred, target = pred.contiguous(), target.contiguous()
        B, _, Hh, W = pred.shape
        L = H.lib()
        n = L.ramnet_msg_workspace_elems(B, Hh, W, num_scales)
        ws = torch.empty(n, device=pred.device)
        stats = torch.empty(2 * num_scales, device=pred.device, dtype=torch.float64)
        loss = torch.empty((), device=pred.device)
        H.check(L.ramnet_msg_loss_fwd(_p(pred), _p(target), B, Hh, W, num_scales, _p(ws), _p(stats), _p(loss), _st()), "msg_fwd")
        ctx.save_for_backward(ws, stats)
        ctx.dims = (B, Hh, W, num_scales)
        return loss

    @staticmethod
    def backward(ctx, g):
        ws, stats = ctx.saved_tensors
        B, Hh, W, ns = ctx.dims
        g = g.contiguous().float()
        dws = torch.empty_like(ws)
        d = torch.empty(B, 1, Hh, W, device=ws.device)
        H.check(H.lib().ramnet_msg_loss_bwd(_p(ws), _p(stats), _p(g), B, Hh, W, ns, _p(dws), _p(d), _st()), "msg_bwd")
        return d, None, None


def multi_scale_grad_loss(prediction, target, num_scales=4):
    """Drop-in for model.loss.multi_scale_grad_loss (non-preview branch) on device tensors."""
    return MSGLoss.apply(prediction.float(), target.to(prediction.device).float(), int(num_scales))


def nhwc_add(a, b):
    """a + b on NHWC tensors (UNet head skip, unet.py:129)."""
    a, b = dense(a).contiguous(), dense(b).contiguous()
    y = torch.empty_like(a)
    H.check(H.lib().ramnet_add(_p(a), _p(b), _p(y), a.numel(), _st()), "ramnet_add")
    return y


def scale_invariant_loss(y_input, y_target, weight=1.0, n_lambda=1.0):
    """Drop-in for model.loss.scale_invariant_loss (model/loss.py:6-9) on device tensors."""
    return SILoss.apply(y_input.float(), y_target.to(y_input.device).float(), float(weight), float(n_lambda))


class Add(Function):
    """Skip sum x1 + x2 (unet.py:14-15) as a HIP kernel; gradient fans out unchanged."""

    @staticmethod
    def forward(ctx, a, b):
        return nhwc_add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class Concat(Function):
    """Skip concatenation torch.cat([x1, x2], dim=1) (unet.py:11-13) on NHWC tensors; the gradient is the two channel slices
    (written dense: the decoders' backward kernels take contiguous gradients)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = dense(a), dense(b)
        assert a.shape[:3] == b.shape[:3]
        B, Hh, W, Ca = a.shape
        Cb = b.shape[3]
        y = torch.empty(B, Hh, W, Ca + Cb, device=a.device)
        H.check(H.lib().ramnet_concat2(_p(a), ld(a), Ca, _p(b), ld(b), Cb, _p(y), B * Hh * W, _st()), "ramnet_concat2")
        ctx.c = (Ca, Cb)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dense(dy)
        B, Hh, W, _ = dy.shape
        Ca, Cb = ctx.c
        da, db = torch.empty(B, Hh, W, Ca, device=dy.device), torch.empty(B, Hh, W, Cb, device=dy.device)
        H.check(H.lib().ramnet_split2(_p(dy), ld(dy), Ca, Cb, _p(da), _p(db), B * Hh * W, _st()), "ramnet_split2")
        return da, db
