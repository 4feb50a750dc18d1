"""ctypes binding of ``libvirnet_hip.so`` (the C ABI declared in ``include/virnet_hip.h``).

There is no CPU fallback: if the library is missing or a call fails this module raises.
``import torch`` happens first on purpose -- PyTorch-ROCm ships its own ``libamdhip64.so.7``; loading it first
makes our library bind to the same HIP runtime instance, so torch's device pointers and stream handles are
valid inside our launches.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# VIRNET_HIP_LIB lets a tuning run point at another in-tree build of the same ABI (A/B kernel experiments)
LIB_PATH = os.environ.get("VIRNET_HIP_LIB") or os.path.join(_HERE, "lib", "libvirnet_hip.so")
ABI_VERSION = 4          # include/virnet_hip.h: VIRNET_ABI_VERSION

c_float_p = C.POINTER(C.c_float)


class ConvPlan(C.Structure):
    _fields_ = [("nrep", C.c_int), ("n_pad", C.c_int), ("cin_pad", C.c_int)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wpack", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p),
        ("mul", C.c_void_p), ("add", C.c_void_p), ("mask", C.c_void_p), ("in_mul", C.c_void_p), ("in_add", C.c_void_p),
        ("y_raw", C.c_void_p), ("y_act", C.c_void_p),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("cin_pad", C.c_int), ("cout", C.c_int),
        ("n_pad", C.c_int), ("nrep", C.c_int), ("ks", C.c_int), ("stride", C.c_int), ("epi", C.c_int),
        ("nchw_op", C.c_int), ("crop_h", C.c_int), ("crop_w", C.c_int), ("res_sf", C.c_int),
        ("in_act", C.c_int), ("in_slope", C.c_float), ("slope", C.c_float), ("mask_slope", C.c_float), ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
    ]


class TEmit(C.Structure):
    _fields_ = [("t_out", C.c_void_p), ("col", C.c_void_p), ("act", C.c_int), ("slope", C.c_float), ("bf16", C.c_int), ("rows", C.c_int)]


class PackDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("vec", C.c_void_p), ("map", C.c_void_p), ("out", C.c_void_p),
        ("n", C.c_int), ("c0", C.c_int), ("h", C.c_int), ("w", C.c_int), ("sf", C.c_int),
        ("ev", C.c_int),
        ("em", C.c_int), ("mh", C.c_int), ("mw", C.c_int), ("msf", C.c_int), ("map_sqrt", C.c_int),
        ("hp", C.c_int), ("wp", C.c_int), ("zero_pad", C.c_int),
    ]


class ThinDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("wpack", C.c_void_p), ("bias", C.c_void_p), ("res", C.c_void_p), ("y", C.c_void_p),
        ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("c", C.c_int), ("cout", C.c_int),
        ("crop_h", C.c_int), ("crop_w", C.c_int), ("op", C.c_int), ("res_sf", C.c_int),
        ("clamp_lo", C.c_float), ("clamp_hi", C.c_float),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("dy", C.c_void_p), ("in_mul", C.c_void_p), ("in_add", C.c_void_p), ("dw", C.c_void_p),
        ("counters", C.c_void_p), ("n", C.c_int), ("h", C.c_int), ("w", C.c_int), ("cx", C.c_int), ("cy", C.c_int),
        ("cin", C.c_int), ("cout", C.c_int), ("ks", C.c_int), ("stride", C.c_int), ("transposed", C.c_int),
        ("in_act", C.c_int), ("in_slope", C.c_float),
    ]


class KnetLayer(C.Structure):
    """virnet_knet_layer (include/virnet_hip.h)"""
    _fields_ = [(k, C.c_void_p) for k in ("w1pack", "b1", "w2pack", "b2", "caw1", "cab1", "caw2", "cab2")]


class SftWeights(C.Structure):
    _fields_ = [
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("wm", C.c_void_p), ("bm", C.c_void_p), ("wa", C.c_void_p), ("ba", C.c_void_p),
        ("e", C.c_int), ("nf1", C.c_int), ("nf2", C.c_int), ("nf", C.c_int),
    ]


EPI_NHWC, EPI_CONVT, EPI_NCHW = 0, 1, 2
GAP_MEAN, GAP_EXPCLAMP, GAP_KINFO = 0, 1, 2
NCHW_PLAIN, NCHW_ADD, NCHW_EXPCLAMP = 0, 1, 2

# every symbol include/virnet_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("virnet_abi_version", C.c_int, []),
    ("virnet_last_error", C.c_char_p, []),
    ("virnet_device_count", C.c_int, []),
    ("virnet_packed_weight_floats", C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    ("virnet_pack_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    ("virnet_conv_get_plan", C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(ConvPlan)]),
    ("virnet_conv_mfma", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("virnet_conv_mfma_variant", C.c_int, [C.POINTER(ConvDesc), C.POINTER(C.c_int * 4)]),
    ("virnet_wino_weight_floats", C.c_size_t, [C.c_int, C.c_int]),
    ("virnet_pack_wino_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv_wino", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("virnet_f16_weight_floats", C.c_size_t, [C.c_int, C.c_int]),
    ("virnet_pack_f16_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv_f16", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("virnet_wx4_weight_floats", C.c_size_t, [C.c_int, C.c_int]),
    ("virnet_pack_wx4_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv_wx4", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("virnet_conv_wx4_last_plan", None, [C.POINTER(C.c_int)]),
    ("virnet_exit_weight_floats", C.c_size_t, [C.c_int]),
    ("virnet_pack_exit_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv_exit", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("virnet_set_range_flag", C.c_int, [C.c_void_p]),
    ("virnet_poison_on_flag", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("virnet_pack_bf16_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv_bf16", C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    ("virnet_f16_convt_weight_floats", C.c_size_t, [C.c_int, C.c_int]),
    ("virnet_pack_f16_convt_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_pack_input", C.c_int, [C.POINTER(PackDesc), C.c_void_p]),
    ("virnet_conv_f16_entry", C.c_int, [C.POINTER(ConvDesc), C.POINTER(PackDesc), C.c_void_p]),
    ("virnet_entry_weight_floats", C.c_size_t, [C.c_int, C.c_int]),
    ("virnet_pack_entry_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv_entry", C.c_int, [C.POINTER(ConvDesc), C.POINTER(PackDesc), C.c_void_p]),
    ("virnet_thin_weight_floats", C.c_size_t, [C.c_int]),
    ("virnet_pack_thin_weight", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    ("virnet_conv3x3_thin", C.c_int, [C.POINTER(ThinDesc), C.c_void_p]),
    ("virnet_conv_wgrad", C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    ("virnet_sft_backward", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_long, C.c_int, C.c_void_p]),
    ("virnet_chsplit_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("virnet_conv_wgrad_f16_scratch_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("virnet_chsplit_colsum_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("virnet_chsplit", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("virnet_conv_wgrad_f16", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_void_p]),
    ("virnet_chsplit_s2_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("virnet_chsplit_s2_colsum_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    ("virnet_chsplit_s2", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("virnet_conv_wgrad_f16_s2_scratch_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    ("virnet_conv_wgrad_f16_s2", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_conv_emit_ok", C.c_int, [C.POINTER(ConvDesc), C.c_int, C.POINTER(C.c_int)]),
    ("virnet_conv_f16_emit", C.c_int, [C.POINTER(ConvDesc), C.POINTER(TEmit), C.c_int, C.c_void_p]),
    ("virnet_conv_wx4_emit", C.c_int, [C.POINTER(ConvDesc), C.POINTER(TEmit), C.c_void_p]),
    ("virnet_colpart_reduce", C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_conv_wgrad_f16_db", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]),
    ("virnet_colsum", C.c_int, [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_zero_stuff2", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_space_to_depth2", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_pack_input_backward", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_conv_head_s4", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    ("virnet_conv_head_s4_wgrad", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p]),
    ("virnet_gap_nchw", C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_float, C.c_void_p]),
    ("virnet_ca_gate", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_knet_body", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_scale_add", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    ("virnet_ca_scale_add", C.c_int, [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]),
    ("virnet_sft_vec", C.c_int, [C.c_void_p, C.POINTER(SftWeights), C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("virnet_sft_vec_multi", C.c_int, [C.c_void_p, C.POINTER(SftWeights), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_void_p]),
    ("virnet_sft_apply", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(SftWeights), C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_void_p]),
]

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load (once) and type the library; raise loudly when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the VIRNet HIP kernels are not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C virnet_amd/csrc`) -- there is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.virnet_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{LIB_PATH}: ABI {lib.virnet_abi_version()} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().virnet_last_error()
        raise RuntimeError(f"libvirnet_hip {what}: {msg.decode() if msg else 'unknown error'}")


def ptr(t) -> int:
    """Device address of a tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()


tls = threading.local()      # ops.forward_scope keeps its per-forward snapshot here (stream handle, environment knobs): one per host thread
# Held by every hipGraph capture of this package (graph.py) and by the few host calls that HIP does not permit while ANY stream of the
# process is capturing -- pinned-memory allocation, destruction of a captured graph -- even from another thread in thread-local capture
# mode (tools/probes/capture_concurrency.py): one of those at the wrong moment invalidates the other thread's capture.
capture_lock = threading.RLock()


def stream_handle() -> int:
    """hipStream_t of torch's current stream, as the void* the C ABI takes."""
    cached = getattr(tls, "stream_cache", None)
    if cached is not None:
        return cached
    return torch.cuda.current_stream().cuda_stream
