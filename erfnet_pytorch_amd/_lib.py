"""ctypes binding of ``liberfnet_hip.so`` (C ABI declared in ``include/erfnet_hip.h``).

The library is the only compute path for tensors that live on a GPU: when it cannot be loaded
every GPU call raises, there is no fallback.  ``build()`` compiles it in-tree with hipcc for gfx950
(cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess
import threading

# torch must come first: it brings the HIP runtime (its bundled libamdhip64) that owns the tensors'
# device memory and streams; liberfnet_hip.so then binds to that same runtime instead of pulling in
# a second copy from /opt/rocm.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# ERFNET_HIP_SO selects another build of the same library (kernel experiments); default in-tree.
_SO = os.environ.get("ERFNET_HIP_SO") or os.path.join(_HERE, "liberfnet_hip.so")
_SRC = os.path.join(_HERE, "csrc")
_lock = threading.Lock()
_lib = None
_load_error = None

vp = C.c_void_p
ll = C.c_longlong


class BnStruct(C.Structure):
    """Mirror of ``erf_bn_t``."""
    _fields_ = [("mode", C.c_int), ("gamma", vp), ("beta", vp), ("running_mean", vp),
                ("running_var", vp), ("save_mean", vp), ("save_invstd", vp),
                ("momentum", C.c_float), ("eps", C.c_float), ("num_batches_tracked", vp)]


class Nb1dBlockStruct(C.Structure):
    """Mirror of ``erf_nb1d_block_t``."""
    _fields_ = [("w", vp * 4), ("b", vp * 4), ("bn1", C.POINTER(BnStruct)), ("bn2", C.POINTER(BnStruct)),
                ("drop_scale", vp), ("dil", C.c_int), ("y", vp), ("save", C.POINTER(vp)),
                ("sign_bits", C.POINTER(vp)), ("sign_bits_written", C.POINTER(C.c_uint))]


# No packed fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32): on gfx950 one such instruction
# issued behind a v_mfma_f32_32x32x16_bf16 holds the next MFMA of that SIMD back by about 16 cycles
# (scripts/ubench/bf16_mfma_rate.hip: 33 -> 49.5 cycles per MFMA with one of them per MFMA, whereas four
# plain v_fma_f32 / v_sub_f32 in the same place cost one cycle), and the line kernels place their fp32
# splits and epilogues exactly there.  The element-wise kernels are bound by memory either way.
DEVICE_FLAGS = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def sources():
    return [os.path.join(_SRC, f) for f in sorted(os.listdir(_SRC))] + \
        [os.path.join(os.path.dirname(_HERE), "include", "erfnet_hip.h")]


def build(force=False, verbose=False):
    """Compile csrc/erfnet_hip.hip -> liberfnet_hip.so for gfx950 if it is missing or stale."""
    srcs = sources()
    if not force and os.path.exists(_SO) and \
            os.path.getmtime(_SO) >= max(os.path.getmtime(s) for s in srcs):
        return _SO
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *DEVICE_FLAGS,
           "-o", _SO, os.path.join(_SRC, "erfnet_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


_PROTOS = {
    "erf_version": (C.c_int, []),
    "erf_last_error": (C.c_char_p, []),
    "erf_launch_count": (C.c_ulonglong, []),
    "erf_batch_begin": (C.c_int, []),
    "erf_batch_pack": (C.c_int, [vp]),
    "erf_batch_end": (C.c_int, [vp]),
    "erf_batch_abort": (None, []),
    "erf_set_option": (C.c_int, [C.c_int, C.c_int]),
    "erf_workspace_bytes": (C.c_size_t, [ll, C.c_int]),
    "erf_downsampler_forward": (C.c_int, [vp, ll, ll, ll, ll, vp, vp, C.POINTER(BnStruct), vp, vp,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                          C.c_size_t, vp]),
    "erf_downsampler_backward": (C.c_int, [vp, ll, ll, ll, ll, vp, C.POINTER(BnStruct), vp, vp, vp,
                                           vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_nb1d_forward": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(BnStruct),
                                   C.POINTER(BnStruct), vp, C.c_int, vp, C.POINTER(vp), C.c_int,
                                   C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_nb1d_chain_forward": (C.c_int, [vp, C.POINTER(Nb1dBlockStruct), C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, vp, C.c_size_t, vp]),
    "erf_conv_pair_pack": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "erf_conv_line_pack_split": (C.c_int, [vp, C.c_int, vp, vp]),
    "erf_conv_line_forward_split": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int,
                                              C.c_int, C.c_int, vp]),
    "erf_conv_pair_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                        vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "erf_nb1d_backward": (C.c_int, [vp, C.POINTER(vp), C.POINTER(BnStruct), C.POINTER(BnStruct), vp,
                                    C.c_int, vp, C.POINTER(vp), vp, vp, C.POINTER(vp),
                                    C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_nb1d_backward_bits": (C.c_int, [vp, C.POINTER(vp), C.POINTER(BnStruct), C.POINTER(BnStruct), vp,
                                         C.c_int, vp, C.POINTER(vp), C.POINTER(vp), vp, vp, C.POINTER(vp),
                                         C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_upsampler_forward": (C.c_int, [vp, vp, vp, C.POINTER(BnStruct), vp, vp, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_upsampler_backward": (C.c_int, [vp, vp, C.POINTER(BnStruct), vp, vp, vp, vp, vp, vp, vp, vp,
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                         C.c_size_t, vp]),
    "erf_outconv_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, vp]),
    "erf_outconv_argmax": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "erf_outconv_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_outconv_ce_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, ll, C.c_int, vp, C.c_size_t, vp]),
    "erf_outconv_ce_forward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ll,
                                         vp, C.c_size_t, vp]),
    "erf_conv2d_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_conv2d_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_cross_entropy_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, ll, vp,
                                            C.c_size_t, vp]),
    "erf_cross_entropy_forward_counts": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, ll, vp,
                                                   C.c_size_t, vp]),
    "erf_cross_entropy_forward_static": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, ll, vp]),
    "erf_accuracy_counts": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]),
    "erf_iou_add_batch": (C.c_int, [vp, vp, ll, C.c_int, C.c_int, vp, vp, C.c_size_t, vp]),
    "erf_masked_loss_forward": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_float, vp, C.c_int, C.c_int, C.c_int,
                                          C.c_int, vp, C.c_size_t, vp]),
    "erf_masked_loss_backward": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_float, vp, vp, vp, vp, vp, C.c_int,
                                           C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]),
    "erf_mse_mean_forward": (C.c_int, [vp, vp, ll, vp, vp, C.c_size_t, vp]),
    "erf_mse_mean_backward": (C.c_int, [vp, vp, vp, vp, ll, vp]),
    "erf_cross_entropy_backward": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                             C.c_int, ll, vp]),
    "erf_adam_step": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, ll, vp]),
    "erf_sgd_step": (C.c_int, [C.c_int, vp, vp, vp, vp, C.c_double, C.c_double, C.c_double, vp]),
    "erf_ema_update": (C.c_int, [C.c_int, vp, vp, vp, C.c_double, vp]),
}

# bf16 variants: same argument lists (activation pointers are bf16 tensors, parameters stay fp32)
for _n in ("erf_downsampler_forward", "erf_downsampler_backward", "erf_nb1d_forward", "erf_nb1d_chain_forward",
           "erf_nb1d_backward", "erf_nb1d_backward_bits", "erf_upsampler_forward", "erf_upsampler_backward", "erf_outconv_argmax",
           "erf_outconv_backward", "erf_outconv_ce_backward", "erf_outconv_ce_forward", "erf_conv2d_forward", "erf_conv2d_backward", "erf_conv_pair_pack",
           "erf_conv_pair_forward"):
    _PROTOS[_n + "_bf16"] = _PROTOS[_n]
# erf_outconv_forward_bf16 has one more argument: y_is_bf16 behind y
_PROTOS["erf_outconv_forward_bf16"] = (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, vp])

BWD_GY_PREMASKED, BWD_MASK_DX = 1, 2      # include/erfnet_hip.h
OPT_PAIR_FUSION, OPT_WGRAD_LINE, OPT_STREAM16, OPT_PAIR16, OPT_LINE2, OPT_SPLIT_BF16, OPT_CONV16_ROWS, OPT_X3_ROLL = 1, 2, 3, 4, 5, 6, 7, 8
OPT_POOL_FUSE = 9

EXPORTS = tuple(_PROTOS)
ABI_VERSION = 105                         # ERF_ABI_VERSION of include/erfnet_hip.h


def lib():
    """The loaded library; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib, _load_error
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_SO):
            _load_error = (f"{_SO} is missing: build it with `python -c 'import __graft_entry__ "
                           f"as g; g.build()'` (hipcc --offload-arch=gfx950)")
            raise RuntimeError(_load_error)
        try:
            bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
            if os.path.exists(bundled):
                C.CDLL(bundled, mode=C.RTLD_GLOBAL)
            handle = C.CDLL(_SO)
            for name, (res, args) in _PROTOS.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
        except (OSError, AttributeError) as e:
            _load_error = f"cannot load {_SO}: {e}"
            raise RuntimeError(_load_error) from e
        have = int(handle.erf_version())
        if have != ABI_VERSION:
            _load_error = (f"{_SO} has ABI revision {have}, this package expects {ABI_VERSION}: rebuild it "
                           f"(python -c 'import __graft_entry__ as g; g.build()')")
            raise RuntimeError(_load_error)
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError("erfnet_hip: " + lib().erf_last_error().decode("utf-8", "replace"))


def launch_count():
    return int(lib().erf_launch_count())


def set_option(option, value):
    """Tuning switch of the library (include/erfnet_hip.h); returns the previous value."""
    return int(lib().erf_set_option(int(option), int(value)))
