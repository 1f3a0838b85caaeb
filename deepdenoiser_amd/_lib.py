"""ctypes binding of libdd_hip.so (the C-ABI declared in include/dd_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the compute
layers raises.  (Build it with `python -m deepdenoiser_amd.build`.)
"""

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DD_LIB", os.path.join(_HERE, "libdd_hip.so"))   # DD_LIB: experiment builds (tools/)

DD_F32, DD_BF16, DD_F16 = 0, 1, 2
IN_RELU, OUT_RELU, ACCUM, PIXSHUF, GATHER2X2 = 1, 2, 4, 8, 16
MAX_FEATURES, MAX_COMBINED = 32, 8

# every symbol include/dd_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "dd_version", "dd_last_error", "dd_pack_weights", "dd_pack_weights_batched", "dd_conv_igemm", "dd_conv_wgrad", "dd_colsum", "dd_colsum_segments",
    "dd_maxpool_fwd", "dd_maxpool_bwd", "dd_avgpool", "dd_prepare_feature", "dd_gather_input",
    "dd_kpcn_fwd", "dd_kpcn_bwd", "dd_kpcn_hidden_fwd", "dd_kpcn_hidden_bwd", "dd_compose_pack", "dd_compose_blend_fwd", "dd_compose_blend_bwd",
    "dd_compose_unpack_bwd", "dd_invert_std_fwd", "dd_invert_std_bwd", "dd_loss_head", "dd_adam_step",
    "dd_stitch", "dd_recombine", "dd_probe_tr16", "dd_masked_add", "dd_zero_stuff", "dd_zero_unstuff", "dd_convert_channels",
    "dd_augment", "dd_loss_mask_sums", "dd_crc32c", "dd_extract_tiles", "dd_compose_net_fwd", "dd_compose_net_bwd",
    "dd_kpcn_head_fwd", "dd_kpcn_head_bwd", "dd_kpcn_head_bwd_multi", "dd_assemble_input", "dd_assemble_input_frames", "dd_conv3x3_bwd", "dd_conv3x3_bwd_multi", "dd_convt2x2_fwd", "dd_convt2x2_bwd", "dd_conv3x3_ks",
    "dd_conv_pw_count", "dd_wgrad_pw_count", "dd_space_to_depth2", "dd_convt3_wgrad", "dd_compose_stream_plan", "dd_compose_bwd_scratch_bytes",
)


class ConvT3WgradArgs(C.Structure):
    """dd_convt3_wgrad_args (include/dd_hip.h)."""
    _fields_ = [("s", C.c_void_p), ("lds", C.c_int), ("cout", C.c_int), ("cp", C.c_int),
                ("x", C.c_void_p), ("ldx", C.c_int), ("cin", C.c_int),
                ("dk", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("cin", C.c_int),
                ("wp", C.c_void_p), ("k_pad", C.c_int), ("n_pad", C.c_int),
                ("bias", C.c_void_p), ("nbias", C.c_int),
                ("res", C.c_void_p), ("ldres", C.c_int),
                ("mask", C.c_void_p), ("ldmask", C.c_int),
                ("y", C.c_void_p), ("ldy", C.c_int), ("n", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("taps", C.c_int), ("flags", C.c_int), ("dtype", C.c_int)]


class WgradArgs(C.Structure):
    _fields_ = [("p", C.c_void_p), ("ldp", C.c_int), ("m", C.c_int),
                ("q", C.c_void_p), ("ldq", C.c_int), ("n", C.c_int),
                ("out", C.c_void_p), ("bias_out", C.c_void_p), ("bias_mode", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("taps", C.c_int), ("flags", C.c_int), ("dtype", C.c_int), ("ksplit", C.c_int),
                ("stack_blocks", C.c_int), ("stack_width", C.c_int), ("stack_m0", C.c_int),
                ("stack_out", C.c_void_p * 8), ("stack_bias", C.c_void_p * 8)]


class ConvBwdArgs(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("ld_dy", C.c_int), ("cout", C.c_int),
                ("x", C.c_void_p), ("ld_x", C.c_int), ("cin", C.c_int),
                ("wd", C.c_void_p), ("n_pad", C.c_int), ("k_pad", C.c_int),
                ("dx", C.c_void_p), ("ld_dx", C.c_int),
                ("dw", C.c_void_p), ("db", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("use_mask", C.c_int), ("accumulate", C.c_int), ("dtype", C.c_int)]


class ConvTArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ld_x", C.c_int), ("cin", C.c_int),
                ("y", C.c_void_p), ("ld_y", C.c_int), ("cout", C.c_int),
                ("w", C.c_void_p), ("n_pad", C.c_int), ("k_pad", C.c_int),
                ("bias", C.c_void_p), ("relu", C.c_int),
                ("dx", C.c_void_p), ("ld_dx", C.c_int), ("dw", C.c_void_p), ("db", C.c_void_p), ("use_mask", C.c_int), ("accumulate", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int)]


class ConvKsArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("cin", C.c_int),
                ("wp", C.c_void_p), ("n_pad", C.c_int), ("k_pad", C.c_int),
                ("bias", C.c_void_p), ("nbias", C.c_int),
                ("y", C.c_void_p), ("ldy", C.c_int),
                ("n0", C.c_int), ("n", C.c_int),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
                ("mode", C.c_int), ("flags", C.c_int), ("dtype", C.c_int),
                ("mask", C.c_void_p), ("ldmask", C.c_int)]


class PackDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("taps", C.c_int), ("n", C.c_int), ("k", C.c_int), ("n_pad", C.c_int),
                ("k_pad", C.c_int), ("tap_flip", C.c_int), ("s_tap", C.c_long), ("s_n", C.c_long), ("s_k", C.c_long),
                ("dst_ld", C.c_long), ("dst_tap_stride", C.c_long)]


class FeatureParams(C.Structure):
    _fields_ = [("use_log1p", C.c_int), ("mean", C.c_float), ("inv_std", C.c_float),
                ("use_variance", C.c_int), ("variance_before", C.c_int), ("mode_neighbor", C.c_int),
                ("relative", C.c_int), ("compress", C.c_int), ("epsilon", C.c_float)]


class GatherEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("pixel_stride", C.c_int), ("batch_stride_pixels", C.c_int),
                ("nch", C.c_int), ("dst_ch", C.c_int)]


class AssembleEntry(C.Structure):
    _fields_ = [("src", C.c_void_p), ("cs", C.c_int), ("kind", C.c_int), ("fp", FeatureParams), ("nch", C.c_int), ("dst_ch", C.c_int),
                ("std_out", C.c_void_p), ("ld_std", C.c_int)]


class LossDesc(C.Structure):
    _fields_ = [("n_features", C.c_int),
                ("pred", C.c_void_p * MAX_FEATURES), ("target", C.c_void_p * MAX_FEATURES),
                ("dpred", C.c_void_p * MAX_FEATURES),
                ("target_ld", C.c_int * MAX_FEATURES), ("pred_ld", C.c_int * MAX_FEATURES), ("nch", C.c_int * MAX_FEATURES),
                ("weight", C.c_float * MAX_FEATURES), ("var_weight", C.c_float * MAX_FEATURES),
                ("masked_weight", C.c_float * MAX_FEATURES), ("mask_feature", C.c_int * MAX_FEATURES),
                ("n_combined", C.c_int), ("comb", (C.c_int * 3) * MAX_COMBINED),
                ("comb_weight", C.c_float * MAX_COMBINED), ("comb_var_weight", C.c_float * MAX_COMBINED),
                ("comb_masked_weight", C.c_float * MAX_COMBINED), ("comb_mask_feature", C.c_int * MAX_COMBINED),
                ("n_image_combined", C.c_int), ("image_combined", C.c_int * MAX_COMBINED),
                ("n_image_features", C.c_int), ("image_features", C.c_int * MAX_FEATURES),
                ("image_weight", C.c_float), ("image_var_weight", C.c_float), ("kind", C.c_int), ("epsilon", C.c_float),
                ("mask_sums", C.c_void_p),
                ("pred_std", C.c_void_p * MAX_FEATURES), ("pred_inv", C.c_void_p * MAX_FEATURES),
                ("inv_log1p", C.c_int * MAX_FEATURES), ("inv_mean", C.c_float * MAX_FEATURES), ("inv_std", C.c_float * MAX_FEATURES)]


class AugmentDraw(C.Structure):
    _fields_ = [("flip", C.c_int), ("rotate", C.c_int), ("permute", C.c_int), ("normal_rotation", C.c_float * 9)]


AUG_PLAIN, AUG_RGB, AUG_NORMAL, AUG_SCREEN_NORMAL = 0, 1, 2, 3


class StitchEntry(C.Structure):
    _fields_ = [("tile", C.c_int), ("crop_y0", C.c_int), ("crop_y1", C.c_int), ("crop_x0", C.c_int),
                ("crop_x1", C.c_int), ("dst_img", C.c_int), ("dst_y", C.c_int), ("dst_x", C.c_int)]


class ComposeArgs(C.Structure):
    _fields_ = [("small", C.c_void_p), ("ld_small", C.c_int), ("fine", C.c_void_p), ("ld_fine", C.c_int), ("out", C.c_void_p), ("ld_out", C.c_int),
                ("w_in", C.c_void_p), ("b_in", C.c_void_p), ("w_res", C.c_void_p * 4), ("b_res", C.c_void_p * 4),
                ("w_out", C.c_void_p), ("b_out", C.c_void_p),
                ("save_netin", C.c_void_p), ("ld_netin", C.c_int), ("save_act", C.c_void_p * 5), ("ld_act", C.c_int * 5),
                ("save_wl", C.c_void_p), ("ld_wl", C.c_int),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int)]


class ComposeBwdArgs(C.Structure):
    _fields_ = [("small", C.c_void_p), ("ld_small", C.c_int), ("fine", C.c_void_p), ("ld_fine", C.c_int), ("dout", C.c_void_p), ("ld_dout", C.c_int),
                ("act", C.c_void_p * 5), ("ld_act", C.c_int * 5), ("wl", C.c_void_p), ("ld_wl", C.c_int),
                ("w_in", C.c_void_p), ("w_res", C.c_void_p * 4), ("w_out", C.c_void_p),
                ("d_small", C.c_void_p), ("ld_dsmall", C.c_int), ("accumulate_small", C.c_int), ("d_fine", C.c_void_p), ("ld_dfine", C.c_int),
                ("dw_in", C.c_void_p), ("db_in", C.c_void_p), ("dw_res", C.c_void_p * 4), ("db_res", C.c_void_p * 4),
                ("dw_out", C.c_void_p), ("db_out", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dtype", C.c_int),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_long)]


class HeadArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("ldx", C.c_int), ("C", C.c_int), ("src", C.c_void_p), ("ldsrc", C.c_int),
                ("wa", C.c_void_p), ("ba", C.c_void_p), ("wb", C.c_void_p), ("bb", C.c_void_p),
                ("out", C.c_void_p), ("ld_out", C.c_int), ("dout", C.c_void_p), ("ld_dout", C.c_int),
                ("dx", C.c_void_p), ("ld_dx", C.c_int), ("accumulate_dx", C.c_int),
                ("dwa", C.c_void_p), ("dba", C.c_void_p), ("dwb", C.c_void_p), ("dbb", C.c_void_p),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("ksize", C.c_int), ("dtype", C.c_int)]


class RecombineDesc(C.Structure):
    _fields_ = [("n_triples", C.c_int), ("color", C.c_void_p * 4), ("direct", C.c_void_p * 4), ("indirect", C.c_void_p * 4),
                ("combined", C.c_void_p * 4), ("n_singles", C.c_int), ("single", C.c_void_p * 8), ("image", C.c_void_p)]


_lib = None


def _check_source_hash(lib):
    """The .so travels prebuilt (git-ignored) next to its sources: refuse one that was built from different sources (dd_version() carries
    the hash of csrc/* and include/dd_hip.h it was compiled from, deepdenoiser_amd/build.py).  DD_LIB experiment builds are exempt."""
    if "DD_LIB" in os.environ:
        return
    from . import build as _build
    try:
        want = _build.source_hash()
    except OSError:          # an installation without the sources next to the library: nothing to compare with
        return
    got = lib.dd_version().decode()
    if ("src=" + want) not in got:
        raise RuntimeError("libdd_hip.so is stale: built from sources %s, the sources here hash to %s -- run `python -m deepdenoiser_amd.build`"
                           % (got.split("src=")[-1], want))


def source_hash():
    """Hash of the kernel sources the loaded library was built from (== build.source_hash() of the tree, checked at load time)."""
    return load().dd_version().decode().split("src=")[-1]


def load():
    """Returns the loaded library; raises RuntimeError (never falls back) if it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime (same SONAME as /opt/rocm's): it must be loaded FIRST so that this library binds
    # to the runtime that owns torch's streams and allocations (otherwise two runtimes coexist and no device is found).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libdd_hip.so is not built (%s missing): run `python -m deepdenoiser_amd.build`. "
                           "There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise RuntimeError("libdd_hip.so lacks symbols %s: rebuild it" % missing)
    lib.dd_version.restype = C.c_char_p
    lib.dd_last_error.restype = C.c_char_p
    _check_source_hash(lib)
    for s in SYMBOLS[2:]:
        getattr(lib, s).restype = C.c_int
    vp, i, l, f = C.c_void_p, C.c_int, C.c_long, C.c_float
    lib.dd_pack_weights.argtypes = [vp, vp, i, i, i, i, i, i, l, l, l, i, vp]
    lib.dd_pack_weights_batched.argtypes = [vp, i, i, vp]
    lib.dd_conv_igemm.argtypes = [C.POINTER(ConvArgs), vp]
    lib.dd_conv_wgrad.argtypes = [C.POINTER(WgradArgs), vp]
    lib.dd_conv3x3_bwd.argtypes = [C.POINTER(ConvBwdArgs), vp]
    lib.dd_conv3x3_bwd_multi.argtypes = [C.POINTER(ConvBwdArgs), i, vp]
    lib.dd_convt2x2_fwd.argtypes = [C.POINTER(ConvTArgs), vp]
    lib.dd_convt2x2_bwd.argtypes = [C.POINTER(ConvTArgs), vp]
    lib.dd_conv3x3_ks.argtypes = [C.POINTER(ConvKsArgs), vp]
    lib.dd_space_to_depth2.argtypes = [vp, i, vp, i, i, i, i, i, i, i, vp]
    lib.dd_convt3_wgrad.argtypes = [C.POINTER(ConvT3WgradArgs), vp]
    lib.dd_conv_pw_count.argtypes = []
    lib.dd_conv_pw_count.restype = C.c_long
    lib.dd_wgrad_pw_count.argtypes = []
    lib.dd_wgrad_pw_count.restype = C.c_long
    lib.dd_colsum.argtypes = [vp, i, i, l, vp, i, vp]
    lib.dd_colsum_segments.argtypes = [vp, i, i, l, i, vp, i, vp, i, vp]
    lib.dd_maxpool_fwd.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, i, i, i, vp]
    lib.dd_maxpool_bwd.argtypes = [vp, i, vp, vp, i, vp, i, i, i, i, i, i, i, i, i, vp]
    lib.dd_avgpool.argtypes = [vp, i, vp, i, i, i, i, i, i, vp]
    lib.dd_prepare_feature.argtypes = [vp, i, vp, i, C.POINTER(FeatureParams), i, i, i, vp]
    lib.dd_gather_input.argtypes = [vp, i, i, vp, i, i, i, i, i, i, vp]
    lib.dd_kpcn_fwd.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, i, vp]
    lib.dd_kpcn_bwd.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, i, i, vp]
    lib.dd_kpcn_hidden_fwd.argtypes = [vp, i, vp, i, i, vp, i, vp, vp, i, i, i, i, i, i, vp]
    lib.dd_kpcn_hidden_bwd.argtypes = [vp, i, vp, i, i, vp, i, vp, vp, i, vp, i, i, i, i, i, i, i, vp]
    lib.dd_compose_pack.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, i, vp]
    lib.dd_compose_blend_fwd.argtypes = [vp, i, vp, i, vp, i, vp, i, i, i, i, i, vp]
    lib.dd_compose_blend_bwd.argtypes = [vp, i, vp, i, vp, i, vp, i, vp, i, i, vp, i, vp, i, i, i, i, i, i, vp]
    lib.dd_compose_unpack_bwd.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, vp]
    lib.dd_invert_std_fwd.argtypes = [vp, vp, l, i, f, f, vp]
    lib.dd_invert_std_bwd.argtypes = [vp, vp, vp, l, i, f, f, vp]
    lib.dd_loss_head.argtypes = [C.POINTER(LossDesc), i, i, i, vp, f, vp]
    lib.dd_adam_step.argtypes = [vp, vp, vp, vp, l, f, f, f, f, f, vp]
    lib.dd_stitch.argtypes = [vp, i, i, vp, i, i, i, i, vp, i, vp]
    lib.dd_recombine.argtypes = [C.POINTER(RecombineDesc), l, vp]
    lib.dd_probe_tr16.argtypes = [vp, vp, vp, vp]
    lib.dd_masked_add.argtypes = [vp, i, vp, i, vp, i, i, l, i, i, vp]
    lib.dd_convert_channels.argtypes = [vp, i, i, vp, i, i, i, i, l, vp]
    lib.dd_zero_stuff.argtypes = [vp, i, vp, i, i, i, i, i, i, vp]
    lib.dd_zero_unstuff.argtypes = [vp, i, vp, i, vp, i, i, i, i, i, i, i, vp]
    lib.dd_augment.argtypes = [vp, vp, i, i, i, i, vp, i, i, i, i, i, vp]
    lib.dd_loss_mask_sums.argtypes = [vp, i, i, i, vp, vp]
    lib.dd_crc32c.argtypes = [vp, C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.dd_extract_tiles.argtypes = [vp, i, i, i, i, vp, i, i, vp, i, vp]
    lib.dd_compose_net_fwd.argtypes = [C.POINTER(ComposeArgs), vp]
    lib.dd_compose_net_bwd.argtypes = [C.POINTER(ComposeBwdArgs), vp]
    lib.dd_compose_stream_plan.argtypes = [i, i, i, i, C.POINTER(C.c_int)]
    lib.dd_compose_bwd_scratch_bytes.argtypes = [i, i, i]
    lib.dd_compose_bwd_scratch_bytes.restype = C.c_long
    lib.dd_assemble_input.argtypes = [vp, i, i, vp, i, i, i, i, i, i, vp]
    lib.dd_assemble_input_frames.argtypes = [vp, i, i, vp, i, i, i, i, i, i, vp, i, i, vp]
    lib.dd_kpcn_head_fwd.argtypes = [C.POINTER(HeadArgs), vp]
    lib.dd_kpcn_head_bwd.argtypes = [C.POINTER(HeadArgs), vp]
    lib.dd_kpcn_head_bwd_multi.argtypes = [C.POINTER(HeadArgs), i, vp]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise RuntimeError("libdd_hip: %s (status %d)" % (load().dd_last_error().decode(), rc))
