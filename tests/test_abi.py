"""The C-ABI library loads and exports every symbol include/dd_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from deepdenoiser_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SYMBOLS) == names


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.dd_version()
    assert isinstance(lib.dd_last_error(), bytes)


def test_library_carries_the_hash_of_the_sources_next_to_it(lib):
    """dd_version() ends in src=<sha256[:16] of csrc/* + include/dd_hip.h>: the prebuilt, git-ignored .so is provably built from this tree
    (and _lib.load() refuses a stale one)."""
    from deepdenoiser_amd import build
    assert lib.dd_version().decode().endswith("src=" + build.source_hash())
    assert _lib.source_hash() == build.source_hash()


def test_invalid_arguments_return_status_not_exception(lib):
    a = _lib.ConvArgs()          # all-null
    assert lib.dd_conv_igemm(ctypes.byref(a), None) == -1
    assert b"null" in lib.dd_last_error()
    w = _lib.WgradArgs()
    assert lib.dd_conv_wgrad(ctypes.byref(w), None) == -1
    assert lib.dd_adam_step(None, None, None, None, 0, 0.0, 0.9, 0.999, 1e-8, 1.0, None) == -1
    # data augmentation: the reference refuses to flip world-space normals (DataAugmentation.py:22-23) and needs square tiles to rotate
    src, dst, draws = ctypes.c_void_p(16), ctypes.c_void_p(32), ctypes.c_void_p(64)      # never dereferenced: validation fails first
    assert lib.dd_augment(src, dst, 3, 1, 8, 8, draws, _lib.AUG_NORMAL, 1, 0, 0, 0, None) == -1
    assert b"normals" in lib.dd_last_error()
    assert lib.dd_augment(src, dst, 3, 1, 8, 12, draws, _lib.AUG_RGB, 0, 1, 0, 0, None) == -1
    assert b"square" in lib.dd_last_error()
    assert lib.dd_augment(src, dst, 2, 1, 8, 8, draws, _lib.AUG_PLAIN, 0, 0, 0, 0, None) == -1
    assert lib.dd_loss_mask_sums(None, 1, 8, 8, None, None) == -1
    assert lib.dd_maxpool_fwd(None, 8, None, 8, None, 8, 1, 8, 8, 3, 2, 1, _lib.DD_BF16, None) == -1


def test_struct_sizes_match_header(lib):
    # compile a tiny C program against the header and compare sizeof() of every struct with the ctypes mirror
    import subprocess, tempfile
    src = r'''
#include <stdio.h>
#include "dd_hip.h"
int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(dd_convt3_wgrad_args), sizeof(dd_conv_ks_args), sizeof(dd_convt_args), sizeof(dd_conv_bwd_args), sizeof(dd_assemble_entry), sizeof(dd_head_args), sizeof(dd_compose_bwd_args), sizeof(dd_compose_args), sizeof(dd_conv_args), sizeof(dd_wgrad_args), sizeof(dd_feature_params),
  sizeof(dd_gather_entry), sizeof(dd_loss_desc), sizeof(dd_stitch_entry), sizeof(dd_recombine_desc), sizeof(dd_pack_desc), sizeof(dd_augment_draw)); return 0; }
'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    mirror = [ctypes.sizeof(t) for t in (_lib.ConvT3WgradArgs, _lib.ConvKsArgs, _lib.ConvTArgs, _lib.ConvBwdArgs, _lib.AssembleEntry, _lib.HeadArgs, _lib.ComposeBwdArgs, _lib.ComposeArgs, _lib.ConvArgs, _lib.WgradArgs, _lib.FeatureParams, _lib.GatherEntry, _lib.LossDesc, _lib.StitchEntry, _lib.RecombineDesc, _lib.PackDesc, _lib.AugmentDraw)]
    assert sizes == mirror


def test_build_refuses_scratch_in_kernels_with_inplace_asm_accumulators():
    """deepdenoiser_amd/build.py: the kernels whose accumulators are updated by in-place inline-asm MFMAs are only correct without spills; the
    build parses hipcc's resource-usage remarks and refuses a non-zero ScratchSize."""
    import pytest
    from deepdenoiser_amd import build
    ok = ("x.hip:1:1: remark: Function Name: _ZN12_GLOBAL__N_115conv_bwd_kernelItLb1ELb0EEEvNS_4BwdPE [-Rpass-analysis=kernel-resource-usage]\n"
          "x.hip:1:1: remark:     VGPRs: 210 [-Rpass-analysis=kernel-resource-usage]\n"
          "x.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n")
    build._check_no_scratch("dd_conv_bwd.hip", ("conv_bwd_kernel",), ok)
    with pytest.raises(RuntimeError, match="scratch"):
        build._check_no_scratch("dd_conv_bwd.hip", ("conv_bwd_kernel",), ok.replace("lane]: 0", "lane]: 48"))
    with pytest.raises(RuntimeError, match="no resource-usage remark"):
        build._check_no_scratch("dd_conv_bwd.hip", ("conv_bwd_kernel",), "nothing here")
    assert set(build.NO_SCRATCH) <= set(build.SOURCES)


def test_compose_stream_plan_covers_every_pixel_once(lib):
    """dd_compose_stream_plan (host only): the strip / band geometry of the row-streaming compose kernels.  Strips and bands must tile the image,
    frames must hold a strip plus its 4-column halo and fit the 128-pixel step, bands must be a whole number of steps."""
    import ctypes
    for N, H, W in [(128, 128, 128), (128, 64, 64), (53, 128, 128), (8, 256, 256), (3, 24, 40), (1, 16, 16), (2, 64, 64), (1, 20, 12), (209, 128, 128),
                    (1, 2, 2), (4, 258, 130), (1, 1080, 1920)]:
        o = (ctypes.c_int * 8)()
        assert lib.dd_compose_stream_plan(N, H, W, 256, o) == 0
        FW, R, TPR, strips, SO, BH, nb, VB = list(o)
        assert FW % 32 == 0 and 32 <= FW <= 128 and TPR == FW // 32 and R == {32: 4, 64: 2, 96: 1, 128: 1}[FW]
        assert strips * SO >= W and SO % 2 == 0 and (strips == 1 or FW >= SO + 8) and (strips > 1 or FW >= W)
        assert BH % 4 == 0 and nb * BH >= H and (nb - 1) * BH < H and VB == BH + 8 and VB % R == 0
    assert lib.dd_compose_stream_plan(0, 8, 8, 256, (ctypes.c_int * 8)()) == -1
    assert lib.dd_compose_bwd_scratch_bytes(2, 16, 32) == 2 * 16 * 32 * 272
