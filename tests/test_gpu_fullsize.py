"""-m gpu: size-independent properties of the hot path at BASELINE.json's FULL sizes (cfg-2: 128x128x32ch U-Net [64,96,128]x4 +
5x5 KPCN + 3 scales, bf16 storage, tens of tiles per batch; cfg-5: a 1080x1920 frame).  The oracle takes minutes per tile at
these sizes, so nothing here compares against it; each test checks an identity that holds for the reference's arithmetic at any
size and that the small oracle-parity tests cannot see (tile walks over > 256 workgroups, XCD partitioning, atomics under load):

  * tiles are independent samples (no cross-sample op in the reference graph: BN is hard-disabled, Architecture.py:506), so two
    copies of a tile in one batch give bit-identical predictions, and a second run reproduces the first bit for bit;
  * the mean loss is linear in per-tile terms (Training.py:126-129): the gradient of a batch is the mean of its shards' gradients
    -- the identity the multi-GPU all-reduce relies on;
  * softmax kernel prediction (KernelPrediction.py:11-63) reproduces a constant image;
  * stitching the halo tiles of a frame gives back the frame (Prediction.py:380-441), bit-exact.
"""
import pytest
import torch

from deepdenoiser_amd import configs
from deepdenoiser_amd.naming import Naming
from gpu_util import rel_l2

pytestmark = pytest.mark.gpu

TILE = 128


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _inputs(arch, B, seed):
    g = torch.Generator().manual_seed(seed)
    feats, labels = {}, {}
    for f in arch.feature_predictions + arch.auxiliary_features:
        v = torch.randn(B, TILE, TILE, f.number_of_channels, generator=g)
        if f.name != "Normal":
            v = v.abs() * torch.exp(0.5 * torch.randn(B, TILE, TILE, 1, generator=g))
        feats[Naming.source_feature_name(f.name, index=0)] = v.cuda()
    for f in arch.feature_predictions:
        labels[Naming.target_feature_name(f.name)] = torch.randn(B, TILE, TILE, f.number_of_channels, generator=g).abs().cuda()
    return feats, labels


def _program(B, training=False):
    from deepdenoiser_amd.architecture import Architecture
    arch = Architecture(configs.cfg2_unet_kpcn(), device="cuda", dtype="bf16", seed=2)
    prog = arch.program(B, TILE, TILE, training_json=configs.bench_training() if training else None)
    return arch, prog


def _predictions(prog):
    return [{k: v.clone() for k, v in d.items()} for d in prog.prediction_dictionaries()]


def test_cfg2_full_size_tiles_are_independent_and_runs_are_reproducible():
    _need_gpu()
    half = 24                                           # 48 tiles: 3072 conv tiles per 128x128 launch, 12 per workgroup
    arch, prog = _program(2 * half)
    feats, _ = _inputs(arch, half, seed=11)
    doubled = {k: torch.cat([v, v], 0) for k, v in feats.items()}
    prog.set_inputs(doubled)
    prog.forward()
    torch.cuda.synchronize()
    first = _predictions(prog)
    prog.forward()
    torch.cuda.synchronize()
    second = _predictions(prog)
    k0 = next(iter(first[0]))
    assert len(first) == 3 and first[0][k0].shape[:3] == (2 * half, TILE, TILE) and first[2][k0].shape[1:3] == (TILE // 4, TILE // 4)
    for da, db in zip(first, second):
        for k in da:
            assert torch.isfinite(da[k]).all(), k
            assert torch.equal(da[k], db[k]), "forward is not reproducible: " + k
            assert torch.equal(da[k][:half], da[k][half:]), "a tile's prediction depends on its position in the batch: " + k
    # and the tiles are not all the same tile
    p = first[0][k0]
    assert not torch.equal(p[0], p[1])


def test_cfg2_full_size_batch_gradient_is_the_mean_of_shard_gradients():
    """What rank r computes on its shard, averaged over ranks, is the single-process gradient."""
    _need_gpu()
    shard = 16
    arch_full, full = _program(2 * shard, training=True)
    feats, labels = _inputs(arch_full, 2 * shard, seed=12)

    def grads_of(prog, arch, f, lab):
        prog.set_inputs(f, lab)
        prog.zero_grads()
        prog.forward()
        prog.backward()
        torch.cuda.synchronize()
        return arch.params.grads.double().clone(), float(prog.loss_buf)

    g_full, l_full = grads_of(full, arch_full, feats, labels)
    arch_s, part = _program(shard, training=True)
    assert torch.equal(arch_s.params.values, arch_full.params.values)          # identical replicas (same seed)
    g_parts, l_parts = [], []
    for r in range(2):
        sl = slice(r * shard, (r + 1) * shard)
        g, l = grads_of(part, arch_s, {k: v[sl] for k, v in feats.items()}, {k: v[sl] for k, v in labels.items()})
        g_parts.append(g)
        l_parts.append(l)
    assert float(g_full.abs().max()) > 0 and torch.isfinite(g_full).all()
    assert abs(l_full - 0.5 * (l_parts[0] + l_parts[1])) <= 1e-5 * abs(l_full)
    # per-tile arithmetic is identical in both layouts; only the fp32 summation order of the weight-gradient atomics differs
    assert rel_l2(0.5 * (g_parts[0] + g_parts[1]), g_full) <= 1e-4


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_full_size_kernel_prediction_reproduces_a_constant_image(lib, dtype):
    _need_gpu()
    from deepdenoiser_amd import _lib as L
    B, ks, ld = 128, 5, 32
    g = torch.Generator().manual_seed(13)
    const = torch.tensor([0.25, 3.0, -1.5, 0.0])
    src = const.expand(B, TILE, TILE, 4).contiguous().cuda()
    tdt = torch.float32 if dtype == "f32" else torch.bfloat16
    logits = (4 * torch.randn(B, TILE, TILE, ld, generator=g)).to(tdt).cuda()
    out = torch.zeros(B, TILE, TILE, 3).cuda()
    L.check(lib.dd_kpcn_fwd(src.data_ptr(), 4, logits.data_ptr(), ld, out.data_ptr(), 3, B, TILE, TILE, ks,
                            L.DD_F32 if dtype == "f32" else L.DD_BF16, None))
    torch.cuda.synchronize()
    err = (out - const[:3].cuda()).abs().max()
    assert float(err) <= 2e-6 * 3.0, float(err)        # softmax weights sum to one, up to fp32 rounding of 25 terms


def test_full_hd_frame_survives_tiling_and_stitching_bit_exact(lib):
    """cfg-5: 1080x1920 -> 11x19 halo tiles of 128 (overlap 14) -> stitch: every output pixel comes from exactly one tile crop."""
    _need_gpu()
    from deepdenoiser_amd import _lib as L
    from deepdenoiser_amd.tiling import tile_plan
    H, W, T = 1080, 1920, 128
    plan = tile_plan(H, W, T, 14)
    assert (plan.rows.count, plan.cols.count) == (11, 19)
    g = torch.Generator().manual_seed(14)
    frame = torch.randn(H, W, 3, generator=g).cuda()
    win = plan.windows()                                                                # row-major tile origins (Prediction.py:380-382)
    assert len(win) == 209 and win[-1] == (952, 1792)
    ys, xs = torch.tensor([y for y, _ in win]), torch.tensor([x for _, x in win])
    ar = torch.arange(T)
    yy = (ys[:, None, None] + ar[None, :, None]).cuda()
    xx = (xs[:, None, None] + ar[None, None, :]).cuda()
    tiles = frame[yy, xx].contiguous()                                                  # [209,128,128,3]
    # the C-ABI extraction (Prediction.py:283-310) gives the same tiles, also for 1- and 4-channel frames with padded rows
    oyx = torch.tensor(win, dtype=torch.int32).cuda()
    got = torch.full_like(tiles, float("nan"))
    L.check(lib.dd_extract_tiles(frame.data_ptr(), H, W, 3, 3, got.data_ptr(), T, 3, oyx.data_ptr(), len(win), None))
    assert torch.equal(got, tiles)
    wide = torch.randn(H, W, 4, generator=g).cuda()
    for C in (1, 4):
        out_c = torch.zeros(len(win), T, T, 8).cuda()
        L.check(lib.dd_extract_tiles(wide.data_ptr(), H, W, 4, C, out_c.data_ptr(), T, 8, oyx.data_ptr(), len(win), None))
        assert torch.equal(out_c[..., :C], wide[yy, xx][..., :C]) and float(out_c[..., C:].abs().max()) == 0.0
    table = (L.StitchEntry * len(win))()
    for i in range(len(win)):
        hi, wi = divmod(i, plan.cols.count)
        (a, b), (c, d) = plan.rows.crops[hi], plan.cols.crops[wi]
        table[i] = L.StitchEntry(i, a, b, c, d, 0, plan.rows.offsets[hi], plan.cols.offsets[wi])
    td = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).cuda()
    out = torch.full((1, H, W, 3), float("nan")).cuda()
    L.check(lib.dd_stitch(tiles.data_ptr(), T, 3, out.data_ptr(), H, W, 3, 3, td.data_ptr(), len(win), None))
    torch.cuda.synchronize()
    assert torch.equal(out[0], frame)
