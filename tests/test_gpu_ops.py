"""-m gpu: hand-written HIP kernels (through the C-ABI + engine) vs the CPU oracle, op by op.
Inputs, weights and upstream gradients are REPRESENTABLE in the storage type, so every product is exact in fp32 and the gates are set by what
an output is (tests/gpu_util.py): storage-type outputs (y, dx) within one-to-two roundings (ROUND: bf16 4e-3, fp16 5e-4), fp32 outputs
(dW, db, kernel-prediction output) within fp32 summation order (ACC32: 5e-6 for bf16 and fp16 alike), the f32 path 5e-6 throughout."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from gpu_util import ACC32, ROUND, TOL, check, fill, read, rel_l2, representable, set_param

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd import engine
    return engine


def _gen(seed):
    return torch.Generator().manual_seed(seed)


def test_tr16_probe_pins_transpose_read_layout(lib):
    """ds_read_b64_tr_b16 semantics the bf16 wgrad kernel relies on: within a 16-lane group lane t receives, as element j,
    the (t%4)-th 16-bit word of the 8 bytes addressed by lane 4*j + t//4."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    img = torch.arange(4096, dtype=torch.int16).cuda()
    rng = np.random.default_rng(0)
    addr = (rng.integers(0, 1000, size=64) * 8).astype(np.int32)       # arbitrary 8-byte aligned per-lane addresses
    a = torch.tensor(addr).cuda()
    out = torch.zeros(64 * 4, dtype=torch.int16).cuda()
    assert lib.dd_probe_tr16(img.data_ptr(), a.data_ptr(), out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(64, 4)
    want = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        grp, t = lane // 16, lane % 16
        for j in range(4):
            src_lane = grp * 16 + 4 * j + t // 4
            want[lane, j] = addr[src_lane] // 2 + (t % 4)
    assert (got == want).all(), (got[:20], want[:20])


CONV_CASES = [
    # k, cin, cout, H, W, relu, in_relu, residual, x_is_relu_output
    (3, 32, 64, 32, 32, True, False, False, False),
    (3, 64, 96, 16, 16, True, False, False, True),
    (3, 192, 96, 16, 16, True, False, False, True),
    (3, 128, 128, 8, 8, True, False, False, True),
    (3, 96, 96, 16, 16, True, False, False, True),      # odd number of 64-byte K chunks: half-slab weights, 48-channel blocks
    (3, 96, 192, 16, 16, False, False, False, True),
    (3, 32, 96, 20, 12, True, False, True, False),
    (3, 24, 24, 20, 28, False, True, True, False),      # compose-net residual block conv, ragged tile
    (1, 64, 25, 32, 32, True, False, False, True),      # AdjustNumberOfChannels
    (1, 25, 25, 32, 32, False, False, False, True),
    (1, 6, 24, 16, 16, True, False, False, False),
    (1, 24, 1, 16, 16, True, False, False, False),
    (3, 3, 16, 24, 24, True, False, False, False),       # cfg-1 first layer
    (1, 320, 320, 8, 8, False, True, False, False),      # Tiramisu transition-down
    (1, 400, 400, 8, 8, False, True, False, False),      # 7 x 7 channel-slice pairs: the weight gradient leaves the LDS-DMA kernel's split table
    # round 4: the ReLU-backward data gradient on the all-wave register-weight kernels (mask tile by LDS-DMA) -- ragged 16 x 8 tiles, output blocks
    # that are not full (80 of 96 channels), 72 / 88 channels of depth, the 97..128-channel form with a ragged last tile row
    (3, 96, 96, 20, 28, True, False, False, True),
    (3, 80, 72, 9, 33, True, False, False, True),
    (3, 128, 112, 13, 19, True, False, False, True),
    (3, 88, 128, 37, 17, True, False, False, True),
    (3, 32, 64, 256, 264, True, False, False, False),    # images of >= 256 rows and a half-filled 64-channel slice: the lanes of the missing channels
                                                         # must stay out of range whatever the image height (a row sentinel of 255 did not: round 3)
]


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("k,cin,cout,H,W,relu,in_relu,residual,x_relu", CONV_CASES)
def test_conv_fwd_bwd(eng, dtype, k, cin, cout, H, W, relu, in_relu, residual, x_relu):
    _conv_case(eng, dtype, k, cin, cout, H, W, relu, in_relu, residual, x_relu, B=2)


# the K-streamed kernel (csrc/dd_conv_ks.hip): Tiramisu's dense-block convs -- pre-activation, reduction over up to 1 088 channels, 16 ... 128 new
# channels written into a channel range of the concat buffer; channel blocks 64 + 32, ragged tiles, one to 17 K-slices, partial last slice
KS_CASES = [
    # cin, cout, H, W, B
    (576, 64, 32, 32, 2),
    (320, 96, 20, 28, 2),
    (1088, 128, 16, 16, 1),
    (144, 16, 40, 24, 1),
    (304, 24, 17, 33, 2),
    (200, 32, 16, 16, 3),
    (160, 48, 16, 16, 1),
    (16, 16, 35, 18, 2),
    (72, 64, 16, 48, 1),
    (168, 16, 16, 16, 1),       # one 16-channel tile, three K-slices: fewer fragment steps per unit than DMA pieces
    (296, 16, 8, 8, 1),
    (136, 24, 24, 8, 2),
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,B", KS_CASES)
def test_conv_k_streamed_dense_block_layers(eng, dtype, cin, cout, H, W, B, monkeypatch):
    monkeypatch.setenv("DD_CONV_KS_THIN", "1")      # (thin layers over a short reduction default to dd_conv_igemm since round 5: next test)
    _conv_case(eng, dtype, 3, cin, cout, H, W, False, True, False, False, B=B, expect_fwd_tag="ks_fwd")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,B", [(144, 16, 40, 24, 1), (16, 16, 35, 18, 2), (136, 24, 24, 8, 2), (96, 32, 16, 16, 2)])
def test_thin_pre_activation_layers_run_on_the_lds_weight_kernel(eng, dtype, cin, cout, H, W, B):
    """<= 32 new channels from <= 144 (the 256 x 256 level of the light Tiramisu, Tiramisu.py:26-41): the whole weight image stays in LDS on
    dd_conv_igemm (round 5: 47 - 64 us against 73 - 95 us K-streamed at B = 8, 256 x 256); same gates as every conv."""
    _conv_case(eng, dtype, 3, cin, cout, H, W, False, True, False, False, B=B)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("form", ["gather", "scatter"])
@pytest.mark.parametrize("c0,f,n,H,W,B", [(16, 16, 3, 20, 28, 2), (40, 24, 4, 16, 16, 1), (64, 32, 2, 17, 33, 2), (96, 64, 3, 16, 16, 1),
                                          (160, 32, 2, 48, 45, 1)])      # a prefix wider than 128 channels on >= 2048 pixels: its gather runs as GEMM tiles
def test_dense_block_backward_gather_and_scatter_forms(eng, dtype, form, c0, f, n, H, W, B, monkeypatch):
    """engine.Graph.dense_block (Tiramisu.py:26-41): n pre-activation 3x3 convs appending f channels each to a concat buffer, differentiated in
    gather form (one K-streamed launch per channel range: all later convs' contributions at once, rounded once; csrc/dd_conv_ks.hip DD_ACCUM with
    stacked weight images) and in scatter form (every conv's data gradient accumulated into its whole prefix).  Against the oracle chain with
    the storage roundings: forward ranges, the gradient of every channel of the buffer (incl. the contributions from outside the block that
    are already stored), every kernel and bias gradient."""
    monkeypatch.setenv("DD_DENSE_GATHER", "1" if form == "gather" else "0")
    from oracle.model import VarStore
    gen = _gen(c0 + f + n)
    g = eng.Graph("cuda", dtype)
    total = c0 + n * f
    buf = g.tensor(B, H, W, total, relu=False)
    buf.gstate["zero_init"] = True
    layers = [g.layer("d/conv2d_%d" % j, 3, c0 + j * f, f) for j in range(n)]
    assert g.dense_block(buf, c0, f, layers) == total
    g.build_backward()
    g.finalize()
    tags = [getattr(op, "__name__", "") for op in g.bwd_ops]
    assert (tags.count("dense_gather") == n) == (form == "gather" and dtype != "f32"), tags
    xv = representable(torch.randn(B, H, W, c0, generator=gen, dtype=torch.float64), dtype)
    ws, bs = [], []
    for lay in layers:
        w = representable(torch.randn(lay.kernel.shape, generator=gen, dtype=torch.float64) / (3 * lay.cin ** 0.5), dtype)
        b = torch.randn(lay.cout, generator=gen, dtype=torch.float64).float().double()
        set_param(g.params, lay.kernel, w); set_param(g.params, lay.bias, b)
        ws.append(w.clone().requires_grad_(True)); bs.append(b.clone().requires_grad_(True))
    buf.buf.zero_()
    buf.buf[..., :c0] = xv.to(buf.buf.dtype).cuda()
    g.run(g.pack_ops); g.run(g.fwd_ops)
    torch.cuda.synchronize()
    emu = VarStore(storage=dtype)
    xo = xv.clone().requires_grad_(True)
    parts = [emu.qgrad(xo)]
    for j in range(n):
        pre = T.conv2d_same(torch.relu(torch.cat(parts, dim=3)), ws[j], bs[j], False)
        parts.append(emu.q(emu.qgrad(pre)))
    full = torch.cat(parts, dim=3)
    got = buf.buf[..., :total].double().cpu()
    check("dense block forward", got, full.detach(), ROUND[dtype] * (1 if dtype == "f32" else 2))
    # gradient arriving from outside the block on every channel (the transition conv, the transposed conv, the heads): already stored
    G = representable(torch.randn(B, H, W, total, generator=gen, dtype=torch.float64), dtype)
    for t in g.zero_init_buffers:
        t.zero_()
    buf.grad().buf[..., :total] = G.to(buf.buf.dtype).cuda()
    g.params.grads.zero_()
    grads = torch.autograd.grad((full * G).sum(), [xo] + ws + bs)
    pw_before = eng.L.load().dd_conv_pw_count()
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    if form == "gather" and dtype != "f32":      # conv_pw_kernel with nine taps takes the prefix gather of the wide case, conv_ks_kernel the rest
        assert eng.L.load().dd_conv_pw_count() - pw_before == (1 if (c0 > 128 and B * H * W >= 2048) else 0)
    # Measured (profiles/r03_parity_errors.txt).  Gather form: every range is rounded ONCE, exactly where the oracle chain rounds it -- the prefix
    # gradient agrees to 2e-7 ... 6e-5 and dW to 6e-8 ... 6e-5 (the upper end: one stored value on the other side of a rounding boundary).
    # Scatter form: the running sum is rounded once per contributing conv -- prefix 3.5e-3 (bf16) / 4.4e-4 (fp16), dW up to 3.1e-3 / 3.7e-4.
    half = dtype != "f32"
    g_prefix = {True: {"gather": 3e-4, "scatter": 3 * ROUND[dtype]}, False: {"gather": ROUND[dtype], "scatter": ROUND[dtype]}}[half][form]
    g_dw = {True: {"gather": 3e-4, "scatter": 1.5 * ROUND[dtype]}, False: {"gather": ACC32[dtype], "scatter": ACC32[dtype]}}[half][form]
    check("d prefix (%s)" % form, buf.grad().buf[..., :c0].double().cpu(), grads[0], g_prefix)
    for j, lay in enumerate(layers):
        check("dW %s (%s)" % (lay.name, form), g.params.grad(lay.kernel).double().cpu(), grads[1 + j], g_dw)
        check("db %s (%s)" % (lay.name, form), g.params.grad(lay.bias).double().cpu(), grads[1 + n + j], g_dw)


# the fused data + weight gradient launch (csrc/dd_conv_bwd.hip: 3x3, <= 64 output channels, bf16 / f16 storage; f32 runs the two-launch path)
FUSED_BWD_CASES = [
    (64, 64, 32, 32, True, 2),        # the U-Net body layer: every wave of both roles busy
    (64, 64, 20, 28, True, 3),        # ragged tiles
    (128, 64, 16, 48, True, 2),       # two 64-channel input blocks (decoder conv over the skip concat)
    (96, 48, 16, 16, True, 2),        # half input block, 3 output-channel tiles
    (32, 64, 16, 16, False, 2),       # no ReLU mask on the input
    (8, 16, 40, 24, True, 1),         # narrow: one tile of each kind, 32-channel K
    (72, 40, 33, 17, True, 2),        # nothing a multiple of 16
    # 65 - 96 output channels (round 6, csrc/dd_conv_bwd96.hip: a 32-channel third of the input per workgroup against all output channels)
    (96, 96, 32, 32, True, 2),        # the U-Net's 64 x 64 level: three input blocks, every wave of both roles busy
    (192, 96, 20, 28, True, 2),       # decoder conv over the skip concat: six input blocks, ragged tiles
    (64, 96, 16, 48, True, 3),        # first conv of the level (two input blocks)
    (96, 80, 33, 17, False, 2),       # the last output-channel tile of one parity is empty; no mask
    (40, 72, 9, 35, True, 1),         # nothing a multiple of 16: half-empty input block, ragged everything
    (24, 96, 48, 16, True, 2),        # one input block with an idle channel tile... (24 = 16 + 8)
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,x_relu,B", FUSED_BWD_CASES)
def test_conv_fused_backward(eng, dtype, cin, cout, H, W, x_relu, B):
    _conv_case(eng, dtype, 3, cin, cout, H, W, True, False, False, x_relu, B=B, expect_fused_bwd=True)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W", [(32, 64, 32, 32), (24, 40, 20, 28), (72, 16, 16, 16)])
def test_conv_first_layer_weight_gradients_only(eng, dtype, cin, cout, H, W):
    """The network's first layer: its input needs no gradient -- the fused backward launch with dx = NULL computes dW / db only."""
    _conv_case(eng, dtype, 3, cin, cout, H, W, True, False, False, False, B=2, expect_fused_bwd=True, x_requires_grad=False)


@pytest.mark.parametrize("k,cin,cout,H,W,B", [(1, 16, 27, 16, 32, 8), (1, 27, 27, 16, 32, 8), (3, 32, 16, 16, 32, 8), (1, 24, 1, 16, 32, 24),
                                               (3, 24, 24, 16, 32, 24), (1, 16, 27, 8, 16, 8), (3, 16, 16, 32, 16, 5), (3, 16, 16, 48, 16, 3)])
def test_conv_non_square_batches(eng, k, cin, cout, H, W, B):
    _conv_case(eng, "f32", k, cin, cout, H, W, True, False, False, True, B=B)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("cin,cout,split_at,H,W", [(192, 96, 96, 16, 16), (160, 64, 64, 20, 12), (192, 80, 96, 21, 27), (168, 96, 80, 9, 35)])      # (round 4: the residual half on the 12-wave kernel, ragged tiles)
def test_conv_over_skip_concat_runs_as_two_resident_launches(eng, dtype, cin, cout, split_at, H, W):
    """conv(concat[a | b]) = conv_a(a) + conv_b(b): the forward of a > 128-channel 3x3 layer over a U-Net skip concat (engine.Graph.conv,
    split_at); forward, data- and weight-gradients against the oracle as for every other layer (f32: the split is bf16-only, one launch)."""
    _conv_case(eng, dtype, 3, cin, cout, H, W, True, False, False, True, B=2, split_at=split_at)


# the wide 1x1 GEMM kernel (csrc/dd_conv_pw.hip): Tiramisu's transition convs (C -> C over the whole concat, pre-activation) and the data gradients
# of the head's 1x1 convs (25 -> C with the consumer's ReLU mask); one to three channel blocks, ragged last pixel tile, one to eleven K-slices
PW_CASES = [
    # cin, cout, H, W, B, relu, in_relu, x_relu, pw launches expected (forward, data gradient, weight gradient)
    (80, 80, 128, 128, 2, False, True, False, 1, 1, 1),
    (176, 176, 64, 64, 8, True, False, True, 1, 1, 0),      # mid-sized weight gradient: stays on the 64 x 64-slice kernel (fewer atomics)
    (320, 320, 100, 100, 4, False, True, False, 1, 1, 1),
    (704, 704, 96, 128, 3, False, True, False, 1, 1, 1),
    (160, 25, 256, 128, 1, False, False, True, 0, 1, 1),   # forward on the igemm kernel (32 padded output channels), data gradient 25 -> 160 here
    (296, 24, 128, 130, 2, False, True, False, 0, 1, 1),
    (640, 160, 64, 64, 8, True, False, False, 1, 1, 1),
    (100, 84, 181, 200, 1, True, True, False, 1, 1, 1),    # channel counts that end inside a 16-byte group, ragged last pixel tile
]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W,B,relu,in_relu,x_relu,nf,nb,nw", PW_CASES)
def test_conv_1x1_wide_gemm(lib, eng, dtype, cin, cout, H, W, B, relu, in_relu, x_relu, nf, nb, nw):
    before, wbefore = lib.dd_conv_pw_count(), lib.dd_wgrad_pw_count()
    _conv_case(eng, dtype, 1, cin, cout, H, W, relu, in_relu, False, x_relu, B=B)
    assert lib.dd_conv_pw_count() - before == nf + nb, "the wide 1x1 kernel did not take the launches it was expected to"
    assert lib.dd_wgrad_pw_count() - wbefore == nw, "the 1x1 weight-gradient GEMM did not take the launch it was expected to"


def _conv_case(eng, dtype, k, cin, cout, H, W, relu, in_relu, residual, x_relu, B, split_at=None, expect_fused_bwd=False, x_requires_grad=True,
               expect_fwd_tag=None):
    gen = _gen(k * 1000 + cin + cout)
    g = eng.Graph("cuda", dtype)
    x = g.tensor(B, H, W, cin, relu=x_relu, requires_grad=x_requires_grad)
    xv = torch.randn(B, H, W, cin, generator=gen, dtype=torch.float64)
    if x_relu:
        xv = torch.relu(xv)
    xv = representable(xv, dtype)
    res = rv = None
    if residual:
        res = g.tensor(B, H, W, cout, requires_grad=True)
        rv = representable(torch.randn(B, H, W, cout, generator=gen, dtype=torch.float64), dtype)
    lay = g.layer("t/conv2d", k, cin, cout)
    y = g.conv(x, lay, relu=relu, in_relu=in_relu, res=res, split_at=split_at)
    if split_at is not None and dtype in ("bf16", "f16"):
        assert len(g.fwd_ops) == 2, "the concat split did not engage"
    y.mark_grad_written()
    g.build_backward()
    g.finalize()
    if expect_fused_bwd:
        assert [getattr(op, "tag", "") for op in g.bwd_ops] == ["conv_bwd"], "the fused backward did not engage"
    if expect_fwd_tag is not None:
        assert [op.__name__ for op in g.fwd_ops] == [expect_fwd_tag], "the forward did not take the expected kernel"
    wv = representable(torch.randn(k, k, cin, cout, generator=gen, dtype=torch.float64) / (k * cin ** 0.5), dtype)
    bv = torch.randn(cout, generator=gen, dtype=torch.float64).float().double()
    set_param(g.params, lay.kernel, wv)
    set_param(g.params, lay.bias, bv)
    fill(x, xv)
    if residual:
        fill(res, rv)
    g.run(g.pack_ops)
    g.run(g.fwd_ops)
    torch.cuda.synchronize()

    xo = xv.clone().requires_grad_(True)
    wo, bo = wv.clone().requires_grad_(True), bv.clone().requires_grad_(True)
    ro = rv.clone().requires_grad_(True) if residual else None
    pre = T.conv2d_same(torch.relu(xo) if in_relu else xo, wo, bo, False)
    if residual:
        pre = pre + ro
    yo = torch.relu(pre) if relu else pre
    check("y", read(y), yo.detach(), ROUND[dtype])
    # pad channels must be exactly zero
    assert float(y.buf[..., y.C:y.Cp].abs().max() if y.Cp > y.C else 0) == 0.0

    G = torch.randn(B, H, W, cout, generator=gen, dtype=torch.float64)
    gpre = representable(G * (pre.detach() > 0) if relu else G, dtype)     # engine convention: stored grads are pre-activation
    fill(y.grad(), gpre)
    grads = torch.autograd.grad((pre * gpre).sum(), [xo, wo, bo] + ([ro] if residual else []))
    g.params.grads.zero_()
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    gx_want = grads[0] * (xv > 0) if (x_relu or in_relu) else grads[0]
    if x_requires_grad:
        check("dx", read(x.grad()), gx_want, ROUND[dtype])
    check("dW", g.params.grad(lay.kernel).double().cpu(), grads[1], ACC32[dtype])
    check("db", g.params.grad(lay.bias).double().cpu(), grads[2], ACC32[dtype])
    if residual:
        check("dres", read(res.grad()), grads[3], ROUND[dtype])


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("f,H,W", [(32, 16, 16), (96, 20, 12), (80, 9, 33)])
def test_conv_grad_accumulation_and_concat_views(eng, dtype, f, H, W):
    """Two consumers of one tensor (the U-Net skip pattern): first writer overwrites, second accumulates; conv writes into a channel range.
    f = 96 / 80 (bf16 / f16): the data gradients take the register-weight kernel (mask, then mask + accumulate; csrc/dd_conv_rw.hip) and the
    last layer the fused backward over three 64-channel input blocks (csrc/dd_conv_bwd.hip)."""
    B = 2
    gen = _gen(7)
    g = eng.Graph("cuda", dtype)
    x = g.tensor(B, H, W, 32, requires_grad=True, relu=True)
    cat = g.tensor(B, H, W, 2 * f, relu=True)
    l1, l2, l3 = g.layer("a/conv2d", 3, 32, f), g.layer("a/conv2d_1", 3, 32, f), g.layer("a/conv2d_2", 3, 2 * f, 16)
    g.conv(x, l1, relu=True, out=cat.view(0, f))
    g.conv(x, l2, relu=True, out=cat.view(f, f))
    z = g.conv(cat, l3, relu=False)
    z.mark_grad_written()
    g.build_backward()
    g.finalize()
    xv = representable(torch.relu(torch.randn(B, H, W, 32, generator=gen, dtype=torch.float64)), dtype)
    ws = []
    for lay in (l1, l2, l3):
        w = representable(torch.randn(lay.kernel.shape, generator=gen, dtype=torch.float64) / (3 * lay.cin ** 0.5), dtype)
        set_param(g.params, lay.kernel, w)
        ws.append(w.clone().requires_grad_(True))
    fill(x, xv)
    g.run(g.pack_ops); g.run(g.fwd_ops)
    # the oracle chain with the storage roundings of the half-precision path (oracle.model.VarStore.q / qgrad): the stored concat is
    # rounded, its gradient is masked and then rounded, x's gradient is the rounded sum of two rounded contributions
    from oracle.model import VarStore
    emu = VarStore(storage=dtype)
    xo = xv.clone().requires_grad_(True)
    xin = emu.qgrad(xo)

    def stored(pre):
        return emu.q(torch.relu(emu.qgrad(pre)))
    co = torch.cat([stored(T.conv2d_same(xin, ws[0], None, False)), stored(T.conv2d_same(xin, ws[1], None, False))], dim=3)
    zo = T.conv2d_same(co, ws[2], None, False)
    check("cat", read(cat), co.detach(), ROUND[dtype])
    check("z", read(z), zo.detach(), ROUND[dtype])
    G = representable(torch.randn(zo.shape, generator=gen, dtype=torch.float64), dtype)
    fill(z.grad(), G)
    grads = torch.autograd.grad((zo * G).sum(), [xo] + ws)
    g.params.grads.zero_()
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    # first writer rounds, second writer reads that, adds and rounds again (the oracle rounds the f64 sum once): <= 2 roundings
    check("dx(accumulated)", read(x.grad()), grads[0] * (xv > 0), 2 * ROUND[dtype])
    for lay, gw in zip((l1, l2, l3), grads[1:]):
        # (half precision: a stored gradient that lands on the other side of a rounding boundary than the f64 chain's moves dW by ~1e-5)
        check("dW " + lay.name, g.params.grad(lay.kernel).double().cpu(), gw, ACC32[dtype] if dtype == "f32" else 1e-4)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("cin,cout,H,W", [(128, 96, 8, 8), (96, 64, 16, 16), (32, 16, 12, 20), (96, 64, 13, 9), (128, 64, 24, 16), (72, 48, 5, 30)])
def test_conv_transpose_2x2(eng, dtype, cin, cout, H, W):
    B = 2
    gen = _gen(cin + cout)
    g = eng.Graph("cuda", dtype)
    x = g.tensor(B, H, W, cin, relu=True, requires_grad=True)
    cat = g.tensor(B, 2 * H, 2 * W, 2 * cout, relu=True)
    lay = g.layer("t/conv2d_transpose", 2, cin, cout, "convT2")
    y = g.conv_transpose2(x, lay, out=cat.view(cout, cout), relu=True)
    y.mark_grad_written()
    g.build_backward()
    g.finalize()
    xv = representable(torch.relu(torch.randn(B, H, W, cin, generator=gen, dtype=torch.float64)), dtype)
    wv = representable(torch.randn(2, 2, cout, cin, generator=gen, dtype=torch.float64) / cin ** 0.5, dtype)
    bv = torch.randn(cout, generator=gen, dtype=torch.float64).float().double()
    set_param(g.params, lay.kernel, wv); set_param(g.params, lay.bias, bv)
    fill(x, xv)
    g.run(g.pack_ops); g.run(g.fwd_ops)
    xo, wo, bo = xv.clone().requires_grad_(True), wv.clone().requires_grad_(True), bv.clone().requires_grad_(True)
    pre = T.conv2d_transpose_s2(xo, wo, bo, False)
    check("y", read(y), torch.relu(pre).detach(), ROUND[dtype])
    assert float(cat.buf[..., :cout].abs().max()) == 0.0       # the skip half of the concat buffer is untouched
    G = torch.randn(pre.shape, generator=gen, dtype=torch.float64)
    gpre = representable(G * (pre.detach() > 0), dtype)
    fill(y.grad(), gpre)
    grads = torch.autograd.grad((pre * gpre).sum(), [xo, wo, bo])
    g.params.grads.zero_()
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    check("dx", read(x.grad()), grads[0] * (xv > 0), ROUND[dtype])
    check("dW", g.params.grad(lay.kernel).double().cpu(), grads[1], ACC32[dtype])
    check("db", g.params.grad(lay.bias).double().cpu(), grads[2], ACC32[dtype])


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,H,W,cin,cout", [(2, 8, 8, 64, 24), (1, 17, 21, 200, 96), (2, 16, 16, 136, 64), (1, 9, 35, 40, 16), (1, 16, 16, 1216, 96),
                                            (1, 16, 16, 168, 16), (1, 8, 8, 160, 24), (2, 20, 12, 328, 32),      # narrow outputs over several K-slices
                                            (2, 32, 40, 328, 32), (1, 48, 45, 1216, 96)])      # >= 2048 pixels, > 128 input channels: data gradient as GEMM tiles
@pytest.mark.parametrize("form", ["s2d", "stuffed", "s2d+mask+accumulate"])
def test_conv_transpose_3x3(eng, dtype, B, H, W, cin, cout, form, monkeypatch):
    """tf.layers.conv2d_transpose(3x3, strides 2, SAME).  bf16 / f16: the forward runs as the four output-parity sub-convolutions of
    csrc/dd_conv_ks.hip (9 real taps on the input grid); the backward on the space-to-depth output gradient (dd_space_to_depth2 +
    dd_conv3x3_ks mode 6 + dd_convt3_wgrad) or, DD_CONVT3_S2D_BWD=0, on the zero-stuffed form; f32: everything on the zero-stuffed form."""
    if dtype == "f32" and form != "stuffed":
        pytest.skip("f32 storage differentiates the zero-stuffed form only")
    monkeypatch.setenv("DD_CONVT3_S2D_BWD", "0" if form == "stuffed" else "1")
    masked = form.endswith("accumulate")
    gen = _gen(33)
    g = eng.Graph("cuda", dtype)
    x = g.tensor(B, H, W, cin, requires_grad=True, relu=masked)
    lay = g.layer("t/conv2d_transpose", 3, cin, cout, "convT3")
    y = g.conv_transpose3(x, lay, relu=True)
    y.mark_grad_written()
    if masked:
        x.mark_grad_written()      # another consumer already stored its part of dx: this layer masks by x > 0 and adds
    g.build_backward()
    g.finalize()
    assert ("ks_convt" in [getattr(op, "__name__", "") for op in g.fwd_ops]) == (dtype != "f32")
    assert ("convt3_wgrad" in [getattr(op, "__name__", "") for op in g.bwd_ops]) == (form != "stuffed")
    pw_before = eng.L.load().dd_conv_pw_count()
    xv = representable(torch.randn(B, H, W, cin, generator=gen, dtype=torch.float64), dtype)
    if masked:
        xv = torch.relu(xv)
    wv = representable(torch.randn(3, 3, cout, cin, generator=gen, dtype=torch.float64) / (3 * cin ** 0.5), dtype)
    bv = torch.randn(cout, generator=gen, dtype=torch.float64).float().double()
    set_param(g.params, lay.kernel, wv); set_param(g.params, lay.bias, bv)
    fill(x, xv)
    g.run(g.pack_ops); g.run(g.fwd_ops)
    xo, wo, bo = xv.clone().requires_grad_(True), wv.clone().requires_grad_(True), bv.clone().requires_grad_(True)
    pre = T.conv2d_transpose_s2(xo, wo, bo, False)
    check("y", read(y), torch.relu(pre).detach(), ROUND[dtype])
    G = torch.randn(pre.shape, generator=gen, dtype=torch.float64)
    gpre = representable(G * (pre.detach() > 0), dtype)
    fill(y.grad(), gpre)
    grads = torch.autograd.grad((pre * gpre).sum(), [xo, wo, bo])
    g.params.grads.zero_()
    want_dx = grads[0]
    if masked:
        g0 = representable(torch.randn(B, H, W, cin, generator=gen, dtype=torch.float64), dtype)
        fill(x.grad(), g0)
        want_dx = g0 + grads[0] * (xv > 0)
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    check("dx", read(x.grad()), want_dx, ROUND[dtype])
    check("dW", g.params.grad(lay.kernel).double().cpu(), grads[1], ACC32[dtype])
    check("db", g.params.grad(lay.bias).double().cpu(), grads[2], ACC32[dtype])
    if form != "stuffed":      # the data gradient of wide inputs on large grids runs as 256-wide GEMM tiles (conv_pw_kernel with four taps)
        assert eng.L.load().dd_conv_pw_count() - pw_before == (1 if (cin > 128 and B * H * W >= 2048) else 0)


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("accum", [False, True])
@pytest.mark.parametrize("pool,stride,H,W", [(3, 2, 16, 16), (3, 2, 10, 14), (2, 2, 8, 12), (3, 2, 9, 13), (3, 2, 1, 5), (2, 2, 7, 11), (3, 3, 9, 10)])
def test_maxpool(eng, dtype, pool, stride, H, W, accum):
    """Stride 2 with a 3x3 / 2x2 window runs the specialised kernels (one thread per 2x2 input block in the backward), anything else the generic
    ones; odd sizes put the SAME padding before the image (pad_before = 1), `accum`: the gradient is added to one already written (a skip)."""
    B, C = 2, 16
    gen = _gen(pool * 10 + H)
    g = eng.Graph("cuda", dtype)
    x = g.tensor(B, H, W, C, relu=True, requires_grad=True)
    y = g.maxpool(x, pool, stride)
    y.mark_grad_written()
    if accum:
        x.mark_grad_written()
    g.build_backward()
    g.finalize()
    xv = representable(torch.relu(torch.randn(B, H, W, C, generator=gen, dtype=torch.float64)) + 0.0, dtype)
    fill(x, xv)
    g.run(g.fwd_ops)
    xo = xv.clone().requires_grad_(True)
    yo = T.max_pool_same(xo, pool, stride)
    check("y", read(y), yo.detach(), 1e-7)
    G = representable(torch.randn(yo.shape, generator=gen, dtype=torch.float64), dtype)
    fill(y.grad(), G)
    (gx,) = torch.autograd.grad((yo * G).sum(), [xo])
    g0 = representable(torch.randn(B, H, W, C, generator=gen, dtype=torch.float64), dtype) if accum else 0.0
    if accum:
        fill(x.grad(), g0)
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    # exact ties only happen at 0 (ReLU), where the mask kills the gradient anyway
    check("dx", read(x.grad()), g0 + gx * (xv > 0), {"bf16": 4e-3 if accum else 2e-3, "f16": 5e-4 if accum else 3e-4, "f32": 1e-6}[dtype])


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("ks", [3, 5, 7])
def test_kernel_prediction_apply(lib, eng, dtype, ks):
    from deepdenoiser_amd import _lib as L
    B, H, W = 2, 12, 20
    k2 = ks * ks
    ld = 32 if ks < 7 else 56
    gen = _gen(ks)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    code = {"f32": L.DD_F32, "bf16": L.DD_BF16, "f16": L.DD_F16}[dtype]
    src = torch.randn(B, H, W, 4, generator=gen).cuda()
    lg = representable(torch.randn(B, H, W, ld, generator=gen, dtype=torch.float64) * 2, dtype)
    lgd = lg.to(tdt).cuda()
    out = torch.zeros(B, H, W, 3).cuda()
    L.check(lib.dd_kpcn_fwd(src.data_ptr(), 4, lgd.data_ptr(), ld, out.data_ptr(), 3, B, H, W, ks, code, None))
    so = src[..., :3].double().cpu()
    lo = lg[..., :k2].clone().requires_grad_(True)
    oo = T.kernel_prediction(so, lo, ks)
    check("kp out", out.cpu(), oo.detach(), ACC32[dtype])
    G = torch.randn(B, H, W, 3, generator=gen)
    (gl,) = torch.autograd.grad((oo * G.double()).sum(), [lo])
    dl = torch.full((B, H, W, ld), 7.0, dtype=tdt).cuda()
    L.check(lib.dd_kpcn_bwd(src.data_ptr(), 4, lgd.data_ptr(), ld, G.cuda().data_ptr(), 3, dl.data_ptr(), ld, ld, B, H, W, ks, code, None))
    torch.cuda.synchronize()
    check("kp dlogits", dl[..., :k2].cpu(), gl, ROUND[dtype])
    assert float(dl[..., k2:].float().abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("ks,members", [(5, 1), (3, 3), (7, 1)])
def test_kernel_prediction_from_hidden_fp32_logits(lib, eng, dtype, ks, members):
    """dd_kpcn_hidden_fwd / _bwd: logits = bb + hid @ wb computed in the launch in fp32 (Architecture.py:237-244 second conv + :260-289), for one
    member of a tuple (column offset j*k*k into wb / bb, as a COMBINED tuple passes them).  The inputs (hid in the storage type, fp32 weights) are
    exact, so the outputs are gated at fp32-accumulation level -- where the stored-logit path is gated at the storage type's rounding."""
    from deepdenoiser_amd import _lib as L
    B, H, W = 2, 12, 20
    k2 = ks * ks
    kh = k2 * members                                   # hidden channels = all members' logits (AdjustNumberOfChannels is K -> K)
    n = 4 if dtype == "f32" else 8
    ldh = (kh + n - 1) // n * n
    gen = _gen(ks * 10 + members)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    code = {"f32": L.DD_F32, "bf16": L.DD_BF16, "f16": L.DD_F16}[dtype]
    src = torch.randn(B, H, W, 4, generator=gen).cuda()
    hid = representable(torch.relu(torch.randn(B, H, W, ldh, generator=gen, dtype=torch.float64)), dtype)
    hid[..., kh:] = 3.0                                 # padding channels of the row must not be read
    wb = (torch.randn(kh, kh, generator=gen) * 4.0 / kh ** 0.5)
    bb = torch.randn(kh, generator=gen)
    hd, wbd, bbd = hid.to(tdt).cuda(), wb.cuda(), bb.cuda()
    for j in range(members):
        out = torch.zeros(B, H, W, 3).cuda()
        L.check(lib.dd_kpcn_hidden_fwd(src.data_ptr(), 4, hd.data_ptr(), ldh, kh, wbd.data_ptr() + 4 * j * k2, kh, bbd.data_ptr() + 4 * j * k2,
                                       out.data_ptr(), 3, B, H, W, ks, code, None))
        so = src[..., :3].double().cpu()
        ho = hid[..., :kh].clone().requires_grad_(True)
        lo = ho @ wb.double()[:, j * k2:(j + 1) * k2] + bb.double()[j * k2:(j + 1) * k2]
        oo = T.kernel_prediction(so, lo, ks)
        check("kp-hidden out", out.cpu(), oo.detach(), 2e-6)
        G = torch.randn(B, H, W, 3, generator=gen)
        lo2 = lo.detach().clone().requires_grad_(True)
        (gl,) = torch.autograd.grad((T.kernel_prediction(so, lo2, ks) * G.double()).sum(), [lo2])
        ldl = (kh + n - 1) // n * n
        dl = torch.full((B, H, W, ldl), 7.0, dtype=tdt).cuda()
        pad = (ldl - j * k2) if j == members - 1 else k2
        esz = 4 if dtype == "f32" else 2
        L.check(lib.dd_kpcn_hidden_bwd(src.data_ptr(), 4, hd.data_ptr(), ldh, kh, wbd.data_ptr() + 4 * j * k2, kh, bbd.data_ptr() + 4 * j * k2,
                                       G.cuda().data_ptr(), 3, dl.data_ptr() + j * k2 * esz, ldl, pad, B, H, W, ks, code, None))
        torch.cuda.synchronize()
        check("kp-hidden dlogits", dl[..., j * k2:(j + 1) * k2].cpu(), gl, ROUND[dtype])
        if j == members - 1:
            assert float(dl[..., (j + 1) * k2:].float().abs().max()) == 0.0


def test_adam_tf_form(lib):
    """dd_adam_step == tf.train.AdamOptimizer update (eps NOT bias-corrected), 4 steps, flat arena."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import math
    from deepdenoiser_amd import _lib as L
    gen = _gen(5)
    n = 10007
    p0 = torch.randn(n, generator=gen)
    p = p0.clone().cuda(); m = torch.zeros(n).cuda(); v = torch.zeros(n).cuda()
    po = [p0.double().clone()]; mo = [torch.zeros(n, dtype=torch.float64)]; vo = [torch.zeros(n, dtype=torch.float64)]
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for step in range(1, 5):
        gr = torch.randn(n, generator=gen) * torch.exp(3 * torch.randn(n, generator=gen))      # gradients over many magnitudes
        gr[::17] = 0.0
        lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        L.check(lib.dd_adam_step(p.data_ptr(), gr.cuda().data_ptr(), m.data_ptr(), v.data_ptr(), n, lr_t, b1, b2, eps, 0.5, None))
        T.adam_step(po, [0.5 * gr.double()], mo, vo, step, lr)
        torch.cuda.synchronize()
        assert float((p.double().cpu() - po[0]).abs().max()) < 2e-6
        check("m", m.cpu(), mo[0], 1e-6)
        check("v", v.cpu(), vo[0], 1e-4)    # g*g over 12 orders of magnitude in fp32


def test_stitch_and_recombine_bit_exact(lib):
    """Crop/stitch through the C-ABI == the literal restatement of Prediction.py:384-441 (bit exact); recombination :443-481."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from deepdenoiser_amd import _lib as L
    from deepdenoiser_amd.tiling import tile_plan
    from oracle import tiling_ref
    H, W, Tt = 300, 420, 128
    rng = np.random.default_rng(0)
    img = rng.standard_normal((H, W, 3)).astype(np.float32)
    plan = tile_plan(H, W)
    tiles = np.stack([img[y:y + Tt, x:x + Tt] for (y, x) in plan.windows()])
    grid = [(hi, wi) for hi in range(plan.rows.count) for wi in range(plan.cols.count)]
    table = (L.StitchEntry * len(grid))()
    for i, (hi, wi) in enumerate(grid):
        (a, b), (c, d) = plan.rows.crops[hi], plan.cols.crops[wi]
        table[i] = L.StitchEntry(i, a, b, c, d, 0, plan.rows.offsets[hi], plan.cols.offsets[wi])
    td = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).cuda()
    tiles_d = torch.tensor(tiles).cuda()
    frame = torch.full((1, H, W, 3), -7.0).cuda()
    L.check(lib.dd_stitch(tiles_d.data_ptr(), Tt, 3, frame.data_ptr(), H, W, 3, 3, td.data_ptr(), len(grid), None))
    torch.cuda.synchronize()
    assert np.array_equal(frame[0].cpu().numpy(), img)
    rows = [[tiles[hi * plan.cols.count + wi] for wi in range(plan.cols.count)] for hi in range(plan.rows.count)]
    assert np.array_equal(tiling_ref.stitch(rows, H, W), img)
    # recombination
    names = [c + s for c in ("Diffuse", "Glossy", "Subsurface", "Transmission") for s in (" Color", " Direct", " Indirect")] + ["Volume Direct", "Volume Indirect", "Environment", "Emission"]
    passes = {n: rng.standard_normal((50, 40, 3)).astype(np.float32) for n in names}
    dev = {n: torch.tensor(v).cuda() for n, v in passes.items()}
    out = torch.zeros(50, 40, 3).cuda()
    d = L.RecombineDesc()
    d.n_triples = 4
    for k, c in enumerate(("Diffuse", "Glossy", "Subsurface", "Transmission")):
        d.color[k], d.direct[k], d.indirect[k] = dev[c + " Color"].data_ptr(), dev[c + " Direct"].data_ptr(), dev[c + " Indirect"].data_ptr()
    d.n_singles = 4
    for j, n in enumerate(("Volume Direct", "Volume Indirect", "Environment", "Emission")):
        d.single[j] = dev[n].data_ptr()
    d.image = out.data_ptr()
    L.check(lib.dd_recombine(C.byref(d), 50 * 40, None))
    torch.cuda.synchronize()
    want = tiling_ref.recombine(passes)
    assert np.array_equal(out.cpu().numpy(), want)     # bit-exact: one IEEE rounding per np.multiply / np.add, in the reference's order


@pytest.mark.parametrize("seed", range(24))
def test_conv_random_shapes(eng, seed):
    """Seeded sweep over layer shapes the fixed cases do not list: channel counts around the 16 / 32 / 48 / 64 block and the 64-byte
    K-chunk boundaries, image sizes around the 16x16 workgroup tile, every epilogue combination; both dtypes against the oracle."""
    import random
    rng = random.Random(1000 + seed)
    k = rng.choice([1, 3, 3])
    cin = rng.choice([1, 3, 6, 8, 16, 24, 25, 32, 40, 48, 56, 64, 72, 96, 104, 128, 136, 160, 192, 200])
    cout = rng.choice([1, 3, 8, 16, 24, 25, 32, 40, 48, 64, 72, 80, 96, 112, 128, 144, 192])
    H, W = rng.choice([1, 5, 15, 16, 17, 31, 33]), rng.choice([2, 7, 16, 18, 32, 35])
    relu, in_relu, residual = rng.random() < 0.6, rng.random() < 0.3, rng.random() < 0.3
    x_relu = (not in_relu) and rng.random() < 0.6
    B = rng.choice([1, 2, 3])
    for dtype in ("f32", "bf16", "f16"):
        _conv_case(eng, dtype, k, cin, cout, H, W, relu, in_relu, residual, x_relu, B=B)


@pytest.mark.parametrize("seed", range(16))
def test_conv3x3_random_shapes_round2_kernels(eng, seed):
    """Seeded sweep aimed at the round-2 kernels' shape ranges (bf16 / f16): register-weight forward on 8 / 12 waves and the 6 + 2 wave kernel
    (csrc/dd_conv_rw.hip: <= 64, 65..96, 97..128 channels of depth; mask / accumulate / residual epilogues), the fused backward with and
    without a data gradient (csrc/dd_conv_bwd.hip), the single-pass 96-channel weight gradient (csrc/dd_conv_wgrad96.hip); images of several
    16x16 / 16x8 tiles with ragged edges so that the tile walks, halos and zero padding are exercised."""
    import random
    rng = random.Random(7000 + seed)
    cin = rng.choice([24, 32, 48, 64, 72, 80, 96, 104, 128, 160, 192])
    cout = rng.choice([16, 48, 64, 80, 96, 128])
    H, W = rng.choice([9, 16, 23, 40, 57]), rng.choice([8, 17, 32, 45, 70])
    relu, residual = rng.random() < 0.7, rng.random() < 0.25
    x_relu = rng.random() < 0.7
    B = rng.choice([1, 2, 3])
    for dtype in ("bf16", "f16"):
        _conv_case(eng, dtype, 3, cin, cout, H, W, relu, False, residual, x_relu, B=B, x_requires_grad=rng.random() < 0.85)


# The GEMM-tile kernels on channel VIEWS of concat buffers, which is how the Tiramisu lowering calls them (row pitch wider than the channel count,
# non-zero channel offsets, neighbours that must not be touched): the 1x1 transition conv reads buf[:, c0:c0+C] and its data gradient is masked by
# and accumulated into the same range of the gradient buffer; the transposed conv reads such a view and writes the next level's range.
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("kind,C,cout,c0,ld,H,W,B", [("1x1", 176, 176, 40, 256, 128, 128, 2), ("1x1", 80, 320, 8, 96, 181, 200, 1),
                                                      ("convT3", 200, 32, 24, 232, 48, 45, 1), ("convT3", 136, 64, 64, 264, 32, 40, 2)])
def test_gemm_tile_kernels_on_concat_views(lib, eng, dtype, kind, C, cout, c0, ld, H, W, B):
    gen = _gen(C * 7 + cout)
    g = eng.Graph("cuda", dtype)
    wide = g.tensor(B, H, W, ld, relu=False)
    wide.gstate["zero_init"] = True                       # a concat buffer: several writers, gradients accumulate
    x = wide.view(c0, C, relu=False)
    all_in = representable(torch.randn(B, H, W, ld, generator=gen, dtype=torch.float64), dtype)
    if kind == "1x1":
        lay = g.layer("v/conv2d", 1, C, cout)
        y = g.conv(x, lay, relu=False, in_relu=True)
        wshape = (1, 1, C, cout)
    else:
        out_buf = g.tensor(B, 2 * H, 2 * W, cout + 48, relu=False)
        lay = g.layer("v/conv2d_transpose", 3, C, cout, "convT3")
        y = g.conv_transpose3(x, lay, out=out_buf.view(16, cout, relu=False), relu=True)
        wshape = (3, 3, cout, C)
    y.mark_grad_written()
    g.build_backward()
    g.finalize()
    wv = representable(torch.randn(wshape, generator=gen, dtype=torch.float64) / (C ** 0.5 * (1 if kind == "1x1" else 3)), dtype)
    bv = torch.randn(cout, generator=gen, dtype=torch.float64).float().double()
    set_param(g.params, lay.kernel, wv); set_param(g.params, lay.bias, bv)
    wide.buf.copy_(all_in.to(wide.buf.dtype))
    if kind != "1x1":
        out_buf.buf.fill_(7.0)                             # the neighbours of the written range must survive
    before, wbefore = lib.dd_conv_pw_count(), lib.dd_wgrad_pw_count()
    g.run(g.pack_ops); g.run(g.fwd_ops)
    xo = all_in[..., c0:c0 + C].clone().requires_grad_(True)
    wo, bo = wv.clone().requires_grad_(True), bv.clone().requires_grad_(True)
    if kind == "1x1":
        pre = T.conv2d_same(torch.relu(xo), wo, bo, False)
        yo = pre
    else:
        pre = T.conv2d_transpose_s2(xo, wo, bo, False)
        yo = torch.relu(pre)
    check("y", read(y), yo.detach(), ROUND[dtype])
    if kind != "1x1":
        assert float((out_buf.buf[..., :16].float() - 7.0).abs().max()) == 0.0 and float((out_buf.buf[..., 16 + cout:].float() - 7.0).abs().max()) == 0.0
    G = torch.randn(pre.shape, generator=gen, dtype=torch.float64)
    gpre = representable(G * (pre.detach() > 0) if kind != "1x1" else G, dtype)
    fill(y.grad(), gpre)
    g0 = representable(torch.randn(B, H, W, ld, generator=gen, dtype=torch.float64), dtype)      # what other consumers already stored
    wide.grad().buf.copy_(g0.to(wide.buf.dtype))
    grads = torch.autograd.grad((pre * gpre).sum(), [xo, wo, bo])
    g.params.grads.zero_()
    g.run(g.bwd_ops)
    torch.cuda.synchronize()
    got = wide.grad().buf.double().cpu()
    want = g0.clone()
    want[..., c0:c0 + C] += grads[0]                        # (1x1: autograd already applied the in_relu mask; convT3: no mask)
    check("d view", got[..., c0:c0 + C], want[..., c0:c0 + C], ROUND[dtype])
    untouched = float((got[..., :c0] - g0[..., :c0]).abs().max()) == 0.0 and float((got[..., c0 + C:] - g0[..., c0 + C:]).abs().max()) == 0.0
    assert untouched, "the gradient of the neighbouring channel ranges was touched"
    check("dW", g.params.grad(lay.kernel).double().cpu(), grads[1], ACC32[dtype])
    check("db", g.params.grad(lay.bias).double().cpu(), grads[2], ACC32[dtype])
    hi, lo = max(C, cout), min(C, cout)
    pw_wgrad = kind == "1x1" and not (128 < hi <= 256 and lo > 32)      # the mid-sized weight gradient stays on the 64 x 64-slice kernel
    assert lib.dd_conv_pw_count() - before >= 1 and lib.dd_wgrad_pw_count() - wbefore == int(pw_wgrad)


# ---------------------------------------------------------------------------------------------------------------- compose net backward, op level
# dd_compose_net_bwd on the row-streaming path (caller scratch; csrc/dd_compose_stream_bwd.hip: the data launch + the weight-gradient launch) fed
# with the activations of the STORAGE-EMULATING oracle itself: no forward of ours in between, so no rounding flip of a stored activation can
# decorrelate the comparison -- what is left is the fp32 summation order of the MFMAs and the one rounding per parked gradient the emulation
# makes too (the model-level gates on reused_compose_scales/* are conditioned; this one is not).  Shapes: one strip, W > 64 (two
# weight-gradient strips), W > 128 (three forward / data strips of 96 with their 4-column halo), ragged rows and columns, several bands.
COMPOSE_BWD_SHAPES = [(2, 16, 32), (1, 20, 72), (2, 36, 136), (1, 18, 200), (3, 64, 64)]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("shape", COMPOSE_BWD_SHAPES)
def test_compose_net_backward_streaming_op_level(dtype, shape):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import ctypes as C
    from deepdenoiser_amd import _lib as L
    from oracle import model as OM
    lib = L.load()
    N, H, W = shape
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
    g = torch.Generator().manual_seed(11 + N + H + W)
    vs = OM.VarStore(torch.float64, seed=5, storage=dtype)
    vs.enter_scope("s")
    recorded = []
    conv2d = vs.conv2d

    def recording(*a, **k):
        y = conv2d(*a, **k)
        recorded.append(y)
        return y
    vs.conv2d = recording
    small = (torch.randn(N, H // 2, W // 2, 3, generator=g, dtype=torch.float64) * 0.7).float().double().requires_grad_(True)
    fine = (torch.randn(N, H, W, 3, generator=g, dtype=torch.float64) * 0.7).float().double().requires_grad_(True)
    gout = torch.randn(N, H, W, 3, generator=g, dtype=torch.float64).float().double()
    out = OM.compose_scales(vs, "s", small, fine)
    for name, v in vs.vars.items():      # glorot kernels, and biases away from zero (as after some training)
        if name.endswith("/bias"):
            with torch.no_grad():
                v.copy_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
    recorded.clear()
    vs.enter_scope("s")
    out = OM.compose_scales(vs, "s", small, fine)
    names = list(vs.vars)
    grads = torch.autograd.grad((out * gout).sum(), [small, fine] + [vs.vars[n] for n in names])
    a1, r1, a2, r3, a3, wl = [t.detach() for t in recorded]
    acts = [a1, torch.relu(r1), a2, torch.relu(r3), a3]
    dev = "cuda"
    keep = [t.to(dev, tdt).contiguous() for t in acts] + [wl.to(dev, tdt).contiguous()]
    w32 = {n: vs.vars[n].detach().float().to(dev).contiguous() for n in names}
    dws = {n: torch.zeros_like(w32[n]) for n in names}
    d_small = torch.zeros(N, H // 2, W // 2, 3, device=dev)
    d_fine = torch.zeros(N, H, W, 3, device=dev)
    sm, fi, go = small.detach().float().to(dev), fine.detach().float().to(dev), gout.float().to(dev)
    cb = L.ComposeBwdArgs()
    C.memset(C.byref(cb), 0, C.sizeof(cb))
    cb.small, cb.ld_small, cb.fine, cb.ld_fine, cb.dout, cb.ld_dout = sm.data_ptr(), 3, fi.data_ptr(), 3, go.data_ptr(), 3
    for i in range(5):
        cb.act[i], cb.ld_act[i] = keep[i].data_ptr(), 24
    cb.wl, cb.ld_wl = keep[5].data_ptr(), 1
    layer = ["s/conv2d"] + ["s/conv2d_%d" % i for i in range(1, 6)]
    cb.w_in, cb.dw_in, cb.db_in = w32[layer[0] + "/kernel"].data_ptr(), dws[layer[0] + "/kernel"].data_ptr(), dws[layer[0] + "/bias"].data_ptr()
    for i in range(4):
        cb.w_res[i], cb.dw_res[i], cb.db_res[i] = (w32[layer[1 + i] + "/kernel"].data_ptr(), dws[layer[1 + i] + "/kernel"].data_ptr(),
                                                   dws[layer[1 + i] + "/bias"].data_ptr())
    cb.w_out, cb.dw_out, cb.db_out = w32[layer[5] + "/kernel"].data_ptr(), dws[layer[5] + "/kernel"].data_ptr(), dws[layer[5] + "/bias"].data_ptr()
    cb.d_small, cb.ld_dsmall, cb.accumulate_small, cb.d_fine, cb.ld_dfine = d_small.data_ptr(), 3, 0, d_fine.data_ptr(), 3
    cb.N, cb.H, cb.W, cb.dtype = N, H, W, {"bf16": L.DD_BF16, "f16": L.DD_F16}[dtype]
    need = int(lib.dd_compose_bwd_scratch_bytes(N, H, W))
    scratch = torch.empty(need, dtype=torch.uint8, device=dev)
    cb.scratch, cb.scratch_bytes = scratch.data_ptr(), need
    L.check(lib.dd_compose_net_bwd(C.byref(cb), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    # fp32 outputs of a chain whose parked gradients are rounded once each (as the emulation rounds them): a flipped rounding of one parked
    # element moves a 24 x 216 gradient by ~1e-5 of its norm; a dropped halo column, tap or strip seam shows at >= 1e-3
    tol = 2e-4
    check("compose bwd d_small", d_small.double().cpu(), grads[0], tol)
    check("compose bwd d_fine", d_fine.double().cpu(), grads[1], tol)
    for n, gr in zip(names, grads[2:]):
        check("compose bwd d %s" % n, dws[n].double().cpu(), gr, tol)


@pytest.mark.parametrize("dtype,c,ld,ch0", [("bf16", 8, 16, 8), ("f16", 8, 16, 8), ("f32", 8, 16, 8), ("bf16", 12, 32, 16), ("bf16", 64, 64, 0), ("f32", 5, 12, 4)])
def test_colsum_segments_matches_the_per_tuple_column_sums(lib, dtype, c, ld, ch0):
    """dd_colsum_segments (round 5): the embedding-row gradients of FeatureFlags.feature_flags (FeatureFlags.py:57-67) for every tuple in one launch,
    against a float64 sum; rows land where the HOST table says (two segments sharing a row accumulate), channels outside [ch0, ch0 + c) are not read
    into the result."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd import _lib as L
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
    code = {"f32": L.DD_F32, "bf16": L.DD_BF16, "f16": L.DD_F16}[dtype]
    nseg, rows = 5, 3001
    x = torch.randn(nseg * rows, ld, generator=_gen(7)).to(tdt).cuda()
    out_row = [3, 0, 3, 6, 1]
    out = torch.zeros(8, c + 3, dtype=torch.float32).cuda()
    out[:, c:] = 7.0
    tab = (ctypes.c_int * nseg)(*out_row)
    esz = x.element_size()
    rc = lib.dd_colsum_segments(x.data_ptr() + ch0 * esz, ld, c, rows, nseg, out.data_ptr(), c + 3, tab, code, None)
    assert rc == 0, lib.dd_last_error()
    torch.cuda.synchronize()
    want = torch.zeros(8, c, dtype=torch.float64)
    xs = x.cpu().double().reshape(nseg, rows, ld)[:, :, ch0:ch0 + c].sum(1)
    for s, r in enumerate(out_row):
        want[r] += xs[s]
    got = out.cpu().double()
    assert torch.all(got[:, c:] == 7.0)
    assert float((got[:, :c] - want).abs().max()) < 1e-3 * float(want.abs().max())
    # arguments it cannot take are refused, not mis-summed
    assert lib.dd_colsum_segments(x.data_ptr() + 2, ld, c, rows, nseg, out.data_ptr(), c + 3, tab, code, None) != 0


@pytest.mark.parametrize("dtype,m0,width,nb,H,W,B", [("bf16", 16, 16, 4, 20, 28, 2), ("f16", 40, 24, 4, 16, 16, 1), ("bf16", 64, 32, 2, 17, 33, 2), ("bf16", 24, 8, 3, 16, 16, 1)])
def test_stacked_weight_gradients_of_a_dense_block_match_one_launch_per_conv(lib, dtype, m0, width, nb, H, W, B):
    """dd_conv_wgrad, stacked form (round 5): the Conv2DBackpropFilter / BiasAddGrad ops of ALL convs of a Tiramisu dense block (Tiramisu.py:26-41
    behind Training.py:701-702) from one pass over the longest prefix and the contiguous output gradients, against one plain launch per conv
    (fp32 atomics either way: summation order differs, ACC32) and against a float64 correlation for the first block."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd import _lib as L
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = L.DD_BF16 if dtype == "bf16" else L.DD_F16
    m, n = m0 + (nb - 1) * width, nb * width
    ld = ((m + n + 15) // 8) * 8
    g = _gen(21)
    buf = representable(torch.randn(B, H, W, ld, generator=g, dtype=torch.float64), dtype).to(tdt).cuda()      # [prefix | conv outputs], pre-activation
    grad = torch.zeros(B, H, W, ld, dtype=tdt).cuda()
    grad[..., m0:m0 + n] = representable(torch.randn(B, H, W, n, generator=g, dtype=torch.float64), dtype).to(tdt).cuda()
    esz = 2

    def call(stack, j=0):
        a = L.WgradArgs()
        a.p, a.ldp, a.q, a.ldq = buf.data_ptr(), ld, grad.data_ptr() + (m0 + (0 if stack else j * width)) * esz, ld
        a.B, a.H, a.W, a.taps, a.flags, a.dtype, a.ksplit, a.bias_mode = B, H, W, 9, L.IN_RELU, code, 0, 1
        if stack:
            outs = [torch.zeros(9, m0 + i * width, width).cuda() for i in range(nb)]
            bs = [torch.zeros(width).cuda() for _ in range(nb)]
            a.m, a.n, a.stack_blocks, a.stack_width, a.stack_m0 = m, n, nb, width, m0
            for i in range(nb):
                a.stack_out[i], a.stack_bias[i] = outs[i].data_ptr(), bs[i].data_ptr()
        else:
            outs, bs = [torch.zeros(9, m0 + j * width, width).cuda()], [torch.zeros(width).cuda()]
            a.m, a.n, a.out, a.bias_out = m0 + j * width, width, outs[0].data_ptr(), bs[0].data_ptr()
        assert lib.dd_conv_wgrad(ctypes.byref(a), None) == 0, lib.dd_last_error()
        torch.cuda.synchronize()
        return outs, bs
    souts, sbs = call(True)
    for j in range(nb):
        pout, pb = call(False, j)
        check("stacked dW block %d vs its own launch" % j, souts[j].cpu(), pout[0].cpu(), ACC32[dtype])
        check("stacked db block %d vs its own launch" % j, sbs[j].cpu(), pb[0].cpu(), ACC32[dtype])
    # block 0 against float64: dW[t][ci][co] = sum_p relu(x)[p + off(t)][ci] dy[p][co]
    xs = torch.relu(buf[..., :m0].cpu().double()).permute(0, 3, 1, 2)
    dy = grad[..., m0:m0 + width].cpu().double().permute(0, 3, 1, 2)
    xp = torch.nn.functional.pad(xs, (1, 1, 1, 1))
    want = torch.stack([torch.einsum("bchw,bdhw->cd", xp[:, :, a_:a_ + H, b_:b_ + W], dy) for a_ in range(3) for b_ in range(3)])
    check("stacked dW block 0 vs float64", souts[0].cpu().double(), want, ACC32[dtype])
    # a stacked call whose shapes do not add up is refused
    a = L.WgradArgs()
    a.p, a.ldp, a.m, a.q, a.ldq, a.n = buf.data_ptr(), ld, m + 8, grad.data_ptr() + m0 * esz, ld, n
    a.B, a.H, a.W, a.taps, a.dtype, a.stack_blocks, a.stack_width, a.stack_m0 = B, H, W, 9, code, nb, width, m0
    for i in range(nb):
        a.stack_out[i] = souts[i].data_ptr()
    assert lib.dd_conv_wgrad(ctypes.byref(a), None) != 0


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_weight_gradient_only_launches_side_by_side_match_their_own_launches(lib, dtype):
    """dd_conv3x3_bwd_multi (round 5): the Conv2DBackpropFilter + BiasAddGrad ops (Training.py:701-702 over UNet.py:38-48) of several layers on one
    pixel grid as one launch, against dd_conv3x3_bwd with dx = NULL per layer (fp32 atomics either way: ACC32) -- different channel counts per
    problem (96 -> 128, 128 -> 128, 72 -> 40), ragged 20 x 12 images."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd import _lib as L
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = L.DD_BF16 if dtype == "bf16" else L.DD_F16
    B, H, W = 3, 20, 12
    shapes = [(96, 128), (128, 128), (72, 40), (128, 128)]
    g = _gen(33)
    xs = [representable(torch.relu(torch.randn(B, H, W, ci, generator=g, dtype=torch.float64)), dtype).to(tdt).cuda() for ci, _ in shapes]
    dys = [representable(torch.randn(B, H, W, co, generator=g, dtype=torch.float64), dtype).to(tdt).cuda() for _, co in shapes]

    def args(arr, outs):
        for a, x, dy, (ci, co), (dw, db) in zip(arr, xs, dys, shapes, outs):
            a.dy, a.ld_dy, a.cout, a.x, a.ld_x, a.cin = dy.data_ptr(), co, co, x.data_ptr(), ci, ci
            a.dw, a.db, a.B, a.H, a.W, a.dtype = dw.data_ptr(), db.data_ptr(), B, H, W, code
    mk = lambda: [(torch.zeros(9, ci, co).cuda(), torch.zeros(co).cuda()) for ci, co in shapes]
    multi, single = mk(), mk()
    arr = (L.ConvBwdArgs * len(shapes))()
    args(arr, multi)
    assert lib.dd_conv3x3_bwd_multi(arr, len(shapes), None) == 0, lib.dd_last_error()
    arr1 = (L.ConvBwdArgs * len(shapes))()
    args(arr1, single)
    for i in range(len(shapes)):
        assert lib.dd_conv3x3_bwd(ctypes.byref(arr1[i]), None) == 0, lib.dd_last_error()
    torch.cuda.synchronize()
    for i, ((dw, db), (dw1, db1)) in enumerate(zip(multi, single)):
        check("side-by-side dW %d" % i, dw.cpu(), dw1.cpu(), ACC32[dtype])
        check("side-by-side db %d" % i, db.cpu(), db1.cpu(), ACC32[dtype])
    assert float(multi[0][0].abs().max()) > 0
    arr[1].H = H + 1      # problems on different grids are refused
    assert lib.dd_conv3x3_bwd_multi(arr, len(shapes), None) != 0
    assert lib.dd_conv3x3_bwd_multi(arr, 5, None) != 0


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_head_backward_of_three_scales_side_by_side_matches_their_own_launches(lib, dtype):
    """dd_kpcn_head_bwd_multi (round 5): the autodiff of AdjustNumberOfChannels + KernelPredictor (Architecture.py:237-289, KernelPrediction.py:11-63)
    of three scales as ONE launch against dd_kpcn_head_bwd per scale: dx bit-identical (a workgroup's pixels change, nothing inside a pixel),
    weight / bias gradients at fp32 summation order (ACC32) -- 64 / 96 / 128 channels, ragged images, one scale accumulating into dx."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd import _lib as L
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    code = L.DD_BF16 if dtype == "bf16" else L.DD_F16
    g = _gen(41)
    N, K = 3, 25
    shapes = [(64, 40, 36), (96, 20, 18), (128, 10, 9)]
    probs = []
    for C, H, W in shapes:
        d = {"x": representable(torch.relu(torch.randn(N, H, W, C, generator=g, dtype=torch.float64)), dtype).to(tdt).cuda(),
             "src": torch.randn(N, H, W, 3, generator=g).cuda(), "dout": torch.randn(N, H, W, 3, generator=g).cuda(),
             "wa": (torch.randn(C, K, generator=g) / C ** 0.5).cuda(), "ba": (0.1 * torch.randn(K, generator=g)).cuda(),
             "wb": (torch.randn(K, K, generator=g) / 5).cuda(), "bb": (0.1 * torch.randn(K, generator=g)).cuda(),
             "dx0": representable(torch.randn(N, H, W, C, generator=g, dtype=torch.float64), dtype).to(tdt).cuda()}
        probs.append(d)

    def fill(a, d, outs, accumulate):
        C = d["x"].shape[3]
        a.x, a.ldx, a.C, a.src, a.ldsrc = d["x"].data_ptr(), C, C, d["src"].data_ptr(), 3
        a.wa, a.ba, a.wb, a.bb = d["wa"].data_ptr(), d["ba"].data_ptr(), d["wb"].data_ptr(), d["bb"].data_ptr()
        a.dout, a.ld_dout, a.dx, a.ld_dx, a.accumulate_dx = d["dout"].data_ptr(), 3, outs["dx"].data_ptr(), C, accumulate
        a.dwa, a.dba, a.dwb, a.dbb = outs["dwa"].data_ptr(), outs["dba"].data_ptr(), outs["dwb"].data_ptr(), outs["dbb"].data_ptr()
        a.N, a.H, a.W, a.ksize, a.dtype = N, d["x"].shape[1], d["x"].shape[2], 5, code
    mk = lambda d: {"dx": d["dx0"].clone(), "dwa": torch.zeros_like(d["wa"]), "dba": torch.zeros_like(d["ba"]), "dwb": torch.zeros_like(d["wb"]),
                    "dbb": torch.zeros_like(d["bb"])}
    multi, single = [mk(d) for d in probs], [mk(d) for d in probs]
    arr = (L.HeadArgs * 3)()
    for k in range(3):
        fill(arr[k], probs[k], multi[k], 1 if k == 1 else 0)
    assert lib.dd_kpcn_head_bwd_multi(arr, 3, None) == 0, lib.dd_last_error()
    one = (L.HeadArgs * 3)()
    for k in range(3):
        fill(one[k], probs[k], single[k], 1 if k == 1 else 0)
        assert lib.dd_kpcn_head_bwd(ctypes.byref(one[k]), None) == 0, lib.dd_last_error()
    torch.cuda.synchronize()
    for k in range(3):
        assert torch.equal(multi[k]["dx"], single[k]["dx"]), k
        assert float(multi[k]["dx"].float().abs().max()) > 0
        for name in ("dwa", "dba", "dwb", "dbb"):
            check("side-by-side head %s scale %d" % (name, k), multi[k][name].cpu(), single[k][name].cpu(), ACC32[dtype])
    arr[2].ksize = 3
    assert lib.dd_kpcn_head_bwd_multi(arr, 3, None) != 0      # one kernel size per launch
