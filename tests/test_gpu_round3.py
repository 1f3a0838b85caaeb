"""-m gpu: the half-precision kernels the bench actually runs, gated at the precision of their arithmetic.

The plain f64 oracle can only bound a bf16 / fp16 network by the storage type's own accumulated error (1e-2 forward, 3.5e-2 median per
gradient tensor at full size): a kernel that drops a channel or a halo column hides under that.  The STORAGE-EMULATING oracle
(oracle.model.OracleArchitecture(storage=...)) is the same float64 graph with every tensor the half-precision path keeps in HBM rounded
where it is stored, and every activation gradient rounded where the reverse program stores it (per consumer branch, then the sum: the
epilogues of csrc/dd_conv_bwd.hip / dd_convt.hip / dd_head.hip).  Against it the fused head / compose / fused-backward / register-weight /
transposed-conv kernels differ only by fp32 summation order -- until the first value lands on the other side of a rounding boundary, after
which a deep network decorrelates (see the note above FULL_SIZE_GATES).  What it certifies: bit-faithful forwards of the small networks
(1e-7), bit-faithful backward of the fused head and compose kernels (<= 5e-6), batch striding at full size; the per-op tests
(tests/test_gpu_ops.py) carry the rounding-level gates for the conv kernels.
"""
import pytest
import torch

from deepdenoiser_amd import configs
from gpu_util import check, gate, rel_l2
from oracle import training as OT
from oracle.model import OracleArchitecture
from test_gpu_model import CASES, _inputs, _with_flags

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _emulated_step(aj, tj, dtype, feats, labels, loss_scale):
    """Predictions, loss and parameter gradients of the storage-emulating oracle.  The stored gradients of the fp16 path are `loss_scale`
    times larger (program.Program.loss_scale): the emulation rounds the scaled gradients and divides the result."""
    oracle = OracleArchitecture(aj, dtype=torch.float64, seed=2, storage=dtype)
    preds = oracle.predict(feats)
    loss = OT.model_loss(oracle, aj, tj, preds, labels)
    params = oracle.parameters()
    grads = torch.autograd.grad(loss * loss_scale, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g / loss_scale for g, p in zip(grads, params)]
    return oracle, [{k: v.detach() for k, v in d.items()} for d in preds], float(loss.detach()), grads


TRAVEL = 0.25      # a tensor whose gradient storage rounding ALONE moves by more than this fraction of its norm is "rounding-dominated", see _compare


def _compare(what, dtype, aj, tj, B, H, W, fwd_gate, loss_gate, grad_median_gate, grad_max_gate, fwd_median_gate=None, tight=None, conditioned=False):
    """tight = (name prefixes, gate): parameters whose gradient must match at summation-order level (see the bit-faithful test below).
    conditioned (round 6, replaces the bias-only allow-list of rounds 4 / 5): every tensor is classified by how far storage rounding ALONE moves
    its gradient, travel = |g_emulated - g_plain| / |g_emulated| (g_plain = the plain f64 oracle's gradient of the same network and inputs).
      * every tensor is first gated against the emulation at grad_median_gate / grad_max_gate (the maximum at <= 2x the largest value measured
        over all cases): where the forward is bit-faithful the device follows the emulation's rounding decisions and agrees with it far below
        the distance either keeps from the plain oracle;
      * a tensor that misses that gate AND has travel > TRAVEL takes the second criterion instead.  Such a tensor's gradient is a near-cancelling sum (the compose net's weights in the small Tiramisu: every one of them moves by
        63 - 126 % of its norm under storage rounding, its 24 -> 1 biases by 45x) and the emulation is ONE sample of that rounding noise: a second
        half-precision evaluation with a different fp32 summation order lands anywhere within the same distance.  What can be certified is that
        the device is AS GOOD an approximation of the true gradient as "round once where the tensor is stored" predicts:
        |g_device - g_plain| <= 1.15 |g_emulated - g_plain| + 0.02 |g_plain|.  Measured (tools/gate_diag.py, profiles/r06_a_gate_diag.txt): the device
        sits at 0.41 - 0.50 of the emulation's distance on every such tensor, fused compose kernels and layer-wise path alike (what VERDICT r5
        saw drifting, 0.447 -> 0.535 on reused_compose_scales/conv2d_5/kernel, was this noise measured against a gate of the wrong kind).
    The streaming compose backward's own gradients are gated without any conditioning, op by op, in
    tests/test_gpu_ops.py::test_compose_net_backward_streaming_op_level."""
    from deepdenoiser_amd.architecture import Architecture
    plain = OracleArchitecture(aj, dtype=torch.float64, seed=2)
    feats, labels = _inputs(plain, B, H, W)
    feats = _with_flags(aj, feats, B, H, W)
    arch = Architecture(aj, device="cuda", dtype=dtype)
    prog = arch.program(B, H, W, training_json=tj)
    oracle, preds_o, loss_o, grads_o = _emulated_step(aj, tj, dtype, feats, labels, prog.loss_scale)
    assert [p.name for p in arch.params.params] == list(oracle.vs.vars.keys())
    arch.params.load_list(list(oracle.vs.vars.values()))
    dev, devl = {k: v.cuda() for k, v in feats.items()}, {k: v.cuda() for k, v in labels.items()}
    loss = float(prog.train_step(dev, devl))
    torch.cuda.synchronize()
    preds = prog.prediction_dictionaries()
    fwd = []
    for s, (dp, do) in enumerate(zip(preds, preds_o)):
        for k in do:
            assert torch.isfinite(dp[k]).all()
            fwd.append(check("%s %s scale %d %s" % (what, dtype, s, k), dp[k].cpu(), do[k], fwd_gate))
    fwd.sort()
    if fwd_median_gate is not None:
        gate("%s %s forward median over %d predictions" % (what, dtype, len(fwd)), fwd[len(fwd) // 2], fwd_median_gate)
    gate("%s %s loss rel err" % (what, dtype), abs(loss - loss_o) / abs(loss_o), loss_gate)
    errs = []
    grads_p = OT.train_step(plain, aj, tj, feats, labels, ([], []), 1)[1] if conditioned else [None] * len(grads_o)
    for p, go, gp in zip(arch.params.params, grads_o, grads_p):
        got = arch.params.grad(p).double().cpu() / prog.loss_scale
        if float(go.norm()) == 0.0:
            assert float(got.abs().max()) < 1e-6, p.name
            continue
        e = rel_l2(got, go)
        if (gp is not None and grad_max_gate is not None and e > grad_max_gate
                and float((go - gp).norm()) > TRAVEL * float(go.norm())):      # rounding-dominated AND off the emulation: gated against the plain oracle
            d_dev, d_emu, n_plain = float((got - gp).norm()), float((go - gp).norm()), float(gp.norm())
            gate("%s %s rounding-dominated gradient %s: device-to-plain over (1.15 x emulation-to-plain + 0.02 |g|)" % (what, dtype, p.name),
                 d_dev / (1.15 * d_emu + 0.02 * n_plain), 1.0)
            continue
        errs.append((e, p.name))
        if tight is not None and any(p.name.startswith(pre) for pre in tight[0]):
            gate("%s %s gradient %s" % (what, dtype, p.name), e, tight[1])
    by_err = sorted(errs)
    med, (mx, mx_name) = by_err[len(by_err) // 2][0], by_err[-1]
    print("%s, %s storage vs the storage-emulating oracle: forward median %.2e worst %.2e, loss rel err %.2e, gradient rel-L2 median %.2e max %.2e (%s) over %d tensors"
          % (what, dtype, fwd[len(fwd) // 2], fwd[-1], abs(loss - loss_o) / abs(loss_o), med, mx, mx_name, len(errs)))
    if grad_median_gate is not None:
        gate("%s %s gradient median" % (what, dtype), med, grad_median_gate)
        gate("%s %s gradient max (%s)" % (what, dtype, mx_name), mx, grad_max_gate)
    assert torch.isfinite(arch.params.grads).all()
    return errs


# What the emulation can and cannot certify (measured, profiles/r03_parity_errors.txt and tools/emu_debug.py):
#   * a half-precision network is CHAOTIC in its rounding decisions: a value that lands on the other side of a rounding boundary than the f64
#     chain's (fp32 vs f64 summation, probability ~1e-4 per element) perturbs 9 x C_out x |w| >> 1 elements of the next layer by up to an ulp
#     each, so one flip decorrelates everything downstream within a few layers -- the comparison is all-or-nothing.  Small networks: most
#     predictions agree to 1e-7 (the fused head / compose / conv kernels are bit-faithful to "round once where the tensor is stored"), the few
#     hit by a flip sit at one-to-two roundings.  Full size: every prediction carries flips (5.5e-3 bf16, 7e-4 fp16) and per-tensor gradients
#     are no closer to the emulation than to the plain oracle (their sums cancel heavily, which amplifies rounding-level differences).
#   * so: forward gates at 3 roundings (max) and 1e-5 (median, small bf16 networks); gradient gates at the storage type's own error here, at
#     summation-order level in test_backward_of_the_fused_head_and_compose_kernels_is_bit_faithful below, and per op in tests/test_gpu_ops.py.
FULL_SIZE_GATES = {"bf16": (1.2e-2, 5e-4, 0.045, 0.16), "f16": (1.5e-3, 1e-4, 0.016, 0.055)}


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_cfg2_full_size_half_precision_against_the_storage_emulating_oracle(dtype):
    """The bench's configuration (cfg-2, 128x128, real filters, 5x5 kernel prediction, 3 scales) at B = 2 -- batch striding on every round-2
    kernel: register-weight forward, fused 3x3 backward, weight-gradient role, transposed-conv, fused head and compose kernels."""
    _need_gpu()
    aj, tj = configs.cfg2_unet_kpcn(), configs.bench_training()
    _compare("cfg-2 128x128 B=2", dtype, aj, tj, 2, 128, 128, *FULL_SIZE_GATES[dtype])


SMALL_GATES = {"bf16": (1.5e-2, 1e-3, 0.1, 0.22), "f16": (4e-3, 2e-4, 0.1, 0.22)}      # gradient max: measured <= 0.109 over all cases (profiles/r06_parity_errors.txt)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("case", ["example_json_single_embedding", "cfg2_unet_kpcn_real_filters", "combined_tuples_kp3", "tiramisu_multiscale",
                                  "ragged_tile_three_scales", "one_hot_no_multiscale_raw_kp_source", "invert_before_multiscale"])
def test_small_networks_half_precision_against_the_storage_emulating_oracle(case, dtype):
    """Every structural variant of tests/test_gpu_model.py on the half-precision path: fused head with 3x3 / 5x5 kernels, layer-wise head for
    COMBINED tuples, compose net on ragged tiles, Tiramisu's dense concats and 3x3/s2 transposed convs, embedding gradients."""
    _need_gpu()
    aj, B, H, W = CASES[case]
    single_feature = len(aj["combined_features"]) == 1
    tj = configs.bench_training() if single_feature else configs.training()
    fwd_gate, loss_gate, gmed, gmax = SMALL_GATES[dtype]
    if case == "one_hot_no_multiscale_raw_kp_source":
        # kernel prediction on the RAW source + expm1 inversion: predictions reach exp(46) here, where the SMAPE gradient (2t + eps) / (p + t + eps)^2
        # cancels catastrophically in fp32 (device and TensorFlow alike; measured 1e-2 against f64 on the f32 path too): forward and loss only
        gmed = gmax = None
    # bf16 networks with many tuples: the majority of the predictions must be bit-faithful (fp16: subnormal handling differs from torch's)
    many = dtype == "bf16" and case in ("example_json_single_embedding", "ragged_tile_three_scales", "combined_tuples_kp3", "invert_before_multiscale")
    _compare(case, dtype, aj, tj, B, H, W, fwd_gate, loss_gate, gmed, gmax, fwd_median_gate=1e-5 if many else None, conditioned=True)


BIT_FAITHFUL = {
    # one feature tuple, so that no other tuple's rounding flips leak into the shared weights' gradients
    "k5_two_scales": (configs.architecture(filters=(16, 16), convs=1, invert_after_multiscale=False, flag_mode="NONE",
                                           combined={"Emission": {"Color": "Emission", "Direct": "", "Indirect": ""}}), 2, 32, 16),
    "k3_two_scales": (configs.architecture(filters=(16, 16), convs=1, kernel_size=3, flag_mode="NONE",
                                           combined={"Emission": {"Color": "Emission", "Direct": "", "Indirect": ""}}), 2, 16, 32),
    "k5_three_scales_ragged": (configs.architecture(filters=(16, 16, 24), convs=1, flag_mode="NONE",
                                                    combined={"Emission": {"Color": "Emission", "Direct": "", "Indirect": ""}}), 1, 24, 40),
}


@pytest.mark.parametrize("case", list(BIT_FAITHFUL))
def test_backward_of_the_fused_head_and_compose_kernels_is_bit_faithful(case):
    """bf16, tiny single-tuple networks whose forward is bit-faithful to the emulation: the gradients the reverse program computes FIRST -- the
    compose net's six layers (csrc/dd_compose.hip backward, 12 waves) and the 1x1 head layers of every scale (csrc/dd_head.hip backward) -- must
    agree with the storage-emulating oracle at summation-order level (measured 3e-8 ... 5e-6), two to three orders of magnitude below one
    bf16 rounding.  Further upstream the comparison decays through rounding flips (see above) and is only sanity-gated.
    If this fails after a change that reorders an fp32 summation, look at the forward first: a single new flip there explains it."""
    _need_gpu()
    aj, B, H, W = BIT_FAITHFUL[case]
    tj = configs.bench_training()
    n_scales = len(aj["architecture"]["core_architecture"]["number_of_filters_for_convolution_blocks"])
    # head layers = the last 2 * n_scales conv2d of the core scope (variable-creation order)
    oracle = OracleArchitecture(aj, dtype=torch.float64, seed=2)
    oracle.predict(_inputs(oracle, B, H, W)[0])
    core_convs = [n for n in oracle.vs.vars if n.startswith("reused_core_architecture/conv2d") and "transpose" not in n and n.endswith("/kernel")]
    head = [n[:-len("kernel")] for n in core_convs[-2 * n_scales:]]
    errs = _compare("bit-faithful " + case, "bf16", aj, tj, B, H, W, 1e-5, 1e-4, 1e-4, 6e-3, tight=(["reused_compose_scales/"] + head, 1e-4))
    assert len(errs) > 10


@pytest.mark.parametrize("dtype,grad_median_gate,grad_max_gate", [("bf16", 0.045, 0.16), ("f16", 0.016, 0.055)])
def test_half_precision_full_size_error_against_the_plain_oracle_max_gated(dtype, grad_median_gate, grad_max_gate):
    """The storage type's OWN error at full size (cfg-2, 128x128, B = 1) against the plain f64 oracle, with the maximum over the 68 gradient
    tensors gated as well as the median (round 2 printed it: bf16 1.08e-1, fp16 3.6e-2)."""
    _need_gpu()
    from test_gpu_model import _pair
    aj, tj, B, H, W = configs.cfg2_unet_kpcn(), configs.bench_training(), 1, 128, 128
    oracle, arch, prog, feats, labels, dev, devl, preds_o = _pair(aj, dtype, B, H, W, tj)
    loss_o, grads_o = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
    prog.train_step(dev, devl)
    torch.cuda.synchronize()
    errs = sorted(rel_l2(arch.params.grad(p).cpu() / prog.loss_scale, go) for p, go in zip(arch.params.params, grads_o) if float(go.norm()) > 0)
    print("%s storage vs the plain oracle, cfg-2 128x128: gradient rel-L2 median %.3e max %.3e" % (dtype, errs[len(errs) // 2], errs[-1]))
    gate("%s plain-oracle gradient median" % dtype, errs[len(errs) // 2], grad_median_gate)
    gate("%s plain-oracle gradient max" % dtype, errs[-1], grad_max_gate)


# ---------------------------------------------------------------------------------------------------------------- BASELINE config 3
def _plain_training_parity(what, aj, B, H, W, dtype, fwd_gate, loss_gate, grad_median_gate, grad_max_gate, tweak=None):
    from test_gpu_model import _pair
    tj = configs.bench_training()
    oracle, arch, prog, feats, labels, dev, devl, preds_o = _pair(aj, dtype, B, H, W, tj, tweak=tweak)
    preds = arch.predict(dev)
    torch.cuda.synchronize()
    worst = max(check("%s %s scale %d %s" % (what, dtype, s, k), dp[k].cpu(), do[k], fwd_gate) for s, (dp, do) in enumerate(zip(preds, preds_o)) for k in do)
    loss_o, grads_o = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
    loss = float(prog.train_step(dev, devl))
    torch.cuda.synchronize()
    gate("%s %s loss rel err" % (what, dtype), abs(loss - float(loss_o)) / abs(float(loss_o)), loss_gate)
    errs = sorted((rel_l2(arch.params.grad(p).cpu() / prog.loss_scale, go), p.name) for p, go in zip(arch.params.params, grads_o) if float(go.norm()) > 0)
    print("%s, %s: forward worst rel-L2 %.2e, gradient rel-L2 median %.2e max %.2e (%s) over %d tensors" % (what, dtype, worst, errs[len(errs) // 2][0], errs[-1][0], errs[-1][1], len(errs)))
    gate("%s %s gradient median" % (what, dtype), errs[len(errs) // 2][0], grad_median_gate)
    gate("%s %s gradient max (%s)" % (what, dtype, errs[-1][1]), errs[-1][0], grad_max_gate)
    assert torch.isfinite(arch.params.grads).all()
    return prog


def test_cfg3_full_size_training_step_parity_f32():
    """BASELINE config 3 at its real size: Tiramisu F = [16, 24, 32] x 4 + 5x5 kernel prediction + 3 scales on a 256x256 tile, B = 1: predictions,
    loss and every parameter gradient of the f32 path against the f64 oracle (round 2 compared the training step at 32x32, n = 2 only)."""
    _need_gpu()
    _plain_training_parity("cfg-3 256x256", configs.cfg3_tiramisu(filters=(16, 24, 32), convs=4), 1, 256, 256, "f32", 1e-4, 2e-5, 5e-4, 4e-3)      # measured 2.6e-4 / 1.6e-3


# Measured (bf16 / f16): forward 4.4e-2 / 5e-3, loss 2.0e-5 / 9.9e-6, gradient median 0.70 / 0.30, max 1.47 / 0.70 -- the same figures to four
# digits with DD_CONV_PW=0 DD_CONVT3_S2D_BWD=0 and with every round-3 kernel off (DD_CONV_KS=0 DD_CONVT3_PARITY=0 DD_DENSE_GATHER=0 as well): as
# for the heavy configuration below, this random 60-layer net turns half-precision rounding into large weight changes of the kernel-
# prediction softmax, whichever kernels run it.  The loss is gated tightly; the gradient gates are sanity bounds.  Round 4: with fp32 logits
# (dd_kpcn_hidden_*) bf16 0.69 / 1.56, f16 0.26 / 0.55 -- unchanged for bf16: the error arrives in the backbone's activations (x logits of ~50);
# test_cfg3_full_size_half_precision_gradients_with_unit_logits below is the gate that certifies the kernels at this size.
@pytest.mark.parametrize("dtype,gates", [("bf16", (7e-2, 1e-3, 1.0, 2.5)), ("f16", (1e-2, 1e-3, 0.5, 1.2))])
def test_cfg3_full_size_half_precision_runs_the_gemm_tile_kernels(dtype, gates):
    """BASELINE config 3 at its real size in the storage types the bench runs it in (256x256, B = 1: 65 536 pixels at the first level, where the
    1x1 transition conv, the head's data gradients and the transposed conv's backward take the GEMM-tile kernels of csrc/dd_conv_pw.hip inside the
    real network -- channel views of the concat buffers, accumulating gradients), against the plain f64 oracle at the storage type's own error."""
    _need_gpu()
    from deepdenoiser_amd import _lib
    lib = _lib.load()
    before, wbefore = lib.dd_conv_pw_count(), lib.dd_wgrad_pw_count()
    _plain_training_parity("cfg-3 256x256", configs.cfg3_tiramisu(filters=(16, 24, 32), convs=4), 1, 256, 256, dtype, *gates)
    assert lib.dd_conv_pw_count() - before >= 4 and lib.dd_wgrad_pw_count() - wbefore >= 2, "the GEMM-tile kernels did not run"


def _tame_logits(scale):
    """Scale the kernel-prediction head's LAST 1x1 layers (kernel and bias) of every scale: the random 60-layer Tiramisu feeds logits of magnitude
    ~50 into the softmax, where any half-precision rounding of the backbone's activations (0.4 % of 50 = 0.2) moves a softmax weight by tens of
    percent -- with logits of order 1 the same network measures the kernels, not the conditioning of a random initialisation."""
    def tweak(oracle):
        convs = [n for n in oracle.vs.vars if n.startswith("reused_core_architecture/conv2d") and "transpose" not in n and n.endswith("/kernel")]
        n_scales = 3
        for n in convs[-2 * n_scales:][1::2]:      # variable-creation order: (1x1 C -> K, 1x1 K -> K) per scale
            with torch.no_grad():
                oracle.vs.vars[n].mul_(scale)
                oracle.vs.vars[n[:-len("kernel")] + "bias"].mul_(scale)
    return tweak


# Round 4: fp32 logits in the layer-wise head (dd_kpcn_hidden_*) move the heavy configuration's bf16 gradient median 0.65 -> 0.43
# (fp16 0.11 -> 0.06) but NOT the light one's (0.69 -> 0.69): what reaches the softmax is the backbone's half-precision activations times logits
# of magnitude 50, not the rounding of the stored logits.  With the head's last layer scaled so that the logits are of order 1 the SAME network,
# at full size, through the same kernels, is gated at rounding level.
# Measured (bf16 / f16): forward 7.2e-3 / 9.5e-4, gradient median 2.4e-2 / 1.3e-2, max 5.3e-2 / 5.6e-2 over the 74 tensors (cfg-2's gates: 0.045 / 0.16).
@pytest.mark.parametrize("dtype,gates", [("bf16", (2e-2, 1e-3, 0.05, 0.12)), ("f16", (3e-3, 1e-3, 0.03, 0.12))])
def test_cfg3_full_size_half_precision_gradients_with_unit_logits(dtype, gates):
    _need_gpu()
    _plain_training_parity("cfg-3 256x256, logits / 32", configs.cfg3_tiramisu(filters=(16, 24, 32), convs=4), 1, 256, 256, dtype, *gates, tweak=_tame_logits(1.0 / 32))


# Measured (bf16 / f16): forward 4.4e-3 / 6.0e-4, gradient median 1.8e-2 / 8.0e-3, max 6.2e-2 / 2.2e-2 over the 74 tensors.
@pytest.mark.parametrize("dtype,gates", [("bf16", (1.5e-2, 2e-2, 0.04, 0.15)), ("f16", (2e-3, 3e-3, 0.02, 0.06))])
def test_cfg3_heavy_filters_gradients_with_unit_logits(dtype, gates):
    """The heavy Tiramisu (K-streamed kernel, 17 K-slices, 1 216-channel transposed convs) with the head's last layer scaled to logits of order 1:
    the gradient gate that means something for this configuration (see test_cfg3_full_size_half_precision_gradients_with_unit_logits)."""
    _need_gpu()
    _plain_training_parity("cfg-3 heavy 64x64, logits / 32", configs.cfg3_tiramisu(filters=(64, 96, 128), convs=4), 1, 64, 64, dtype, *gates, tweak=_tame_logits(1.0 / 32))


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16"])
def test_cfg3_heavy_filters_deep_reduction_parity(dtype):
    """The heavy Tiramisu, F = [64, 96, 128] x 4 (12.9 M parameters), on a 64x64 tile: implicit-GEMM reductions up to K = 9 x 1 088 = 9 792 and
    the 1 216-channel 3x3/s2 transposed convs, forward and training step.  f32: the LDS-weight kernels (<= 1e-4); bf16 / f16: the K-streamed
    kernel of csrc/dd_conv_ks.hip (17 K-slices, channel blocks 64 + 32, the four-parity transposed conv) at the storage type's own error."""
    _need_gpu()
    aj = configs.cfg3_tiramisu(filters=(64, 96, 128), convs=4)
    # half precision: this random 60-layer net feeds logits of magnitude ~50 into the kernel-prediction softmax, which turns a logit rounding of
    # 0.4 % into weight changes of tens of percent -- forward (a convex combination) and loss stay within the storage type's tolerance, the
    # gradients do not (bf16 median 0.65, fp16 0.11; identical with every kernel switch, tools/heavy_check.py): gated for finiteness and sanity only
    gates = {"f32": (1e-4, 2e-5, 1e-3, 2e-2), "bf16": (3e-2, 2e-2, 1.0, 3.0), "f16": (5e-3, 3e-3, 0.2, 0.4)}[dtype]
    prog = _plain_training_parity("cfg-3 heavy 64x64", aj, 1, 64, 64, dtype, *gates)
    if dtype != "f32":
        names = [getattr(op, "__name__", "") for op in prog.g.fwd_ops]
        assert names.count("ks_fwd") == 20 and names.count("ks_convt") == 2, "the dense blocks / transposed convs did not take the K-streamed kernel"
