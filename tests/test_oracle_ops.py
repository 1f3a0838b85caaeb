"""Pins the oracle: (1) hand-derived known answers from SURVEY Appendix A where TF != torch defaults;
(2) the torch restatement (oracle/tf_ops.py) against the independent literal numpy-loop restatement
(oracle/np_ops.py); (3) properties.  PARITY UNPINNED against live TensorFlow (not installable here)."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ops as N
from oracle import tf_ops as T

torch.manual_seed(0)
RNG = np.random.default_rng(0)


def t64(a):
    return torch.tensor(np.asarray(a), dtype=torch.float64)


def close(a, b, tol=1e-10):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.max(np.abs(a - b)) <= tol * max(1.0, np.max(np.abs(b))), np.max(np.abs(a - b))


# ------------------------------------------------------------------ known answers (App. A)
def test_same_padding_rule():
    assert T.same_padding(128, 3, 2) == (64, 0, 1)     # extra pixel goes AFTER (A.2 / A.4)
    assert T.same_padding(5, 3, 2) == (3, 1, 1)
    assert T.same_padding(6, 2, 2) == (3, 0, 0)
    assert T.same_padding(7, 3, 1) == (7, 1, 1)


def test_maxpool_3x3_s2_same_window_is_shifted_not_centred():
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    # windows rows {0,1,2},{2,3}; cols same => maxima at (2,2),(2,3),(3,2),(3,3)
    want = np.array([[10., 11.], [14., 15.]]).reshape(1, 2, 2, 1)
    close(T.max_pool_same(t64(x), 3, 2).numpy(), want)
    close(N.max_pool_same(x, 3, 2), want)
    # a centred (torch padding=1) pool would give 5,7,13,15 -- make sure we are not that
    centred = torch.nn.functional.max_pool2d(t64(x).permute(0, 3, 1, 2), 3, 2, padding=1).permute(0, 2, 3, 1).numpy()
    assert not np.allclose(centred, want)


def test_symmetric_pad_includes_edge():
    x = np.array([1., 2., 3.]).reshape(1, 1, 3, 1).repeat(3, axis=1)
    got = T.pad_symmetric(t64(x), 2).numpy()[0, 2, :, 0]
    assert got.tolist() == [2., 1., 1., 2., 3., 3., 2.]     # [b a | a b c | c b]
    close(N.pad_symmetric(x, 2), T.pad_symmetric(t64(x), 2).numpy())


def test_conv_transpose_2x2_is_pixel_shuffle_gemm():
    x = RNG.standard_normal((1, 2, 2, 3)); k = RNG.standard_normal((2, 2, 4, 3)); b = RNG.standard_normal(4)
    got = T.conv2d_transpose_s2(t64(x), t64(k), t64(b), relu=False).numpy()
    for i in range(2):
        for j in range(2):
            for a in range(2):
                for c in range(2):
                    close(got[0, 2 * i + a, 2 * j + c], k[a, c] @ x[0, i, j] + b)


def test_conv_transpose_3x3_crops_at_the_end():
    x = np.zeros((1, 2, 2, 1)); x[0, 0, 0, 0] = 1.0
    k = np.arange(1, 10, dtype=np.float64).reshape(3, 3, 1, 1)
    got = T.conv2d_transpose_s2(t64(x), t64(k)).numpy()[0, :, :, 0]
    want = np.zeros((4, 4)); want[0:3, 0:3] = k[:, :, 0, 0]     # o = 2i + a, nothing shifted to negative indices
    close(got, want)
    x = np.zeros((1, 2, 2, 1)); x[0, 1, 1, 0] = 1.0
    got = T.conv2d_transpose_s2(t64(x), t64(k)).numpy()[0, :, :, 0]
    want = np.zeros((4, 4)); want[2:4, 2:4] = k[0:2, 0:2, 0, 0]   # row/col 4 is dropped
    close(got, want)


def test_kernel_prediction_uniform_logits_is_symmetric_box_filter():
    src = RNG.standard_normal((1, 6, 6, 3))
    out = T.kernel_prediction(t64(src), torch.zeros(1, 6, 6, 25, dtype=torch.float64), 5).numpy()
    pad = N.pad_symmetric(src, 2)
    want = np.zeros_like(src)
    for y in range(6):
        for x in range(6):
            want[0, y, x] = pad[0, y:y + 5, x:x + 5].mean(axis=(0, 1))
    close(out, want)


def test_kernel_prediction_one_hot_is_shift():
    src = RNG.standard_normal((1, 6, 6, 3))
    logits = torch.full((1, 6, 6, 25), -1e4, dtype=torch.float64)
    logits[..., 1 * 5 + 3] = 0.0     # tap (i=1, j=3) -> offset (-1, +1)
    out = T.kernel_prediction(t64(src), logits, 5).numpy()
    close(out[0, 2, 2], src[0, 1, 3])
    close(out[0, 0, 5], src[0, 0, 5])   # (-1, 6) mirrors to (0, 5)


def test_smape_values_and_sign_at_zero():
    p = t64([[[[1.0, 0.0, -2.0]]]]); t = t64([[[[3.0, 0.0, 2.0]]]])
    got = float(T.loss_difference(p, t, "SMAPE")[0, 0, 0])
    assert abs(got - (2 / 4.01 + 0.0 + 4 / 4.01)) < 1e-12
    assert float(T.signed_log1p(t64([0.0]))) == 0.0 and abs(float(T.signed_expm1(t64([-1.0]))) + math.expm1(1.0)) < 1e-14


def test_adam_tf_form_three_steps_scalar():
    grads = [0.5, -1.5, 2.0]
    want = N.adam_scalar(1.0, grads, lr=0.1)
    p = [torch.tensor([1.0], dtype=torch.float64)]; m = [torch.zeros(1, dtype=torch.float64)]; v = [torch.zeros(1, dtype=torch.float64)]
    for step, g in enumerate(grads, start=1):
        T.adam_step(p, [torch.tensor([g], dtype=torch.float64)], m, v, step, 0.1)
        assert abs(float(p[0]) - want[step - 1]) < 1e-14
    # first step moves by ~lr*sign(g) (epsilon is NOT bias corrected: lr_t*m/(sqrt(v)+eps))
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    assert abs(want[0] - (1.0 - lr_t * 0.05 / (math.sqrt(0.00025) + 1e-8))) < 1e-15
    torch_adam = 1.0 - 0.1 * 0.5 / (abs(0.5) + 1e-8)    # torch.optim.Adam's first step (eps inside the corrected denominator)
    assert abs(want[0] - torch_adam) > 1e-9


# ------------------------------------------------------------------ restatement vs restatement
@pytest.mark.parametrize("k,cin,cout,h,w", [(3, 3, 4, 5, 6), (1, 4, 3, 4, 4), (3, 2, 2, 1, 7)])
def test_conv2d_same(k, cin, cout, h, w):
    x = RNG.standard_normal((2, h, w, cin)); kern = RNG.standard_normal((k, k, cin, cout)); b = RNG.standard_normal(cout)
    close(T.conv2d_same(t64(x), t64(kern), t64(b), True).numpy(), N.conv2d_same(x, kern, b, True))


@pytest.mark.parametrize("k", [2, 3])
def test_conv2d_transpose(k):
    x = RNG.standard_normal((2, 3, 4, 3)); kern = RNG.standard_normal((k, k, 5, 3)); b = RNG.standard_normal(5)
    close(T.conv2d_transpose_s2(t64(x), t64(kern), t64(b), True).numpy(), N.conv2d_transpose_s2(x, kern, b, True))


@pytest.mark.parametrize("pool,stride,h,w", [(3, 2, 8, 6), (3, 2, 5, 7), (2, 2, 6, 4), (2, 2, 5, 5)])
def test_max_pool(pool, stride, h, w):
    x = RNG.standard_normal((2, h, w, 3))
    close(T.max_pool_same(t64(x), pool, stride).numpy(), N.max_pool_same(x, pool, stride))


@pytest.mark.parametrize("f,h,w", [(2, 4, 6), (4, 8, 4), (2, 5, 5), (4, 6, 6)])
def test_avg_pool(f, h, w):
    x = RNG.standard_normal((2, h, w, 3))
    close(T.avg_pool_same(t64(x), f).numpy(), N.avg_pool_same(x, f))


def test_resize_and_kpcn_and_variance_and_smape():
    x = RNG.standard_normal((2, 3, 4, 3))
    close(T.resize_nearest_x2(t64(x)).numpy(), N.resize_nearest_x2(x))
    src = RNG.standard_normal((2, 6, 5, 3)); lg = RNG.standard_normal((2, 6, 5, 25))
    close(T.kernel_prediction(t64(src), t64(lg), 5).numpy(), N.kernel_prediction(src, lg, 5))
    lg9 = RNG.standard_normal((2, 6, 5, 9))
    close(T.kernel_prediction(t64(src), t64(lg9), 3).numpy(), N.kernel_prediction(src, lg9, 3))
    for mode in ("uniform", "neighbor"):
        for rel in (False, True):
            for comp in (False, True):
                close(T.variance(t64(src), mode, rel, comp).numpy(), N.variance(src, mode, rel, comp))
    one = RNG.standard_normal((2, 4, 4, 1))
    close(T.variance(t64(one), "uniform", True, True).numpy(), N.variance(one, "uniform", True, True))
    close(T.loss_difference(t64(src), t64(src[::-1].copy()), "SMAPE").numpy(), N.smape(src, src[::-1]))


def test_standardize_roundtrip_and_compose_properties():
    x = t64(RNG.standard_normal((1, 4, 4, 3)) * 3)
    z = T.standardize(x, True, 0.3, 2.0)
    close(T.invert_standardization(z, True, 0.3, 2.0).numpy(), x.numpy(), 1e-12)
    xn = x.numpy()
    close(z.numpy(), (np.sign(xn) * np.log1p(np.abs(xn)) - 0.3) / math.sqrt(2.0))
    # compose_scales blend: w == 0 -> identity on the fine image; w == 1 -> low frequencies replaced
    fine = t64(RNG.standard_normal((1, 4, 4, 3))); small = t64(RNG.standard_normal((1, 2, 2, 3)))
    low = T.resize_nearest_x2(T.avg_pool_same(fine, 2)); up = T.resize_nearest_x2(small)
    out1 = fine - 1.0 * low + 1.0 * up
    close(T.avg_pool_same(out1, 2).numpy(), small.numpy(), 1e-12)
