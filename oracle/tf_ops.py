"""ORACLE (test infrastructure, not product code) -- TensorFlow-1.x op semantics on torch-CPU.

PARITY UNPINNED: TensorFlow is not installable in the build container and the reference
ships no tests / golden vectors / checkpoints, so these restatements are
anchored on the reference's call sites plus the published TF-1.x op semantics (SURVEY.md
Appendix A), cross-checked against an independent literal numpy-loop restatement
(oracle/np_ops.py) and hand-derived known answers (tests/test_oracle_ops.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

All tensors are NHWC (`channels_last`), any float dtype (float64 for the parity gate).
Differentiable through torch autograd, which provides the backward oracle.
"""

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- layout helpers
def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def same_padding(size, kernel, stride):
    """TF SAME rule: out=ceil(in/s); extra pixel goes AFTER."""
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    before = total // 2
    return out, before, total - before


# ----------------------------------------------------------------------------- convolutions
def conv2d_same(x, kernel_hwio, bias=None, relu=False):
    """tf.layers.conv2d(padding='same', strides 1), kernel HWIO, cross-correlation.
    Call sites: UNet.py:29-31, Tiramisu.py:35-37,50-52,77-79, Architecture.py:238-243,
    MultiScalePrediction.py:64-66,73-75,88-90."""
    k = kernel_hwio.shape[0]
    assert k % 2 == 1 and kernel_hwio.shape[1] == k
    w = kernel_hwio.permute(3, 2, 0, 1).contiguous()       # (contiguous: torch-CPU's slow_conv2d backward refuses a strided grad_weight when C_out == 1)
    y = F.conv2d(_nchw(x), w, bias, stride=1, padding=(k - 1) // 2)
    y = _nhwc(y)
    return torch.relu(y) if relu else y


def conv2d_transpose_s2(x, kernel_hwoi, bias=None, relu=False):
    """tf.layers.conv2d_transpose(strides=(2,2), padding='same'); kernel [k,k,C_out,C_in].
    k=2 (UNet.py:56-58): out[2i+a,2j+b] = sum_ci x[i,j]K[a,b,:,ci]; k=3 (Tiramisu.py:62-64):
    o = 2i+a, the extra row/col past 2I-1 is dropped at the END."""
    k = kernel_hwoi.shape[0]
    assert k in (2, 3)
    w = kernel_hwoi.permute(3, 2, 0, 1)  # torch conv_transpose weight: [C_in, C_out, kh, kw]
    y = F.conv_transpose2d(_nchw(x), w, bias, stride=2, padding=0)
    hh, ww = 2 * x.shape[1], 2 * x.shape[2]
    y = _nhwc(y[:, :, :hh, :ww])
    return torch.relu(y) if relu else y


# ----------------------------------------------------------------------------- pooling / resize
def max_pool_same(x, pool, stride):
    """tf.layers.max_pooling2d(padding='same'): padded cells never win.
    UNet.py:42-44 (3x3/s2), Tiramisu.py:55-57 (2x2/s2)."""
    _, hb, ha = same_padding(x.shape[1], pool, stride)
    _, wb, wa = same_padding(x.shape[2], pool, stride)
    xp = F.pad(_nchw(x), (wb, wa, hb, ha), value=float("-inf"))
    return _nhwc(F.max_pool2d(xp, pool, stride))


def avg_pool_same(x, factor):
    """tf.layers.average_pooling2d(x, f, f, 'same'): divisor counts valid cells only
    (SURVEY App. A.5).  MultiScalePrediction.py:11-13."""
    if factor == 1:
        return x
    _, hb, ha = same_padding(x.shape[1], factor, factor)
    _, wb, wa = same_padding(x.shape[2], factor, factor)
    xc = _nchw(x)
    if hb == ha == wb == wa == 0:
        return _nhwc(F.avg_pool2d(xc, factor, factor))
    ones = torch.ones_like(xc[:1, :1])
    num = F.avg_pool2d(F.pad(xc, (wb, wa, hb, ha)), factor, factor)
    den = F.avg_pool2d(F.pad(ones, (wb, wa, hb, ha)), factor, factor)
    return _nhwc(num / den)


def resize_nearest_x2(x):
    """tf.image.resize_images(NEAREST_NEIGHBOR), align_corners=False: out[y,x]=in[y//2,x//2]
    (MultiScalePrediction.py:16-33)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def pad_symmetric(x, pad):
    """tf.pad(..., 'SYMMETRIC') on H and W: mirror INCLUDING the edge sample
    (Conv2dUtilities.py:77-95)."""
    if pad == 0:
        return x

    def idx(n):
        return torch.tensor(list(range(pad - 1, -1, -1)) + list(range(n)) + list(range(n - 1, n - 1 - pad, -1)))

    x = x.index_select(1, idx(x.shape[1]))
    return x.index_select(2, idx(x.shape[2]))


# ----------------------------------------------------------------------------- elementwise
def signed_log1p(x):
    """Utilities.py:3-4."""
    return torch.sign(x) * torch.log1p(torch.abs(x))


def signed_expm1(x):
    """Utilities.py:6-7."""
    return torch.sign(x) * torch.expm1(torch.abs(x))


def standardize(x, use_log1p, mean, variance):
    """FeatureStandardization.standardize, Architecture.py:39-46."""
    if use_log1p:
        x = signed_log1p(x)
    if mean != 0.0:
        x = x - mean
    if variance != 1.0:
        x = x / math.sqrt(variance)
    return x


def invert_standardization(x, use_log1p, mean, variance):
    """FeatureStandardization.invert_standardization, Architecture.py:48-55."""
    if variance != 1.0:
        x = x * math.sqrt(variance)
    if mean != 0.0:
        x = x + mean
    if use_log1p:
        x = signed_expm1(x)
    return x


# ----------------------------------------------------------------------------- feature engineering
def local_mean(x, variance_mode="uniform"):
    """FeatureEngineering._local_mean (FeatureEngineering.py:11-55): symmetric pad 1, then a
    per-channel 3x3 VALID correlation with a normalised box ('uniform', /9) or plus-shaped
    ('neighbor', /5) filter."""
    c = x.shape[3]
    if variance_mode == "uniform":
        f = torch.ones(3, 3, dtype=x.dtype)
    else:
        assert variance_mode == "neighbor"
        f = torch.tensor([[0.0, 1.0, 0.0], [1.0, 1.0, 1.0], [0.0, 1.0, 0.0]], dtype=x.dtype)
    f = f / f.sum()
    w = f.reshape(1, 1, 3, 3).repeat(c, 1, 1, 1)
    xp = _nchw(pad_symmetric(x, 1))
    return _nhwc(F.conv2d(xp, w, groups=c))


def variance(x, variance_mode="uniform", relative_variance=False, compress_to_one_channel=False, epsilon=1e-4):
    """FeatureEngineering.variance (FeatureEngineering.py:57-70)."""
    mean = local_mean(x, variance_mode)
    sq_mean = mean * mean
    mean_sq = local_mean(x * x, variance_mode)
    result = mean_sq - sq_mean
    if relative_variance:
        result = result / torch.clamp(sq_mean, min=epsilon)
    if compress_to_one_channel:
        result = result.mean(dim=3, keepdim=True)
    return result


# ----------------------------------------------------------------------------- kernel prediction
def kernel_prediction(inputs, kernel_inputs, kernel_size, use_softmax=True):
    """KernelPrediction.kernel_prediction (KernelPrediction.py:11-63): softmax over the k*k
    channels (:23), symmetric pad (:30), taps stacked rows-outer / cols-inner (:32-33),
    out[c] = sum_taps w[tap] * shifted_src[c] (:58)."""
    assert inputs.shape[1:3] == kernel_inputs.shape[1:3]
    assert kernel_inputs.shape[3] == kernel_size ** 2
    pad = (kernel_size - 1) // 2
    if use_softmax:
        kernel_inputs = torch.softmax(kernel_inputs, dim=3)
    h, w = inputs.shape[1], inputs.shape[2]
    padded = pad_symmetric(inputs, pad)
    out = torch.zeros_like(inputs)
    for i in range(kernel_size):
        for j in range(kernel_size):
            out = out + padded[:, i:i + h, j:j + w, :] * kernel_inputs[:, :, :, i * kernel_size + j:i * kernel_size + j + 1]
    return out


# ----------------------------------------------------------------------------- losses
def loss_difference(predicted, target, kind="SMAPE", epsilon=1e-2):
    """LossDifference.difference (LossDifference.py:15-35); result is summed over axis 3."""
    if kind == "DIFFERENCE":
        r = predicted - target
    elif kind == "ABSOLUTE":
        r = torch.abs(predicted - target)
    elif kind == "SMOOTH_ABSOLUTE":
        a = torch.abs(predicted - target)
        r = torch.where(a < 1, 0.5 * a * a, a - 0.5)
    elif kind == "SQUARED":
        r = (predicted - target) ** 2
    elif kind == "SMAPE":
        r = torch.abs(predicted - target) / (torch.abs(predicted) + torch.abs(target) + epsilon)
    else:
        raise KeyError(kind)
    return r.sum(dim=3)


# ----------------------------------------------------------------------------- optimizer
def adam_step(params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (Training.py:701-702), TF formulation:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps).  In-place; step is 1-based."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    with torch.no_grad():
        for p, g, mm, vv in zip(params, grads, m, v):
            mm.mul_(beta1).add_(g, alpha=1.0 - beta1)
            vv.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            p.sub_(lr_t * mm / (vv.sqrt() + eps))


def glorot_uniform_(tensor, fan_in, fan_out, generator):
    """TF default kernel initialiser for tf.layers.* / tf.get_variable."""
    limit = math.sqrt(6.0 / (fan_in + fan_out))
    with torch.no_grad():
        tensor.copy_((torch.rand(tensor.shape, generator=generator, dtype=torch.float64) * 2 - 1) * limit)
    return tensor
