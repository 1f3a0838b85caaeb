"""ORACLE (test infrastructure) -- literal numpy / pure-loop restatement of the op semantics.

Independent of oracle/tf_ops.py: every function here is written straight from the index
formulas of SURVEY.md Appendix A with explicit loops (small cases only), so that the two
restatements check each other.  PARITY UNPINNED against live TensorFlow (see tf_ops.py).
float64 throughout.  NHWC.
"""

import math

import numpy as np


def same_padding(size, kernel, stride):
    out = int(math.ceil(size / stride))
    total = max((out - 1) * stride + kernel - size, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, kernel, bias=None, relu=False):
    """y[b,y,x,o] = sum_{i,j,c} x[b,y+i-p,x+j-p,c]*K[i,j,c,o] + bias, zero outside (App. A.1)."""
    b, h, w, c = x.shape
    k = kernel.shape[0]
    p = (k - 1) // 2
    o = kernel.shape[3]
    y = np.zeros((b, h, w, o))
    for yy in range(h):
        for xx in range(w):
            for i in range(k):
                for j in range(k):
                    sy, sx = yy + i - p, xx + j - p
                    if 0 <= sy < h and 0 <= sx < w:
                        y[:, yy, xx, :] += x[:, sy, sx, :] @ kernel[i, j]
    if bias is not None:
        y += bias
    return np.maximum(y, 0) if relu else y


def conv2d_transpose_s2(x, kernel, bias=None, relu=False):
    """out[o] = sum_{i,a: 2i+a-pad_before=o} x[i]*K[a]; pad_before = max(k-2,0)//2 = 0 for
    k in {2,3}; rows/cols >= 2I dropped (App. A.3).  kernel [k,k,C_out,C_in]."""
    b, h, w, c = x.shape
    k = kernel.shape[0]
    co = kernel.shape[2]
    y = np.zeros((b, 2 * h, 2 * w, co))
    for i in range(h):
        for j in range(w):
            for a in range(k):
                for bb in range(k):
                    oy, ox = 2 * i + a, 2 * j + bb
                    if oy < 2 * h and ox < 2 * w:
                        y[:, oy, ox, :] += x[:, i, j, :] @ kernel[a, bb].T
    if bias is not None:
        y += bias
    return np.maximum(y, 0) if relu else y


def max_pool_same(x, pool, stride):
    b, h, w, c = x.shape
    oh, hb, _ = same_padding(h, pool, stride)
    ow, wb, _ = same_padding(w, pool, stride)
    y = np.full((b, oh, ow, c), -np.inf)
    for i in range(oh):
        for j in range(ow):
            for a in range(pool):
                for bb in range(pool):
                    sy, sx = i * stride + a - hb, j * stride + bb - wb
                    if 0 <= sy < h and 0 <= sx < w:
                        y[:, i, j, :] = np.maximum(y[:, i, j, :], x[:, sy, sx, :])
    return y


def avg_pool_same(x, f):
    b, h, w, c = x.shape
    oh, hb, _ = same_padding(h, f, f)
    ow, wb, _ = same_padding(w, f, f)
    y = np.zeros((b, oh, ow, c))
    for i in range(oh):
        for j in range(ow):
            n = 0
            for a in range(f):
                for bb in range(f):
                    sy, sx = i * f + a - hb, j * f + bb - wb
                    if 0 <= sy < h and 0 <= sx < w:
                        y[:, i, j, :] += x[:, sy, sx, :]
                        n += 1
            y[:, i, j, :] /= n
    return y


def resize_nearest_x2(x):
    b, h, w, c = x.shape
    y = np.zeros((b, 2 * h, 2 * w, c))
    for i in range(2 * h):
        for j in range(2 * w):
            y[:, i, j, :] = x[:, i // 2, j // 2, :]
    return y


def symmetric_index(i, n):
    """index into the unpadded axis for padded coordinate i (may be <0 or >=n): mirror incl. edge."""
    if i < 0:
        return -i - 1
    if i >= n:
        return 2 * n - 1 - i
    return i


def pad_symmetric(x, pad):
    b, h, w, c = x.shape
    y = np.zeros((b, h + 2 * pad, w + 2 * pad, c))
    for i in range(h + 2 * pad):
        for j in range(w + 2 * pad):
            y[:, i, j, :] = x[:, symmetric_index(i - pad, h), symmetric_index(j - pad, w), :]
    return y


def softmax(x, axis=-1):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def kernel_prediction(src, logits, k):
    """out[b,y,x,c] = sum_{i,j} softmax(logits)[b,y,x,i*k+j] * src_sym[b, y+i-p, x+j-p, c]."""
    b, h, w, c = src.shape
    p = (k - 1) // 2
    wts = softmax(logits, axis=3)
    out = np.zeros_like(src)
    for yy in range(h):
        for xx in range(w):
            for i in range(k):
                for j in range(k):
                    sy = symmetric_index(yy + i - p, h)
                    sx = symmetric_index(xx + j - p, w)
                    out[:, yy, xx, :] += wts[:, yy, xx, i * k + j][:, None] * src[:, sy, sx, :]
    return out


def variance(x, variance_mode="uniform", relative_variance=False, compress_to_one_channel=False, epsilon=1e-4):
    b, h, w, c = x.shape
    taps = [(i, j) for i in (-1, 0, 1) for j in (-1, 0, 1)]
    if variance_mode == "neighbor":
        taps = [(0, 0), (-1, 0), (1, 0), (0, -1), (0, 1)]
    mean = np.zeros_like(x)
    mean_sq = np.zeros_like(x)
    for yy in range(h):
        for xx in range(w):
            for (i, j) in taps:
                v = x[:, symmetric_index(yy + i, h), symmetric_index(xx + j, w), :]
                mean[:, yy, xx, :] += v
                mean_sq[:, yy, xx, :] += v * v
    mean /= len(taps)
    mean_sq /= len(taps)
    r = mean_sq - mean * mean
    if relative_variance:
        r = r / np.maximum(mean * mean, epsilon)
    if compress_to_one_channel:
        r = r.mean(axis=3, keepdims=True)
    return r


def smape(p, t, eps=1e-2):
    return (np.abs(p - t) / (np.abs(p) + np.abs(t) + eps)).sum(axis=3)


def adam_scalar(theta, grads, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """TF-form Adam on a python float for a list of gradients; returns the trajectory."""
    m = v = 0.0
    out = []
    for t, g in enumerate(grads, start=1):
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
        theta = theta - lr_t * m / (math.sqrt(v) + eps)
        out.append(theta)
    return out
