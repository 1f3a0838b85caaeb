"""A stand-in `tensorflow` module for tests/golden/make_wiring_golden.py -- FIXTURE GENERATION ONLY.

The reference's hot path is Python on TensorFlow 1.x, which is not installable here.  This module lets the reference's OWN graph-building code
(Architecture.predict, UNet / Tiramisu, SourceEncoder, FeatureEngineering, KernelPrediction, MultiScalePrediction, LossDifference and
Training.model_fn) run eagerly on torch float64 tensors: the ~60 `tf.*` names those files touch are implemented here, each as a thin call into
oracle/tf_ops.py (the restated op semantics of SURVEY Appendix A) or a one-line torch expression.

What this buys and what it does not: it pins NO TensorFlow arithmetic (the ops are this repository's restatements, so `parity` stays
"unpinned" for them).  It turns the reference's WIRING -- which tensor is sliced, concatenated, scaled, pooled, fed to which layer, in which
variable scope, with which loss weight -- from a restatement (oracle/model.py, oracle/training.py) into executed reference code; the unpinned
surface shrinks to the primitive ops listed in SURVEY Appendix A.

Variable scopes follow TF 1.x: tf.layers.* without a name take `conv2d`, `conv2d_1`, ... unique within the enclosing variable scope;
entering a variable scope by name restarts the counters below it; reuse=False creating an existing variable and reuse=True reading a missing
one both raise, as TensorFlow does (a wiring error must not be papered over).  Variables are float64, Glorot-uniform / zeros from a seeded generator, recorded in
creation order.
"""
import contextlib
import math
import os
import sys
import types

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import tf_ops as T  # noqa: E402

DTYPE = torch.float64
AUTO_REUSE = "AUTO_REUSE"
float32 = "float32"      # dtype tokens are accepted and ignored: every tensor is float64


class _Store:
    def __init__(self):
        self.reset(0)

    def reset(self, seed):
        self.vars = {}              # name -> tensor (insertion order = creation order)
        self.gen = torch.Generator().manual_seed(seed)
        self.scopes = []            # [(full name, reuse)]
        self.counters = {}          # scope full name -> {base layer name: count}
        self.log = []               # (variable name, 'create' | 'reuse')

    def scope_name(self):
        return self.scopes[-1][0] if self.scopes else ""

    def reuse(self):
        return self.scopes[-1][1] if self.scopes else False

    def unique_layer_name(self, base):
        c = self.counters.setdefault(self.scope_name(), {})
        n = c.get(base, 0)
        c[base] = n + 1
        return base if n == 0 else "%s_%d" % (base, n)

    def get(self, name, shape, init):
        full = (self.scope_name() + "/" if self.scope_name() else "") + name
        reuse = self.reuse()
        if full in self.vars:
            if reuse is False or reuse is None:
                raise ValueError("Variable %s already exists, disallowed. Did you mean to set reuse=True?" % full)
            self.log.append((full, "reuse"))
        else:
            if reuse is True:
                raise ValueError("Variable %s does not exist, or was not created with tf.get_variable()." % full)
            t = torch.zeros(tuple(int(s) for s in shape), dtype=DTYPE)
            init(t)
            t.requires_grad_(True)
            self.vars[full] = t
            self.log.append((full, "create"))
        v = self.vars[full]
        assert tuple(v.shape) == tuple(int(s) for s in shape), (full, tuple(v.shape), tuple(shape))
        return v


STORE = _Store()


@contextlib.contextmanager
def name_scope(name, *a, **k):
    yield name


@contextlib.contextmanager
def variable_scope(name, reuse=None, **k):
    outer = STORE.scope_name()
    full = (outer + "/" if outer else "") + name
    inherited = STORE.reuse()
    eff = reuse if (reuse is not None and reuse is not False) else (inherited if inherited else reuse)
    # entering a variable scope by name restarts tf.layers' default-name counters below it
    for k2 in [k2 for k2 in STORE.counters if k2 == full or k2.startswith(full + "/")]:
        del STORE.counters[k2]
    STORE.scopes.append((full, eff))
    try:
        yield full
    finally:
        STORE.scopes.pop()


def _glorot(fan_in, fan_out):
    def init(t):
        T.glorot_uniform_(t, fan_in, fan_out, STORE.gen)
    return init


def get_variable(name, shape, trainable=True, **k):
    shape = [int(s) for s in shape]
    fan_in, fan_out = (shape[0], shape[1]) if len(shape) == 2 else (shape[-2] * math.prod(shape[:-2]), shape[-1] * math.prod(shape[:-2]))
    return STORE.get(name, shape, _glorot(fan_in, fan_out))


def _cl(x, data_format):
    return x.permute(0, 2, 3, 1) if data_format == "channels_first" else x


def _back(y, data_format):
    return y.permute(0, 3, 1, 2) if data_format == "channels_first" else y


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _layer_scope(base, name):
    return variable_scope(name if name is not None else STORE.unique_layer_name(base))


def _conv2d(inputs, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, data_format="channels_last", name=None, **k):
    assert not k, k
    kh, kw = _pair(kernel_size)
    assert _pair(strides) == (1, 1) and padding.lower() == "same" and kh == kw, "only the forms the reference's hot path uses"
    x = _cl(inputs, data_format)
    cin = int(x.shape[3])
    with _layer_scope("conv2d", name):
        kernel = STORE.get("kernel", (kh, kw, cin, filters), _glorot(kh * kw * cin, kh * kw * filters))
        bias = STORE.get("bias", (filters,), lambda t: None)
    y = T.conv2d_same(x, kernel, bias, False)
    if activation is not None:
        y = activation(y)
    return _back(y, data_format)


def _conv2d_transpose(inputs, filters, kernel_size, strides=(1, 1), padding="valid", activation=None, data_format="channels_last", name=None, **k):
    assert not k, k
    kh, kw = _pair(kernel_size)
    assert _pair(strides) == (2, 2) and padding.lower() == "same" and kh == kw
    x = _cl(inputs, data_format)
    cin = int(x.shape[3])
    with _layer_scope("conv2d_transpose", name):
        kernel = STORE.get("kernel", (kh, kw, filters, cin), _glorot(kh * kw * cin, kh * kw * filters))
        bias = STORE.get("bias", (filters,), lambda t: None)
    y = T.conv2d_transpose_s2(x, kernel, bias, False)
    if activation is not None:
        y = activation(y)
    return _back(y, data_format)


def _max_pooling2d(inputs, pool_size, strides, padding="valid", data_format="channels_last", name=None):
    p, s = _pair(pool_size), _pair(strides)
    assert p[0] == p[1] and s[0] == s[1] and padding.lower() == "same"
    return _back(T.max_pool_same(_cl(inputs, data_format), p[0], s[0]), data_format)


def _average_pooling2d(inputs, pool_size, strides, padding="valid", data_format="channels_last", name=None):
    p, s = _pair(pool_size), _pair(strides)
    assert p[0] == p[1] == s[0] == s[1] and padding.lower() == "same"
    return _back(T.avg_pool_same(_cl(inputs, data_format), p[0]), data_format)


def _unsupported(what):
    def f(*a, **k):
        raise NotImplementedError("%s is outside the hot path (SURVEY section 2: batch norm / dropout are disabled by Architecture.py:505-506)" % what)
    return f


layers = types.SimpleNamespace(conv2d=_conv2d, conv2d_transpose=_conv2d_transpose, max_pooling2d=_max_pooling2d,
                               average_pooling2d=_average_pooling2d, flatten=lambda x, name=None: x.reshape(int(x.shape[0]), -1),
                               batch_normalization=_unsupported("tf.layers.batch_normalization"),
                               dropout=_unsupported("tf.layers.dropout"))


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.as_tensor(x, dtype=DTYPE)


def _nn_conv2d(input, filter=None, strides=None, padding="VALID", data_format="NHWC", **k):
    assert list(strides) == [1, 1, 1, 1] and padding == "VALID"
    x = input if data_format == "NHWC" else input.permute(0, 2, 3, 1)
    w = _t(filter)
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1).contiguous()).permute(0, 2, 3, 1)
    return y if data_format == "NHWC" else y.permute(0, 3, 1, 2)


nn = types.SimpleNamespace(relu=torch.relu, softmax=lambda x, axis=-1: torch.softmax(x, dim=axis), conv2d=_nn_conv2d,
                           embedding_lookup=lambda m, ids: m[torch.as_tensor(list(ids), dtype=torch.long)])


# ----------------------------------------------------------------------------- array ops
def concat(values, axis):
    return torch.cat([_t(v) for v in values], dim=int(axis))


def split(value, num_or_size_splits, axis=0):
    if isinstance(num_or_size_splits, int):
        n = int(value.shape[axis])
        assert n % num_or_size_splits == 0
        return list(torch.split(value, n // num_or_size_splits, dim=axis))
    return list(torch.split(value, [int(s) for s in num_or_size_splits], dim=axis))


def stack(values, axis=0):
    return torch.stack([_t(v) for v in values], dim=axis)


def tile(x, multiples):
    return _t(x).repeat(*[int(m) for m in multiples])


def reshape(x, shape):
    return _t(x).reshape([int(s) for s in shape])


def transpose(x, perm):
    return x.permute(*perm)


def shape(x):
    return [int(s) for s in x.shape]


def slice(x, begin, size):      # noqa: A001  (the TensorFlow name)
    idx = []
    for d, (b, s) in enumerate(zip(begin, size)):
        b, s = int(b), int(s)
        idx.append(builtins_slice(b, None if s == -1 else b + s))
    return x[tuple(idx)]


builtins_slice = __builtins__["slice"] if isinstance(__builtins__, dict) else __builtins__.slice


def pad(x, paddings, mode="CONSTANT"):
    paddings = [[int(a), int(b)] for a, b in paddings]
    assert all(a == b for a, b in paddings)
    nz = [d for d, (a, _) in enumerate(paddings) if a]
    if not nz:
        return x
    p = paddings[nz[0]][0]
    assert all(paddings[d][0] == p for d in nz) and len(nz) == 2 and nz[1] == nz[0] + 1 and mode.lower() == "symmetric"
    # oracle/tf_ops.pad_symmetric mirrors dims 1 and 2 of a 4-D tensor: move the two padded dims there
    lead = nz[0]
    if x.dim() == 4 and lead == 1:
        return T.pad_symmetric(x, p)
    if x.dim() == 4 and lead == 2:      # channels_first batch
        return T.pad_symmetric(x.permute(0, 2, 3, 1), p).permute(0, 3, 1, 2)
    if x.dim() == 3 and lead == 0:
        return T.pad_symmetric(x[None], p)[0]
    if x.dim() == 3 and lead == 1:
        return T.pad_symmetric(x.permute(1, 2, 0)[None], p)[0].permute(2, 0, 1)
    raise NotImplementedError((tuple(x.shape), paddings))


def ones(shape, dtype=None):
    return torch.ones([int(s) for s in shape], dtype=DTYPE)


def zeros(shape, dtype=None):
    return torch.zeros([int(s) for s in shape], dtype=DTYPE)


def map_fn(fn, elems):
    return torch.stack([fn(e) for e in elems], dim=0)


def cond(pred, true_fn, false_fn):
    return true_fn() if bool(pred) else false_fn()


def where(c, a, b):
    return torch.where(c, _t(a), _t(b))


# ----------------------------------------------------------------------------- math
def _bin(f):
    return lambda a, b, name=None: f(_t(a), _t(b))


subtract, add, multiply, divide = _bin(torch.sub), _bin(torch.add), _bin(torch.mul), _bin(torch.div)
minimum, maximum, less, greater = _bin(torch.minimum), _bin(torch.maximum), _bin(torch.lt), _bin(torch.gt)
squared_difference = _bin(lambda a, b: (a - b) ** 2)
scalar_mul = _bin(torch.mul)
abs, sign, square, sqrt, log, sigmoid = torch.abs, torch.sign, torch.square, torch.sqrt, torch.log, torch.sigmoid      # noqa: A001
log1p, expm1 = torch.log1p, torch.expm1


def add_n(values):
    out = _t(values[0])
    for v in values[1:]:
        out = out + _t(v)
    return out


def reduce_sum(x, axis=None, keepdims=False):
    x = _t(x)
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False):
    x = _t(x)
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=keepdims)


# ----------------------------------------------------------------------------- image / estimator / bookkeeping
def _resize_images(x, size, method=None, **k):
    assert method == "NEAREST_NEIGHBOR" and int(size[0]) == 2 * x.shape[1] and int(size[1]) == 2 * x.shape[2]
    return T.resize_nearest_x2(x)


image = types.SimpleNamespace(resize_images=_resize_images, ResizeMethod=types.SimpleNamespace(NEAREST_NEIGHBOR="NEAREST_NEIGHBOR"),
                              ssim_multiscale=_unsupported("tf.image.ssim_multiscale"))
estimator = types.SimpleNamespace(ModeKeys=types.SimpleNamespace(TRAIN="train", EVAL="eval", PREDICT="infer"),
                                  EstimatorSpec=lambda **k: types.SimpleNamespace(**k))
summary = types.SimpleNamespace(scalar=lambda *a, **k: None, histogram=lambda *a, **k: None, image=lambda *a, **k: None)
metrics = types.SimpleNamespace(mean=lambda x, *a, **k: x)


class _Adam:
    """Records that Training.model_fn asked for tf.train.AdamOptimizer(lr).minimize(loss, global_step); the update itself is
    oracle/tf_ops.adam_step, tested against hand-derived values in tests/test_oracle_ops.py."""
    calls = []

    def __init__(self, learning_rate):
        self.learning_rate = learning_rate

    def minimize(self, loss, global_step=None):
        _Adam.calls.append((self.learning_rate, loss))
        return "train_op"


train = types.SimpleNamespace(AdamOptimizer=_Adam, get_or_create_global_step=lambda: "global_step")


def install():
    """Registers this module as `tensorflow` and returns it."""
    mod = sys.modules[__name__]
    sys.modules["tensorflow"] = mod
    return mod
