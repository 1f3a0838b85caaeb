"""ORACLE (test infrastructure) -- CPU restatement of the reference's model graph.

Follows, literally and per tuple (no batching tricks), the reference's
Architecture.predict (TensorFlow/Architecture.py:537-617) and everything it calls:
FeaturePrediction (:82-165), SourceEncoder.prepare_neural_network_input
(SourceEncoder.py:29-79), FeatureFlags.feature_flags (FeatureFlags.py:50-69),
UNet.predict (UNet.py:61-99), Tiramisu.predict (Tiramisu.py:67-111),
AdjustNumberOfChannels (Architecture.py:230-244), KernelPredictor (:247-289),
MultiScalePredictor (:292-325) + MultiScalePrediction.compose_scales
(MultiScalePrediction.py:36-93).  Op semantics come from oracle/tf_ops.py.

PARITY UNPINNED against live TensorFlow (not installable here; the reference ships no
golden vectors) -- see tf_ops.py header and DESIGN.md.

Trainable variables live in a VarStore that reproduces tf.variable_scope(reuse=...)
+ tf.layers auto-naming: creation order == first-use order.

`storage="bf16" | "f16"` (default None = the reference's arithmetic) makes the restatement STORAGE-EMULATING: the same float64 graph, but
every tensor the MI355X half-precision path keeps in HBM in its 2-byte storage type is rounded to that type at the point where it is
stored -- the network input, the MFMA weight images, every conv / transposed-conv output, the partial sum of a conv over a skip concat
(engine.Graph.conv split_at), the compose net's packed input -- and every activation GRADIENT is rounded where the reverse program stores
it (after the ReLU-backward mask).  Nothing else changes (fp32/f64 accumulation, fp32 losses, softmax, blends), so the half-precision
kernels can be gated against this oracle at summation-order tolerance instead of at the storage type's own error.
"""

from collections import OrderedDict

import torch

from . import tf_ops as T

_STORAGE = {None: None, "f32": None, "bf16": torch.bfloat16, "f16": torch.float16}


class _RoundForward(torch.autograd.Function):
    """y = x rounded to the storage type; the gradient passes unchanged (the consumer's data-gradient launch reads the STORED value)."""

    @staticmethod
    def forward(ctx, x, st):
        return x.to(st).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBackward(torch.autograd.Function):
    """y = x; the gradient is rounded to the storage type (a stored activation gradient)."""

    @staticmethod
    def forward(ctx, x, st):
        ctx.st = st
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.st).to(g.dtype), None

# The pure-contract helpers (names / keys) are shared with the product package on purpose:
# they are pinned separately against the reference's own modules (tests/golden/naming_golden.json).
from deepdenoiser_amd.naming import Naming
from deepdenoiser_amd.render_passes import RenderPasses


class _RoundBackwardMasked(torch.autograd.Function):
    """y = x (a ReLU output with several consumers); this consumer's gradient contribution is masked by x > 0 and rounded to the storage
    type before it is added to the others': every data-gradient launch of the half-precision path rounds its own result, then adds the
    gradient already stored and rounds the sum (csrc/dd_conv_bwd.hip, dd_convt.hip, dd_head.hip epilogues)."""

    @staticmethod
    def forward(ctx, x, st):
        ctx.st = st
        ctx.save_for_backward(x)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return (g * (x > 0)).to(ctx.st).to(g.dtype), None


class VarStore:
    """Ordered variable store emulating TF variable scopes with reuse + per-scope layer counters."""

    def __init__(self, dtype=torch.float64, seed=2, storage=None):
        self.dtype = dtype
        self.storage = _STORAGE[storage]
        self.vars = OrderedDict()
        self.gen = torch.Generator().manual_seed(seed)
        self._counters = {}

    def enter_scope(self, scope):
        # re-entering a scope restarts tf.layers' name counters (App. A.10)
        self._counters[scope] = {}

    def _layer_name(self, scope, kind):
        c = self._counters[scope]
        n = c.get(kind, 0)
        c[kind] = n + 1
        return "%s/%s" % (scope, kind if n == 0 else "%s_%d" % (kind, n))

    def get(self, name, shape, fan_in=None, fan_out=None):
        if name not in self.vars:
            t = torch.zeros(shape, dtype=self.dtype)
            if fan_in is not None:
                T.glorot_uniform_(t, fan_in, fan_out, self.gen)
            t.requires_grad_(True)
            self.vars[name] = t
        assert tuple(self.vars[name].shape) == tuple(shape), (name, self.vars[name].shape, shape)
        return self.vars[name]

    # -- storage emulation (no-ops with storage=None)
    def q(self, x):
        """A stored activation: value rounded to the storage type."""
        return x if self.storage is None else _RoundForward.apply(x, self.storage)

    def qgrad(self, x):
        """The point where the reverse program stores this tensor's gradient."""
        return x if self.storage is None else _RoundBackward.apply(x, self.storage)

    def branch(self, x):
        """One consumer's view of a ReLU output that has several consumers (U-Net skip tensors, backbone outputs feeding both the next
        transposed conv and a kernel-prediction head)."""
        return x if self.storage is None else _RoundBackwardMasked.apply(x, self.storage)

    def _stored(self, pre, relu):
        pre = self.qgrad(pre)                  # stored gradients are PRE-activation gradients: masked by the ReLU, then rounded
        return self.q(torch.relu(pre) if relu else pre)

    def conv2d(self, scope, x, filters, k, relu, res=None, split_at=None, fp32_out=False):
        """tf.layers.conv2d(SAME) [+ res] [+ ReLU].  res / split_at only matter under storage emulation: the half-precision path rounds
        conv + residual ONCE, and runs a 3x3 conv over a > 128-channel skip concat as conv(first part) -> stored partial sum ->
        conv(second part) + partial sum (engine.Graph.conv).
        fp32_out (storage emulation only): the layer-wise kernel-prediction head's last 1x1 layer -- its consumer computes the logits itself, in
        fp32, from the stored input and the fp32 MASTER weights (dd_kpcn_hidden_*), so the value is neither rounded nor multiplied with rounded
        weights; the layer's gradient launches still read the stored logit gradient and the rounded weights."""
        name = self._layer_name(scope, "conv2d")
        cin = x.shape[3]
        kernel = self.get(name + "/kernel", (k, k, cin, filters), fan_in=k * k * cin, fan_out=k * k * filters)
        bias = self.get(name + "/bias", (filters,))
        if self.storage is None:
            y = T.conv2d_same(x, kernel, bias, False)
            if res is not None:
                y = y + res
            return torch.relu(y) if relu else y
        kq = self.q(kernel)
        if fp32_out:
            assert res is None and not relu
            pre = T.conv2d_same(x, kq, bias, False) + T.conv2d_same(x, kernel - kq, None, False).detach()      # value: x * W; d/dx: through Wq
            return self.qgrad(pre)
        if (split_at is not None and k == 3 and res is None and cin > 128 and 0 < split_at < cin and split_at % 8 == 0
                and max(split_at, cin - split_at) <= 128):
            part = self.q(T.conv2d_same(x[..., :split_at], kq[:, :, :split_at], bias, False))
            pre = T.conv2d_same(x[..., split_at:], kq[:, :, split_at:], None, False) + part
        else:
            pre = T.conv2d_same(x, kq, bias, False)
            if res is not None:
                pre = pre + res
        return self._stored(pre, relu)

    def conv2d_transpose(self, scope, x, filters, k, relu):
        name = self._layer_name(scope, "conv2d_transpose")
        cin = x.shape[3]
        # TF computes Glorot fans from the variable shape [k,k,out,in]: fan_in=k*k*out, fan_out=k*k*in
        kernel = self.get(name + "/kernel", (k, k, filters, cin), fan_in=k * k * filters, fan_out=k * k * cin)
        bias = self.get(name + "/bias", (filters,))
        if self.storage is None:
            return T.conv2d_transpose_s2(x, kernel, bias, relu)
        kq = self.q(kernel)
        if k == 2 and filters > 64:      # dd_convt2x2_bwd: one launch per 64 output channels, the later ones accumulating into dx
            pre = torch.cat([T.conv2d_transpose_s2(self.branch(x), kq[:, :, c0:c0 + 64], bias[c0:c0 + 64], False) for c0 in range(0, filters, 64)], dim=3)
        else:
            pre = T.conv2d_transpose_s2(x, kq, bias, False)
        return self._stored(pre, relu)


# ----------------------------------------------------------------------------- backbones
def unet_predict(vs, scope, x, filters, convs_per_block, multiscale):
    """UNet.py:61-99 (BN/dropout are hard-disabled, Architecture.py:506)."""
    steps = len(filters) - 1
    results, skips = [], []

    def block(x, f, split_at=None):
        for c in range(convs_per_block):
            x = vs.conv2d(scope, x, f, 3, relu=True, split_at=split_at if c == 0 else None)           # UNet.py:25-36
        return x

    for i in range(steps):
        x = block(x, filters[i])
        skips.append(x)
        x = T.max_pool_same(vs.branch(x), 3, 2)                 # UNet.py:38-52
    for i in range(steps):
        index = steps - i
        x = block(x, filters[index], split_at=filters[index] if i > 0 else None)     # [skip | upsampled]: filters[index] channels each
        if multiscale:
            results.append(x)
        x = vs.conv2d_transpose(scope, vs.branch(x) if multiscale else x, filters[index - 1], 2, relu=True)   # UNet.py:54-59
        x = torch.cat([vs.branch(skips[index - 1]), x], dim=3)  # UNet.py:91-92
    x = block(x, filters[0], split_at=filters[0])
    results.append(x)
    return results


def tiramisu_predict(vs, scope, x, filters, convs_per_block, multiscale):
    """Tiramisu.py:67-111; preprocessing filters = filters[0] (Architecture.py:214-215)."""
    steps = len(filters) - 1
    results, skips = [], []

    def block(x, f):
        for _ in range(convs_per_block):
            layer = vs.conv2d(scope, torch.relu(x), f, 3, relu=False)     # Tiramisu.py:26-41
            x = torch.cat([x, layer], dim=3)
        return x

    x = vs.conv2d(scope, x, filters[0], 3, relu=True)                      # :76-79
    for i in range(steps):
        x = block(x, filters[i])
        skips.append(x)
        x = vs.conv2d(scope, torch.relu(x), x.shape[3], 1, relu=False)     # :43-58
        x = T.max_pool_same(x, 2, 2)
    for i in range(steps):
        index = steps - i
        x = block(x, filters[index])
        if multiscale:
            results.append(x)
        x = vs.conv2d_transpose(scope, x, filters[index - 1], 3, relu=True)   # :60-65
        x = torch.cat([skips[index - 1], x], dim=3)
    x = block(x, filters[0])
    results.append(x)
    return results


# ----------------------------------------------------------------------------- multiscale compose
def compose_scales(vs, scope, small, fine):
    """MultiScalePrediction.compose_scales (:36-54) with its weight net (:57-93)."""
    small = T.resize_nearest_x2(small)
    x = vs.q(vs.qgrad(torch.cat([small, fine], dim=3)))      # (storage emulation: the net reads the packed 6-channel input in the storage type)
    x = vs.conv2d(scope, x, 24, 1, relu=True)
    for _ in range(2):
        r = vs.conv2d(scope, torch.relu(x), 24, 3, relu=False)
        x = vs.conv2d(scope, torch.relu(r), 24, 3, relu=False, res=x)      # x + 1.0 * r (MultiScalePrediction.py:81-93)
    x = vs.conv2d(scope, x, 1, 1, relu=True)
    wts = torch.sigmoid(x)
    low = T.resize_nearest_x2(T.avg_pool_same(fine, 2))
    return fine - wts * low + wts * small


# ----------------------------------------------------------------------------- JSON -> structure
class _Std:
    def __init__(self, j):
        self.use_log1p, self.mean, self.variance = j["use_log1p"], float(j["mean"]), float(j["variance"])


class _Var:
    def __init__(self, j):
        self.use_variance = j["use_variance"]
        self.mode = j["variance_mode"]
        self.relative = j["relative_variance"]
        self.before = j["compute_before_standardization"]
        self.compress = j["compress_to_one_channel"]


class _Feature:
    def __init__(self, ftype, load_data, is_target, std, invert, var, channels, name):
        self.ftype, self.load_data, self.is_target = ftype, load_data, is_target
        self.std, self.invert, self.var, self.channels, self.name = std, invert, var, channels, name
        self.predictions = []


class OracleArchitecture:
    """Restatement of Architecture.__init__ (:343-535) + predict (:537-617)."""

    def __init__(self, parsed_json, dtype=torch.float64, seed=2, storage=None):
        self.dtype = dtype
        self.vs = VarStore(dtype, seed, storage)
        self.sources_per_target = parsed_json["number_of_sources_per_target"]
        arch = parsed_json["architecture"]
        self.tuple_type = arch["source_encoder"]["feature_prediction_tuple_type"]
        self.flag_mode = arch["source_encoder"]["feature_flag_mode"]
        core = arch["core_architecture"]
        self.core_name = core["name"]
        self.filters = list(core["number_of_filters_for_convolution_blocks"])
        self.convs_per_block = core["number_of_convolutions_per_block"]
        kp = arch["kernel_prediction"]
        self.use_kp, self.kernel_size = kp["use_kernel_prediction"], kp["kernel_size"]
        self.kp_standardized_source = kp["use_standardized_source_for_kernel_prediction"]
        ms = arch["multiscale_prediction"]
        self.use_multiscale = ms["use_multiscale_predictions"]
        self.invert_after_multiscale = ms["invert_standardization_after_multiscale_predictions"]

        # auxiliaries, sorted by name (Architecture.py:369-392)
        self.auxiliary = []
        for name in sorted(parsed_json["auxiliary_features"].keys()):
            j = parsed_json["auxiliary_features"][name]
            self.auxiliary.append(_Feature("AUXILIARY", True, False, _Std(j["standardization"]), False,
                                           _Var(j["feature_variance"]), j["number_of_channels"], name))
        handling = parsed_json["combined_features_handling"]
        self.features, self.tuples = [], []
        for cname in sorted(parsed_json["combined_features"].keys()):        # :420
            members = []
            for ftype in ("Color", "Direct", "Indirect"):
                fname = parsed_json["combined_features"][cname][ftype]
                h = handling[ftype]
                channels = RenderPasses.number_of_channels(fname)
                load = True
                if fname is None or fname == "":
                    fname, load = cname + " " + ftype, False
                f = None
                if load or self.tuple_type == "COMBINED":                   # :443
                    f = _Feature(ftype.upper(), load, True, _Std(h["standardization"]), h["invert_standardization"],
                                 _Var(h["feature_variance"]), channels, fname)
                    self.features.append(f)
                members.append(f)
            if self.tuple_type == "COMBINED":
                self.tuples.append((cname, members))
        if self.tuple_type == "SINGLE":
            self.tuples = [(f.name, [f]) for f in self.features]               # :467-473
        self.flag_names = sorted(n for n, _ in self.tuples)                   # FeatureFlags.py:22
        tuple_size = 1 if self.tuple_type == "SINGLE" else 3
        self.post_channels = (self.sources_per_target * tuple_size * self.kernel_size ** 2) if self.use_kp \
            else tuple_size * 3                                                # :515-522

    # -- per-feature pre-processing (FeaturePrediction.standardize, :114-132)
    def _prepare(self, f, features):
        f.source, f.variance, f.preserved = [], [], []
        for i in range(self.sources_per_target):
            s = features[Naming.source_feature_name(f.name, index=i)].to(self.dtype)
            f.preserved.append(s)
            v = None
            if f.var.use_variance and f.var.before:
                v = T.variance(s, f.var.mode, f.var.relative, f.var.compress)
            s = T.standardize(s, f.std.use_log1p, f.std.mean, f.std.variance)
            if f.var.use_variance and not f.var.before:
                v = T.variance(s, f.var.mode, f.var.relative, f.var.compress)
            f.source.append(s)
            f.variance.append(v)

    def _network_input(self, tname, members, features):
        """SourceEncoder.prepare_neural_network_input (SourceEncoder.py:29-79)."""
        parts = []
        for i in range(self.sources_per_target):
            for f in list(members) + self.auxiliary:
                s = f.source[i]
                if s.shape[3] != 3:
                    assert s.shape[3] == 1
                    s = torch.cat([s, s, s], dim=3)
                parts.append(s)
                if f.var.use_variance:
                    parts.append(f.variance[i])
        x = torch.cat(parts, dim=3)
        if self.flag_mode == "ONE_HOT_ENCODING":
            x = torch.cat([x, features[Naming.feature_flags_name(tname)].to(self.dtype)], dim=3)
        elif self.flag_mode == "EMBEDDING":
            v = len(self.flag_names)
            d = v // 2
            matrix = self.vs.get("embedding/feature_flags_embedding_matrix", (v, d), fan_in=v, fan_out=d)
            row = matrix[self.flag_names.index(tname)]
            x = torch.cat([x, row.reshape(1, 1, 1, d).expand(x.shape[0], x.shape[1], x.shape[2], d)], dim=3)
        return x

    def predict(self, features, return_internals=False):
        vs = self.vs
        for f in self.features + self.auxiliary:
            f.predictions = []
            self._prepare(f, features)
        internals = {}
        for tname, members in self.tuples:
            x = vs.q(vs.qgrad(self._network_input(tname, members, features)))
            internals.setdefault("network_input", []).append(x)
            scope = "reused_core_architecture"
            vs.enter_scope(scope)
            if self.core_name == "U-Net":
                outs = unet_predict(vs, scope, x, self.filters, self.convs_per_block, self.use_multiscale)
            else:
                assert self.core_name == "Tiramisu"
                outs = tiramisu_predict(vs, scope, x, self.filters, self.convs_per_block, self.use_multiscale)
            internals.setdefault("core_outputs", []).append(outs)
            post = []
            for i_out, o in enumerate(outs):                          # coarsest first (:573-575)
                if self.core_name == "U-Net" and i_out < len(outs) - 1:
                    o = vs.branch(o)                                  # (storage emulation: this output also feeds the next transposed conv)
                # (storage emulation: program.py runs the head layer by layer unless its fused kernel applies -- one tuple member, 3x3 / 5x5 kernels,
                #  backbone outputs of <= 128 channels in multiples of 8 -- and then the kernel-prediction launch computes the logits in fp32)
                fused = self.use_kp and len(members) == 1 and self.kernel_size in (3, 5) and all(t.shape[3] % 8 == 0 and t.shape[3] <= 128 for t in outs)
                o = vs.conv2d(scope, o, self.post_channels, 1, relu=True)
                o = vs.conv2d(scope, o, self.post_channels, 1, relu=False, fp32_out=vs.storage is not None and self.use_kp and not fused)
                post.append(o)
            if self.use_multiscale:
                post = list(reversed(post))                           # :577-579
            internals.setdefault("post", []).append(post)
            for s, o in enumerate(post):
                chunks = torch.chunk(o, len(members), dim=3)          # :582
                for f, ch in zip(members, chunks):
                    while len(f.predictions) <= s:
                        f.predictions.append(None)
                    f.predictions[s] = ch
            for f in members:                                         # KernelPredictor.predict :260-289
                if self.use_kp:
                    src = f.source[0] if self.kp_standardized_source else f.preserved[0]
                    if src.shape[3] != 3:
                        src = torch.cat([src, src, src], dim=3)
                    for s in range(len(f.predictions)):
                        ssrc = T.avg_pool_same(src, 2 ** s) if s > 0 else src
                        f.predictions[s] = T.kernel_prediction(ssrc, f.predictions[s], self.kernel_size)
            for f in members:                                         # MultiScalePredictor.predict :302-325
                if not self.invert_after_multiscale and f.invert:
                    self._invert(f)
                if self.use_multiscale:
                    for s in range(len(f.predictions) - 1, 0, -1):
                        cscope = "reused_compose_scales"
                        vs.enter_scope(cscope)
                        f.predictions[s - 1] = compose_scales(vs, cscope, f.predictions[s], f.predictions[s - 1])
                if self.invert_after_multiscale and f.invert:
                    self._invert(f)
        target = next(f for f in self.features if f.is_target)
        dicts = []
        for s in range(len(target.predictions)):
            d = {}
            for f in self.features:
                p = f.predictions[s]
                if not f.load_data:                                   # :151-157: echo the source
                    p = f.source[0][:, :p.shape[1], :p.shape[2], :p.shape[3]]
                if p.shape[3] != f.channels:
                    assert f.channels == 1
                    p = p[:, :, :, :1]
                d[Naming.feature_prediction_name(f.name)] = p
            dicts.append(d)
        if return_internals:
            return dicts, internals
        return dicts

    @staticmethod
    def _invert(f):
        f.predictions = [T.invert_standardization(p, f.std.use_log1p, f.std.mean, f.std.variance) for p in f.predictions]

    def parameters(self):
        return list(self.vs.vars.values())
