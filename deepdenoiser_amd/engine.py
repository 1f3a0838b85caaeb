"""Static launch-graph executor over the libdd_hip C-ABI.

The reference delegates execution to TensorFlow's graph runtime + autodiff (tf.estimator, reference
TensorFlow/Training.py:1214-1232, :701-702).  The MI355X-native replacement is a *static program*: the model
code records forward kernel launches once (shapes are static per (batch, tile) configuration); every op pushes a
closure that emits its backward launches, so `build_backward()` produces the reverse program.  Programs are flat
lists of bound C calls on one HIP stream -- cheap to replay and capturable into a hipGraph.

Conventions (see include/dd_hip.h): NHWC; tensor = (pointer, ld); activations are stored in the graph dtype
('f32' parity path / 'bf16' training throughput path / 'f16' inference path) with channel counts padded to a multiple of 8 (pad channels are
always zero); ReLU backward is fused into the PRODUCER of a gradient (mask epilogues), so stored gradients are
pre-activation gradients; multi-consumer tensors accumulate (first writer overwrites, later writers add).
PyTorch provides device memory and streams only.
"""

import ctypes as C
import math
import os

import torch

from . import _lib as L

# storage types of activations and packed weights: f32 = parity path (exact-f32 MFMA), bf16 = training throughput path, f16 = inference
# path of BASELINE cfg-5 (fp16 MFMA at the bf16 rate, 3 more mantissa bits; range +-65504, so training in it needs the loss scale
# of Program(loss_scale=...)); accumulation, losses, kernel-prediction softmax and blends are fp32 in every mode
_TORCH_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
_CODE = {"f32": L.DD_F32, "bf16": L.DD_BF16, "f16": L.DD_F16}
_ESZ = {"f32": 4, "bf16": 2, "f16": 2}


def round_up(v, m):
    return (v + m - 1) // m * m


class DT:
    """Device tensor handle: a channel-range view [ch0, ch0+C) of an NHWC torch buffer [B,H,W,ld]."""

    def __init__(self, buf, B, H, W, C, Cp, ch0, dtype, relu=False, requires_grad=False, gstate=None):
        self.buf, self.B, self.H, self.W, self.C, self.Cp, self.ch0, self.dtype = buf, B, H, W, C, Cp, ch0, dtype
        self.ld = buf.shape[-1]
        self.relu = relu                      # produced by a ReLU epilogue => incoming gradients get masked by (self > 0)
        self.requires_grad = requires_grad
        self.self_mask = False                # ReLU output whose consumers cannot mask (dense-concat ranges): the producer masks its own gradient in place
        # "zero_list": the owning Graph's list of gradient buffers that are zeroed before every backward pass
        self.gstate = gstate if gstate is not None else {"buf": None, "written": False, "zero_init": False, "zero_list": None}

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.ch0 * _ESZ[self.dtype]

    @property
    def npix(self):
        return self.B * self.H * self.W

    def view(self, ch0, C, Cp=None, relu=None):
        return DT(self.buf, self.B, self.H, self.W, C, C if Cp is None else Cp, self.ch0 + ch0, self.dtype,
                  self.relu if relu is None else relu, self.requires_grad, self.gstate)

    def grad(self):
        """Gradient view with the same layout (allocated on first use)."""
        if self.gstate["buf"] is None:
            # zeros: pad channels of gradients must be zero; fully-written tensors overwrite anyway
            self.gstate["buf"] = torch.zeros_like(self.buf)
            if self.gstate["zero_init"]:
                self.gstate["zero_list"].append(self.gstate["buf"])
        return DT(self.gstate["buf"], self.B, self.H, self.W, self.C, self.Cp, self.ch0, self.dtype)

    @property
    def grad_written(self):
        # zero_init storages (several partial-range writers) are zeroed every step and always accumulated into
        return self.gstate["written"] or self.gstate["zero_init"]

    def mark_grad_written(self):
        self.gstate["written"] = True

    def torch(self):
        """Logical [B,H,W,C] view (for tests / outputs)."""
        return self.buf[..., self.ch0:self.ch0 + self.C]


class Param:
    def __init__(self, name, shape, fan_in, fan_out, offset):
        self.name, self.shape, self.fan_in, self.fan_out, self.offset = name, tuple(shape), fan_in, fan_out, offset
        self.size = int(math.prod(shape))


class ParamStore:
    """Flat fp32 arenas (values / grads / Adam m / Adam v) in TF variable-creation order.
    One contiguous gradient arena = zero-copy buckets for the data-parallel all-reduce and one Adam launch."""

    def __init__(self):
        self.params, self.by_name, self.total = [], {}, 0
        self.values = self.grads = self.m = self.v = None

    def get(self, name, shape, fan_in=None, fan_out=None):
        if name in self.by_name:
            p = self.by_name[name]
            assert p.shape == tuple(shape), (name, p.shape, shape)
            return p
        assert self.values is None, "parameter store already finalized"
        p = Param(name, shape, fan_in, fan_out, self.total)
        self.total += round_up(p.size, 4)       # keep every parameter 16-byte aligned
        self.params.append(p)
        self.by_name[name] = p
        return p

    def finalize(self, device, seed=2):
        if self.values is not None:
            return
        n = max(self.total, 4)
        self.values = torch.zeros(n, dtype=torch.float32, device=device)
        self.grads = torch.zeros(n, dtype=torch.float32, device=device)
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        gen = torch.Generator().manual_seed(seed)
        host = torch.zeros(n, dtype=torch.float32)
        for p in self.params:   # Glorot-uniform kernels, zero biases (TF defaults)
            if p.fan_in is not None:
                limit = math.sqrt(6.0 / (p.fan_in + p.fan_out))
                host[p.offset:p.offset + p.size] = (torch.rand(p.size, generator=gen) * 2 - 1) * limit
        self.values.copy_(host)

    def state_key(self):
        """Changes whenever the values may have: torch counts the in-place writes it performs on the arena and its views (load_list, checkpoint
        restore, tests), launches that write through the raw pointer (dd_adam_step) count themselves in `raw_writes`.  Lets an inference loop
        skip re-packing unchanged weights (prediction.Predictor)."""
        return (self.values._version, getattr(self, "raw_writes", 0)) if self.values is not None else None

    def value(self, p):
        return self.values[p.offset:p.offset + p.size].view(p.shape)

    def grad(self, p):
        return self.grads[p.offset:p.offset + p.size].view(p.shape)

    def value_ptr(self, p):
        return self.values.data_ptr() + 4 * p.offset

    def grad_ptr(self, p):
        return self.grads.data_ptr() + 4 * p.offset

    def load_list(self, tensors):
        """Copy a list of tensors (creation order) into the arena -- used to share weights with the oracle."""
        if self.values is None:
            raise RuntimeError("parameters are created when the first program is built: call Architecture.program(...) / "
                               "Predictor.prepare(H, W) before load_list()")
        assert len(tensors) == len(self.params), (len(tensors), len(self.params))
        for p, t in zip(self.params, tensors):
            assert tuple(t.shape) == p.shape, (p.name, tuple(t.shape), p.shape)
            self.value(p).copy_(t.detach().to(torch.float32))


class ConvLayer:
    """One conv-like layer: TF-layout fp32 master weights + MFMA-packed copies (forward / data-gradient operand)."""

    def __init__(self, graph, name, k, cin, cout, kind="conv"):
        self.g, self.name, self.k, self.cin, self.cout, self.kind = graph, name, k, cin, cout, kind
        ps = graph.params
        if kind == "conv":
            self.kernel = ps.get(name + "/kernel", (k, k, cin, cout), k * k * cin, k * k * cout)
        else:   # transpose conv: TF variable [k,k,C_out,C_in]; Glorot fans follow the variable shape
            self.kernel = ps.get(name + "/kernel", (k, k, cout, cin), k * k * cout, k * k * cin)
        self.bias = ps.get(name + "/bias", (cout,))
        self._packed = {}

    def packed_k_range(self, k0, kn):
        """Forward operand of input channels [k0, k0 + kn) only (conv layers): the K-split of a conv over a channel concat."""
        key = ("fwd", k0, kn)
        if key in self._packed:
            return self._packed[key]
        assert self.kind == "conv" and 0 <= k0 and k0 + kn <= self.cin
        g, cout = self.g, self.cout
        chunk = 64 // _ESZ[g.dtype]
        taps = self.k * self.k
        n_pad, k_pad = round_up(cout, 16), round_up(round_up(kn, 8), chunk)
        buf = torch.zeros(taps * n_pad * k_pad, dtype=_TORCH_DT[g.dtype], device=g.device)
        g.register_pack(self.kernel, buf, taps, cout, kn, n_pad, k_pad, self.cin * cout, 1, cout, 0, src_offset=k0 * cout)
        self._packed[key] = (buf, taps, n_pad, k_pad)
        return self._packed[key]

    def packed(self, role):
        """role 'fwd' | 'dgrad' -> (buffer, taps, n_pad, k_pad); registers the pack launch on first use."""
        if role in self._packed:
            return self._packed[role]
        g, cin, cout, kind = self.g, self.cin, self.cout, self.kind
        chunk = 64 // _ESZ[g.dtype]
        if kind == "conv":
            taps = self.k * self.k
            if role == "fwd":
                n, k, st, sn, sk, flip = cout, cin, cin * cout, 1, cout, 0
            else:
                n, k, st, sn, sk, flip = cin, cout, cin * cout, cout, 1, 1
        elif kind == "convT2":
            if role == "fwd":
                taps, n, k, st, sn, sk, flip = 1, 4 * cout, cin, 0, cin, 1, 0
            else:
                taps, n, k, st, sn, sk, flip = 4, cin, cout, cout * cin, 1, cin, 0
        else:   # convT3 == 3x3 SAME conv of the zero-stuffed input with the flipped kernel
            taps = 9
            if role == "fwd":
                n, k, st, sn, sk, flip = cout, cin, cout * cin, cin, 1, 1
            else:
                n, k, st, sn, sk, flip = cin, cout, cout * cin, 1, cin, 0
        n_pad, k_pad = round_up(n, 16), round_up(round_up(k, 8), chunk)
        buf = torch.zeros(taps * n_pad * k_pad, dtype=_TORCH_DT[g.dtype], device=g.device)
        g.register_pack(self.kernel, buf, taps, n, k, n_pad, k_pad, st, sn, sk, flip)
        self._packed[role] = (buf, taps, n_pad, k_pad)
        return self._packed[role]


def convt3_s2d_blocks():
    """The nine (a, b) taps of the 3x3 / stride-2 transposed conv as seen from the INPUT grid once the output gradient is rearranged by output
    parity, s[i][j][plane*cp + co] = dy[2i + py][2j + px][co] with plane = py*2 + px (dd_space_to_depth2):
        dx[i][j][ci]      = sum_(a,b) sum_co s[i + di][j + dj][plane*cp + co] * K[a][b][co][ci]
        dK[a][b][co][ci]  = sum_(i,j)        s[i + di][j + dj][plane*cp + co] * x[i][j][ci]
    Returns [(a, b, di, dj, plane, image_tap)]: output row o = 2i + a is row i of parity a & 1 for a < 2 and row i + 1 of parity
    0 for a = 2; image_tap = (1 + di)*3 + (1 + dj) is where dd_conv3x3_ks mode 6 (taps at offsets 0 / +1) expects the block."""
    out = []
    for a in range(3):
        for b in range(3):
            di, dj = int(a == 2), int(b == 2)
            out.append((a, b, di, dj, (a & 1) * 2 + (b & 1), (1 + di) * 3 + (1 + dj)))
    return out


class Graph:
    def __init__(self, device, dtype="f32", params=None):
        assert dtype in _CODE
        self.lib = L.load()
        self.device, self.dtype, self.code = torch.device(device), dtype, _CODE[dtype]
        self.params = params if params is not None else ParamStore()
        self.pack_ops, self.fwd_ops, self.bwd_ops, self._tape = [], [], [], []
        self.layers = {}
        self.keep = []     # keeps auxiliary device buffers alive
        self.zero_init_buffers = []     # gradient buffers with several partial-range writers: zeroed before every backward pass
        self.conv_records, self.wgrad_records, self.bwd_records, self.convt_records = [], [], [], []
        self._pack_records = []

    # ------------------------------------------------------------------ weight packing: every layer in ONE launch
    def register_pack(self, kernel, buf, taps, n, k, n_pad, k_pad, st, sn, sk, flip, src_offset=0, dst_offset=0, dst_ld=0, dst_tap_stride=0):
        """dst_offset / dst_ld / dst_tap_stride (elements): this record fills the [n_pad][k_pad] corner at `dst_offset` of every tap of a wider
        image [taps][n_pad][dst_ld] (stacked reductions: the gather-form data gradient of a dense block)."""
        self._pack_records.append((kernel, buf, taps, n, k, n_pad, k_pad, st, sn, sk, flip, src_offset, dst_offset, dst_ld, dst_tap_stride))
        if not self.pack_ops:
            state = {}

            def pack_all(stream):
                if state.get("n") != len(self._pack_records):   # (re)build the device table when layers were added
                    tab = (L.PackDesc * len(self._pack_records))()
                    for i, (kern, b, tp, nn, kk, npad, kpad, s_t, s_n, s_k, fl, off, doff, dld, dts) in enumerate(self._pack_records):
                        tab[i] = L.PackDesc(self.params.value_ptr(kern) + 4 * off, b.data_ptr() + doff * _ESZ[self.dtype], tp, nn, kk, npad, kpad, fl,
                                            s_t, s_n, s_k, dld, dts)
                    state["dev"] = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
                    state["n"] = len(self._pack_records)
                L.check(self.lib.dd_pack_weights_batched(state["dev"].data_ptr(), state["n"], self.code, stream))
            pack_all.tag = "pack_weights"
            self.pack_ops.append(pack_all)

    # ------------------------------------------------------------------ tensors
    def tensor(self, B, H, W, C, dtype=None, relu=False, requires_grad=True, ld=None, zero=True):
        dtype = dtype or self.dtype
        Cp = round_up(C, 8)
        ld = ld or Cp
        alloc = torch.zeros if zero else torch.empty
        buf = alloc((B, H, W, ld), dtype=_TORCH_DT[dtype], device=self.device)
        t = DT(buf, B, H, W, C, Cp, 0, dtype, relu, requires_grad)
        t.gstate["zero_list"] = self.zero_init_buffers
        return t

    def layer(self, name, k, cin, cout, kind="conv"):
        if name not in self.layers:
            self.layers[name] = ConvLayer(self, name, k, cin, cout, kind)
        lay = self.layers[name]
        assert (lay.k, lay.cin, lay.cout, lay.kind) == (k, cin, cout, kind), name
        return lay

    # ------------------------------------------------------------------ recording helpers
    def fwd(self, fn, tag="pointwise"):
        if not hasattr(fn, "tag"):
            fn.tag = tag
        self.fwd_ops.append(fn)

    def on_backward(self, builder):
        """builder() is called once, in reverse recording order, and appends launches via self.bwd(...)."""
        self._tape.append(builder)

    def bwd(self, fn, tag="pointwise", grad_params=()):
        if not hasattr(fn, "tag"):
            fn.tag = tag
        fn.grad_params = tuple(grad_params)      # parameters whose gradient this launch (partially) produces
        self.bwd_ops.append(fn)

    def build_backward(self):
        for builder in reversed(self._tape):
            builder()
        self._tape = []
        self._merge_weights_only()

    def _merge_weights_only(self, max_pixels=1 << 18, group=4):
        """Weight-gradient-only launches of the fused backward kernel on one SMALL pixel grid (the 128-channel layers of the U-Net's 32 x 32 level:
        B x 1 024 pixels) run side by side as one launch, at the position of the last of them (dd_conv3x3_bwd_multi): their operands -- a layer's
        input and its output gradient -- are final from the original position on and nothing in between reads the weight gradients.  On a large
        grid the flush this saves is a few per cent of the launch and the launches stay apart (DD_WGRAD_MULTI=0: always)."""
        if os.environ.get("DD_WGRAD_MULTI", "1") == "0":
            return
        ops, lib = self.bwd_ops, self.lib
        by_grid = {}
        for i, op in enumerate(ops):
            w = getattr(op, "weights_only", None)
            if w is not None and w[1].B * w[1].H * w[1].W <= max_pixels:
                by_grid.setdefault((w[1].B, w[1].H, w[1].W), []).append(i)
        drop, put = set(), {}
        for idxs in by_grid.values():
            for g0 in range(0, len(idxs), group):
                chunk = idxs[g0:g0 + group]
                if len(chunk) < 2:
                    continue
                members = [ops[i].weights_only for i in chunk]
                records = []
                for gy, x, layer in members:
                    rec = {"flops": 2.0 * x.B * x.H * x.W * 9 * layer.cin * layer.cout, "B": x.B, "H": x.H, "W": x.W, "taps": 9, "m": layer.cin, "n": layer.cout}
                    self.wgrad_records.append(rec)
                    records.append(rec)
                info = dict(records[0], flops=sum(r["flops"] for r in records), merged=len(members))

                def run(stream, members=members, cell=[]):
                    if not cell:      # parameter pointers exist only after ParamStore.finalize()
                        arr = (L.ConvBwdArgs * len(members))()
                        for a, (gy, x, layer) in zip(arr, members):
                            a.dy, a.ld_dy, a.cout = gy.ptr, gy.ld, layer.cout
                            a.x, a.ld_x, a.cin = x.ptr, x.ld, layer.cin
                            a.wd, a.n_pad, a.k_pad, a.dx, a.ld_dx = None, 0, 0, None, 0
                            a.dw, a.db = self.params.grad_ptr(layer.kernel), self.params.grad_ptr(layer.bias)
                            a.B, a.H, a.W, a.use_mask, a.accumulate, a.dtype = x.B, x.H, x.W, 0, 0, self.code
                        cell.append(arr)
                    L.check(lib.dd_conv3x3_bwd_multi(cell[0], len(members), stream))
                run.tag, run.info = "conv_wgrad", info
                run.grad_params = tuple(p for _, _, layer in members for p in (layer.kernel, layer.bias))
                run.keep = [(gy.buf, x.buf) for gy, x, _ in members]
                put[chunk[-1]] = run
                drop.update(chunk[:-1])
        if put:
            self.bwd_ops = [put.get(i, op) for i, op in enumerate(ops) if i not in drop]

    # ------------------------------------------------------------------ conv launches
    def _conv_call(self, x, wp, taps, n_pad, k_pad, bias, nbias, res, mask, y, B, H, W, flags, nk=None):
        if nk is not None:   # algorithmic FLOPs of this launch from the LOGICAL layer shape (roofline accounting, bench.py)
            self.conv_records.append({"flops": 2.0 * B * H * W * taps * nk[0] * nk[1], "B": B, "H": H, "W": W, "taps": taps, "n": nk[0], "k": nk[1],
                                      "extra_reads": (1 if mask is not None else 0) + (1 if res is not None else 0) + (1 if flags & L.ACCUM else 0)})
        a = L.ConvArgs()
        a.x, a.ldx, a.cin = x.ptr, x.ld, x.Cp
        a.wp, a.k_pad, a.n_pad = wp.data_ptr(), k_pad, n_pad
        a.bias, a.nbias = bias, nbias
        a.res, a.ldres = (res.ptr, res.ld) if res is not None else (None, 0)
        a.mask, a.ldmask = (mask.ptr, mask.ld) if mask is not None else (None, 0)
        a.y, a.ldy, a.n = y.ptr, y.ld, y.Cp
        a.B, a.H, a.W, a.taps, a.flags, a.dtype = B, H, W, taps, flags, self.code
        lib = self.lib
        keep = (x.buf, wp, y.buf, res.buf if res is not None else None, mask.buf if mask is not None else None)

        def run(stream, a=a, keep=keep):
            L.check(lib.dd_conv_igemm(C.byref(a), stream))
        run.info = dict(self.conv_records[-1], flags=flags) if nk is not None else None
        return run

    def _wgrad_call(self, p, m, q, n, out_ptr, B, H, W, taps, flags, bias_ptr=None, bias_mode=0):
        self.wgrad_records.append({"flops": 2.0 * B * H * W * taps * m * n, "B": B, "H": H, "W": W, "taps": taps, "m": m, "n": n})
        a = L.WgradArgs()
        a.p, a.ldp, a.m = p.ptr, p.ld, m
        a.q, a.ldq, a.n = q.ptr, q.ld, n
        a.out = out_ptr
        a.bias_out, a.bias_mode = bias_ptr, bias_mode
        a.B, a.H, a.W, a.taps, a.flags, a.dtype, a.ksplit = B, H, W, taps, flags, self.code, 0
        lib = self.lib
        keep = (p.buf, q.buf)

        def run(stream, a=a, keep=keep):
            L.check(lib.dd_conv_wgrad(C.byref(a), stream))
        run.info = self.wgrad_records[-1]
        return run

    def _wgrad_stack_call(self, p, q, layers, m0, width, B, H, W, flags):
        """The weight / bias gradients of every conv of a dense block in ONE launch (dd_conv_wgrad, stacked form): p = the longest input prefix,
        q = the contiguous output gradients of all its convs."""
        ps, nb = self.params, len(layers)
        m, n = m0 + (nb - 1) * width, nb * width
        flops = sum(2.0 * B * H * W * 9 * (m0 + j * width) * width for j in range(nb))
        self.wgrad_records.append({"flops": flops, "B": B, "H": H, "W": W, "taps": 9, "m": m, "n": n, "stacked": nb})
        rec = self.wgrad_records[-1]
        a = L.WgradArgs()
        a.p, a.ldp, a.m = p.ptr, p.ld, m
        a.q, a.ldq, a.n = q.ptr, q.ld, n
        a.out, a.bias_out, a.bias_mode = None, None, 1
        a.B, a.H, a.W, a.taps, a.flags, a.dtype, a.ksplit = B, H, W, 9, flags, self.code, 0
        a.stack_blocks, a.stack_width, a.stack_m0 = nb, width, m0
        lib = self.lib
        keep = (p.buf, q.buf)

        def run(stream, a=a, keep=keep, bound=[]):
            if not bound:      # parameter pointers exist only after ParamStore.finalize()
                for j, lay in enumerate(layers):
                    a.stack_out[j], a.stack_bias[j] = ps.grad_ptr(lay.kernel), ps.grad_ptr(lay.bias)
                bound.append(1)
            L.check(lib.dd_conv_wgrad(C.byref(a), stream))
        run.info, run.tag = rec, "conv_wgrad"
        return run

    def _conv_bwd_call(self, gy, x, layer, wd, n_pad, k_pad, gx, use_mask, accumulate, as_wgrad=False):
        """Data + weight + bias gradients of a 3x3 layer in one launch (csrc/dd_conv_bwd.hip); gx None: weight / bias gradients only
        (as_wgrad: accounted with the weight-gradient launches)."""
        B, H, W = x.B, x.H, x.W
        if as_wgrad:
            self.wgrad_records.append({"flops": 2.0 * B * H * W * 9 * layer.cin * layer.cout, "B": B, "H": H, "W": W, "taps": 9, "m": layer.cin, "n": layer.cout})
        (self.bwd_records if not as_wgrad else []).append({"flops": (4.0 if gx is not None else 2.0) * B * H * W * 9 * layer.cin * layer.cout, "B": B, "H": H, "W": W, "taps": 9,
                                 "n": layer.cin, "k": layer.cout, "accumulate": bool(accumulate), "weights_only": gx is None})
        ps = self.params
        a = L.ConvBwdArgs()
        a.dy, a.ld_dy, a.cout = gy.ptr, gy.ld, layer.cout
        a.x, a.ld_x, a.cin = x.ptr, x.ld, layer.cin
        a.wd, a.n_pad, a.k_pad = (wd.data_ptr() if wd is not None else None), n_pad, k_pad
        a.dx, a.ld_dx = (gx.ptr, gx.ld) if gx is not None else (None, 0)
        a.dw, a.db = ps.grad_ptr(layer.kernel), ps.grad_ptr(layer.bias)
        a.B, a.H, a.W = B, H, W
        a.use_mask, a.accumulate, a.dtype = int(bool(use_mask)), int(bool(accumulate)), self.code
        lib = self.lib
        keep = (gy.buf, x.buf, wd, gx.buf if gx is not None else None)

        def run(stream, a=a, keep=keep):
            L.check(lib.dd_conv3x3_bwd(C.byref(a), stream))
        run.info = self.wgrad_records[-1] if as_wgrad else self.bwd_records[-1]
        return run

    def _convt_call(self, x, y, layer, w, n_pad, k_pad, relu, gx=None, use_mask=False, accumulate=False):
        """2x2 / stride-2 transposed conv: forward (gx is None; y = output) or the fused backward (y = dy) -- csrc/dd_convt.hip."""
        B, H, W = x.B, x.H, x.W
        bwd = gx is not None
        self.convt_records.append({"flops": (3 if bwd else 1) * 2.0 * B * H * W * 4 * layer.cin * layer.cout, "B": B, "H": H, "W": W, "taps": 4,
                                   "n": layer.cout, "k": layer.cin, "backward": bwd})
        ps = self.params
        a = L.ConvTArgs()
        a.x, a.ld_x, a.cin = x.ptr, x.ld, layer.cin
        a.y, a.ld_y, a.cout = y.ptr, y.ld, layer.cout
        a.w, a.n_pad, a.k_pad = w.data_ptr(), n_pad, k_pad
        a.bias, a.relu = (None if bwd else ps.value_ptr(layer.bias)), int(bool(relu))
        if bwd:
            a.dx, a.ld_dx, a.dw, a.db = gx.ptr, gx.ld, ps.grad_ptr(layer.kernel), ps.grad_ptr(layer.bias)
        a.use_mask, a.accumulate = int(bool(use_mask)), int(bool(accumulate))
        a.B, a.H, a.W, a.dtype = B, H, W, self.code
        lib = self.lib
        keep = (x.buf, y.buf, w, gx.buf if bwd else None)
        fn = lib.dd_convt2x2_bwd if bwd else lib.dd_convt2x2_fwd

        def run(stream, a=a, keep=keep):
            L.check(fn(C.byref(a), stream))
        run.info = self.convt_records[-1]
        return run

    def _conv_ks_call(self, x, y, layer, wp, n_pad, k_pad, flags, mode=0):
        """K-streamed 3x3 conv (csrc/dd_conv_ks.hip), one launch: mode 0 = the conv itself, 5 = the 3x3/s2 transposed conv as its four output
        parities (the library runs channel blocks and parities as sub-problems of one grid)."""
        ps, lib = self.params, self.lib
        a = L.ConvKsArgs()
        a.x, a.ldx, a.cin = x.ptr, x.ld, x.C
        a.wp, a.n_pad, a.k_pad = wp.data_ptr(), n_pad, k_pad
        a.bias, a.nbias = ps.value_ptr(layer.bias), layer.cout
        a.y, a.ldy = y.ptr, y.ld
        a.n0, a.n = 0, round_up(layer.cout, 4)
        a.B, a.H, a.W = x.B, x.H, x.W
        a.mode, a.flags, a.dtype = mode, flags, self.code

        def run(stream, a=a, keep=(x.buf, y.buf, wp)):
            L.check(lib.dd_conv3x3_ks(C.byref(a), stream))
        return run

    def _bias_grad_call(self, gy, cout, bias_param):
        lib, code, ps = self.lib, self.code, self.params

        def run(stream):
            L.check(lib.dd_colsum(gy.ptr, gy.ld, cout, gy.npix, ps.grad_ptr(bias_param), code, stream))
        return run

    # ------------------------------------------------------------------ differentiable ops
    def conv(self, x, layer, relu=False, in_relu=False, res=None, out=None, split_at=None, no_backward=False):
        """tf.layers.conv2d(k x k, SAME) [+ residual] [+ ReLU]; `out` may be a channel view of a concat buffer.
        split_at: x is the concat [x[:, :split_at] | x[:, split_at:]] (the U-Net skip concat).  With more than 128 input channels in
        bf16 the 3x3 weights of a 32-channel block no longer fit LDS and the launch falls to 16-channel blocks (6 passes over the
        input for 192 -> 96); the forward then runs as conv(first part) followed by conv(second part) + that partial sum as residual:
        two launches with resident weights (192 -> 96 at 64^2, B=128: 371 -> ~250 us).  The backward is unchanged (one dgrad, one wgrad
        over the whole concat)."""
        assert layer.kind == "conv" and x.C == layer.cin, (layer.name, x.C, layer.cin)
        y = out if out is not None else self.tensor(x.B, x.H, x.W, layer.cout, relu=relu)
        y.relu = relu
        assert y.C == layer.cout
        ps = self.params
        flags = (L.OUT_RELU if relu else 0) | (L.IN_RELU if in_relu else 0)
        do_split = (split_at is not None and self.dtype in ("bf16", "f16") and layer.k == 3 and res is None and layer.cin > 128
                    and 0 < split_at < layer.cin and split_at % 8 == 0 and max(split_at, layer.cin - split_at) <= 128
                    and os.environ.get("DD_CONV_SPLIT_CONCAT", "1") != "0")
        if do_split:
            xa, xb = x.view(0, split_at), x.view(split_at, layer.cin - split_at)
            part = self.tensor(x.B, x.H, x.W, layer.cout, requires_grad=False)
            wa, taps, n_pad, ka_pad = layer.packed_k_range(0, split_at)
            wb, _, _, kb_pad = layer.packed_k_range(split_at, layer.cin - split_at)
            self.fwd(self._defer(lambda: self._conv_call(xa, wa, taps, n_pad, ka_pad, ps.value_ptr(layer.bias), layer.cout, None, None, part,
                                                         x.B, x.H, x.W, flags & L.IN_RELU, nk=(layer.cout, split_at)), "conv_igemm"))
            self.fwd(self._defer(lambda: self._conv_call(xb, wb, taps, n_pad, kb_pad, None, 0, part, None, y,
                                                         x.B, x.H, x.W, flags, nk=(layer.cout, layer.cin - split_at)), "conv_igemm"))
        elif (self.dtype in ("bf16", "f16") and layer.k == 3 and res is None and (in_relu or layer.cin > 128) and layer.cout % 4 == 0 and layer.cout <= 256
              and x.ld % 8 == 0 and y.ld % 4 == 0 and x.ch0 % 8 == 0 and y.ch0 % 4 == 0 and os.environ.get("DD_CONV_KS", "1") != "0"
              # thin layers over a short reduction (<= 32 new channels from <= 144: the 256 x 256 level of the light Tiramisu) keep their whole weight
              # image in LDS on dd_conv_igemm: measured per layer (round 5, B = 8 at 256 x 256) 47 - 64 us against 73 - 95 us K-streamed, where all
              # eight waves of a workgroup re-load the same 18 weight fragments per 64-channel slice (DD_CONV_KS_THIN=1 streams them anyway)
              and not (layer.cout <= 32 and layer.cin <= 144 and os.environ.get("DD_CONV_KS_THIN", "0") != "1")):
            # Tiramisu's dense-block convs (pre-activation, reduction over the growing concat: K = 9 x up to 1 088 channels, 16 ... 128 new
            # channels): both operands streamed per 64-channel K-slice (csrc/dd_conv_ks.hip); the LDS-weight kernel can keep none of it resident
            wp, taps, n_pad, k_pad = layer.packed("fwd")
            rec = {"flops": 2.0 * x.B * x.H * x.W * 9 * layer.cin * layer.cout, "B": x.B, "H": x.H, "W": x.W, "taps": 9, "n": layer.cout, "k": layer.cin,
                   "extra_reads": 0, "flags": flags}
            self.conv_records.append(rec)

            def ks_fwd(stream, cell=[]):
                if not cell:
                    cell.append(self._conv_ks_call(x, y, layer, wp, n_pad, k_pad, flags))
                cell[0](stream)
            ks_fwd.tag, ks_fwd.info = "conv_igemm", rec
            self.fwd(ks_fwd)
        else:
            wp, taps, n_pad, k_pad = layer.packed("fwd")
            self.fwd(self._defer(lambda: self._conv_call(x, wp, taps, n_pad, k_pad, ps.value_ptr(layer.bias), layer.cout, res, None, y,
                                                         x.B, x.H, x.W, flags, nk=(layer.cout, layer.cin)), "conv_igemm"))

        def backward():
            if not y.grad_written:
                return
            gy = y.grad()
            self._self_mask(y, gy)
            # one pass over dy and x for both gradients (3x3, <= 64 output channels, bf16 / f16 storage): csrc/dd_conv_bwd.hip
            # (more than 96 output channels with a data gradient: the register-weight data gradient + the weight-gradient role as two launches)
            # (round 6: 65 - 96 output channels with a data gradient have a fused kernel of their own, csrc/dd_conv_bwd96.hip: a 32-channel third
            #  of the input per workgroup against all output channels; DD_CONV_BWD96=0 restores the two-launch path)
            wide_ok = layer.cout <= 64 or (x.requires_grad and layer.cout <= 96 and os.environ.get("DD_CONV_BWD96", "1") != "0")
            wide_ok = wide_ok and x.B * x.H * x.W < (1 << 23)      # dd_conv3x3_bwd's own limit (linear pixel index in the DMA swizzle): larger problems split
            if (layer.k == 3 and self.dtype in ("bf16", "f16") and wide_ok and not in_relu and (x.requires_grad or layer.cin >= 16)
                    and os.environ.get("DD_FUSE_CONV_BWD", "1") != "0"):
                if x.requires_grad:
                    wd, _, dn_pad, dk_pad = layer.packed("dgrad")
                    gx = x.grad()
                    use_mask, accumulate = x.relu, x.grad_written
                    self.bwd(self._defer(lambda: self._conv_bwd_call(gy, x, layer, wd, dn_pad, dk_pad, gx, use_mask, accumulate), "conv_bwd"),
                             grad_params=[layer.kernel, layer.bias])
                    x.mark_grad_written()
                else:       # the network's first layer: weight / bias gradients only (the same launch, its data-gradient waves only feed the DMA)
                    self.bwd(self._defer(lambda: self._conv_bwd_call(gy, x, layer, None, 0, 0, None, False, False), "conv_bwd"),
                             grad_params=[layer.kernel, layer.bias])
                if res is not None and res.requires_grad:
                    self._masked_add_bwd(res, gy)
                return
            wflags = L.IN_RELU if in_relu else 0
            if (layer.k == 3 and self.dtype in ("bf16", "f16") and layer.cout > 64 and not in_relu and layer.cin >= 16 and x.B * x.H * x.W < (1 << 23)
                    and os.environ.get("DD_FUSE_CONV_BWD", "1") != "0"):
                # > 64 output channels: the weight-gradient role of the fused backward kernel per (input block, output block) pair of 64 x 64
                # channels (dx = NULL).  Measured faster than the dedicated weight-gradient kernel: 96->96 at 64x64 154 -> 115 us,
                # 128->128 at 32x32 71 -> 60 us against csrc/dd_conv_wgrad.hip
                self.bwd(self._defer(lambda: self._conv_bwd_call(gy, x, layer, None, 0, 0, None, False, False, as_wgrad=True), "conv_wgrad"),
                         grad_params=[layer.kernel, layer.bias])
                self.bwd_ops[-1].weights_only = (gy, x, layer)      # (build_backward may run several of these as one launch: _merge_weights_only)
            else:
                self.bwd(self._defer(lambda: self._wgrad_call(x, layer.cin, gy, layer.cout, ps.grad_ptr(layer.kernel), x.B, x.H, x.W, taps, wflags,
                                                              ps.grad_ptr(layer.bias), 1), "conv_wgrad"), grad_params=[layer.kernel, layer.bias])
            if x.requires_grad:
                wd, dtaps, dn_pad, dk_pad = layer.packed("dgrad")
                gx = x.grad()
                mask = x if (x.relu or in_relu) else None
                dflags = L.ACCUM if x.grad_written else 0
                self.bwd(self._defer(lambda: self._conv_call(gy, wd, dtaps, dn_pad, dk_pad, None, 0, None, mask, gx, x.B, x.H, x.W, dflags,
                                                             nk=(layer.cin, layer.cout)), "conv_igemm"))
                x.mark_grad_written()
            if res is not None and res.requires_grad:
                self._masked_add_bwd(res, gy)
        if not no_backward:      # (dense_block differentiates its convs together)
            self.on_backward(backward)
        return y

    # ------------------------------------------------------------------ Tiramisu dense block
    def dense_block(self, buf, c0, f, layers):
        """Tiramisu.__dense_block (Tiramisu.py:26-41): conv j reads relu(buf[:, :c0 + j f]) and appends f channels at c0 + j f (the concat is a view).

        Backward.  Layer by layer every conv's data gradient accumulates into the whole prefix it read: the prefix of a block is read, added to and
        re-written once per conv (O(n^2) traffic), each pass rounding the running sum to the storage type.  Here (bf16 / f16, growth <= 32
        channels) the block is differentiated in GATHER form: the gradient
        of a channel range receives the contributions of ALL later convs of the block in ONE launch -- their output gradients are one contiguous
        channel range of the gradient buffer, i.e. the reduction of a deep-K conv with the stacked transposed kernels (csrc/dd_conv_ks.hip,
        DD_ACCUM) -- masked by the consumers' ReLU, added to what the consumers outside the block stored, rounded once.  Ranges go last to first
        (a conv's weight gradient needs its output gradient complete), the block's input prefix last."""
        n = len(layers)
        # Measured (B = 8, 256x256): growth 16 / 24 / 32 (BASELINE cfg-3): step 7.81 -> 6.96 ms in gather form; growth 64 / 96 / 128 (the heavy
        # stress configuration): 23.98 -> 26.45 ms -- there the layer-wise data gradients have K = 9 x 64 ... 128, whose weights stay resident in LDS
        # (csrc/dd_conv_igemm.hip at ~700 TFLOP/s), while the gather of a 384 ... 800-channel prefix re-streams its operands per 64-channel
        # block (~400 TFLOP/s).  DD_DENSE_GATHER=1 / 0 forces either form.
        force = os.environ.get("DD_DENSE_GATHER", "")
        gather = (self.dtype in ("bf16", "f16") and f % 8 == 0 and c0 % 8 == 0 and buf.ld % 8 == 0 and os.environ.get("DD_CONV_KS", "1") != "0"
                  and (force == "1" or (force != "0" and f <= 32)))
        for j, lay in enumerate(layers):
            cj = c0 + j * f
            assert lay.cin == cj and lay.cout == f and lay.k == 3
            self.conv(buf.view(0, cj), lay, relu=False, in_relu=True, out=buf.view(cj, f, relu=False), no_backward=gather)
        if not gather or not bool(getattr(self, "training", True)):      # the stacked images below only feed the backward
            return c0 + n * f
        ps, lib, code = self.params, self.lib, self.code

        def stacked_image(s0, sn, later):
            """MFMA operand [9][n_pad][k_pad]: row s - s0 (a channel of the target range), column block of conv i = its kernel W_i[8 - tap][s][:]."""
            n_pad, k_tot = round_up(sn, 16), sum(l.cout for l in later)
            k_pad = round_up(k_tot, 64 // _ESZ[self.dtype])
            img = torch.zeros(9 * n_pad * k_pad, dtype=_TORCH_DT[self.dtype], device=self.device)
            koff = 0
            for l in later:
                self.register_pack(l.kernel, img, 9, sn, l.cout, n_pad, l.cout, l.cin * l.cout, l.cout, 1, 1, src_offset=s0 * l.cout,
                                   dst_offset=koff, dst_ld=k_pad, dst_tap_stride=n_pad * k_pad)
                koff += l.cout
            return img, n_pad, k_pad

        plans = []      # (target start, width, first later conv): images are registered now, so that they are packed with every other layer's
        for j in range(n - 1):
            plans.append((c0 + j * f, f, j + 1) + stacked_image(c0 + j * f, f, layers[j + 1:]))
        plans.append((0, c0, 0) + stacked_image(0, c0, layers))

        def backward():
            gbuf = buf.grad()
            assert buf.grad_written, "a dense-block buffer accumulates: its gradient storage must be zero-initialised"

            def gather_call(s0, sn, first, img, n_pad, k_pad):
                kch = (n - first) * f
                x0 = c0 + first * f
                rec = {"flops": 2.0 * buf.B * buf.H * buf.W * 9 * kch * sn, "B": buf.B, "H": buf.H, "W": buf.W, "taps": 9, "n": sn, "k": kch,
                       "extra_reads": 2, "flags": L.ACCUM}
                self.conv_records.append(rec)
                # (thin target ranges stay on the K-streamed kernel: dd_conv_igemm's accumulate epilogue rounds the sum before it adds -- measured on the
                #  prefix gradient 2.7e-3 against the 3e-4 of the single rounding this form exists for; it was 4 % faster on those launches)
                a = L.ConvKsArgs()
                a.x, a.ldx, a.cin = gbuf.ptr + x0 * _ESZ[self.dtype], gbuf.ld, kch
                a.wp, a.n_pad, a.k_pad = img.data_ptr(), n_pad, k_pad
                a.bias, a.nbias = None, 0
                a.y, a.ldy = gbuf.ptr + s0 * _ESZ[self.dtype], gbuf.ld
                a.mask, a.ldmask = buf.ptr + s0 * _ESZ[self.dtype], buf.ld
                a.n0, a.n = 0, round_up(sn, 4)
                a.B, a.H, a.W = buf.B, buf.H, buf.W
                a.mode, a.flags, a.dtype = 0, L.ACCUM, code

                def dense_gather(stream, a=a, keep=(gbuf.buf, buf.buf, img)):
                    L.check(lib.dd_conv3x3_ks(C.byref(a), stream))
                dense_gather.tag, dense_gather.info = "conv_igemm", rec
                self.bwd(dense_gather)

            # Round 5: the weight gradients of the block's convs are ONE launch behind the gathers (their output gradients are one contiguous
            # channel range, their inputs nested prefixes: dd_conv_wgrad's stacked form).  With 16 ... 32 new channels per conv a launch of its own
            # filled a quarter of the kernel's 64-channel slices and four of its eight waves: 95 - 125 us each for ~10 us of HBM traffic.
            stack = (1 < n <= 8 and os.environ.get("DD_WGRAD_STACK", "1") != "0" and os.environ.get("DD_WGRAD_DMA", "1") != "0"
                     and -(-(c0 + (n - 1) * f) // 64) * -(-(n * f) // 64) <= 64)
            for j in reversed(range(n)):
                lay, cj = layers[j], c0 + j * f
                if j < n - 1:
                    gather_call(*plans[j])
                if stack:
                    continue
                x, gy = buf.view(0, cj), gbuf.view(cj, f)
                self.bwd(self._defer(lambda x=x, gy=gy, lay=lay: self._wgrad_call(x, lay.cin, gy, lay.cout, ps.grad_ptr(lay.kernel), x.B, x.H, x.W, 9, L.IN_RELU,
                                                                                   ps.grad_ptr(lay.bias), 1), "conv_wgrad"), grad_params=[lay.kernel, lay.bias])
            if stack:
                self.bwd(self._wgrad_stack_call(buf.view(0, c0 + (n - 1) * f), gbuf.view(c0, n * f), layers, c0, f, buf.B, buf.H, buf.W, L.IN_RELU),
                         "conv_wgrad", grad_params=[p_ for lay in layers for p_ in (lay.kernel, lay.bias)])
            if buf.requires_grad and c0 > 0:
                gather_call(*plans[n - 1])
            buf.mark_grad_written()
        self.on_backward(backward)
        return c0 + n * f

    def _self_mask(self, y, gy):
        if not y.self_mask:
            return
        lib, code = self.lib, self.code

        def run(stream):
            L.check(lib.dd_masked_add(gy.ptr, gy.ld, gy.ptr, gy.ld, y.ptr, y.ld, y.Cp, y.npix, 0, code, stream))
        self.bwd(run)

    def _masked_add_bwd(self, dst_tensor, g_src):
        """dst_tensor.grad (+)= g_src * (dst_tensor > 0 if it is a ReLU output)."""
        gd = dst_tensor.grad()
        acc = 1 if dst_tensor.grad_written else 0
        mask = dst_tensor if dst_tensor.relu else None
        lib, code = self.lib, self.code

        def run(stream):
            L.check(lib.dd_masked_add(gd.ptr, gd.ld, g_src.ptr, g_src.ld, mask.ptr if mask is not None else None,
                                      mask.ld if mask is not None else 0, dst_tensor.Cp, dst_tensor.npix, acc, code, stream))
        self.bwd(run)
        dst_tensor.mark_grad_written()

    def conv_transpose2(self, x, layer, out=None, relu=True):
        """tf.layers.conv2d_transpose(2x2, strides 2) + ReLU (UNet.py:54-59)."""
        assert layer.kind == "convT2" and x.C == layer.cin and layer.cout % 8 == 0
        y = out if out is not None else self.tensor(x.B, 2 * x.H, 2 * x.W, layer.cout, relu=relu)
        y.relu = relu
        ps = self.params
        wp, taps, n_pad, k_pad = layer.packed("fwd")
        # streaming kernels of csrc/dd_convt.hip (bf16 / f16 storage): forward for <= 96 output channels, fused backward for <= 64
        stream_ok = (self.dtype in ("bf16", "f16") and layer.cin <= 128 and layer.cout % 16 == 0 and os.environ.get("DD_CONVT_STREAM", "1") != "0")
        if stream_ok and layer.cout <= 96:
            self.fwd(self._defer(lambda: self._convt_call(x, y, layer, wp, n_pad, k_pad, relu), "convt"))
        else:
            flags = L.PIXSHUF | (L.OUT_RELU if relu else 0)
            yv = DT(y.buf, y.B, y.H, y.W, 4 * layer.cout, 4 * layer.cout, y.ch0, y.dtype)   # n = (a,b,co)
            self.fwd(self._defer(lambda: self._conv_call(x, wp, taps, n_pad, k_pad, ps.value_ptr(layer.bias), layer.cout, None, None, yv,
                                                         x.B, x.H, x.W, flags, nk=(4 * layer.cout, layer.cin)), "conv_igemm"))

        def backward():
            if not y.grad_written:
                return
            gy = y.grad()
            if stream_ok and layer.cout <= 128 and x.requires_grad:
                wd, _, dn_pad, dk_pad = layer.packed("dgrad")
                gx = x.grad()
                use_mask, accumulate = x.relu, x.grad_written
                self.bwd(self._defer(lambda: self._convt_call(x, gy, layer, wd, dn_pad, dk_pad, False, gx=gx, use_mask=use_mask, accumulate=accumulate),
                                     "convt"), grad_params=[layer.kernel, layer.bias])
                x.mark_grad_written()
                return
            self.bwd(self._defer(lambda: self._wgrad_call(gy, layer.cout, x, layer.cin, ps.grad_ptr(layer.kernel), x.B, x.H, x.W, 4, L.GATHER2X2,
                                                          ps.grad_ptr(layer.bias), 2), "conv_wgrad"), grad_params=[layer.kernel, layer.bias])
            if x.requires_grad:
                wd, dtaps, dn_pad, dk_pad = layer.packed("dgrad")
                gx = x.grad()
                mask = x if x.relu else None
                dflags = L.GATHER2X2 | (L.ACCUM if x.grad_written else 0)
                self.bwd(self._defer(lambda: self._conv_call(gy, wd, dtaps, dn_pad, dk_pad, None, 0, None, mask, gx, x.B, x.H, x.W, dflags,
                                                             nk=(layer.cin, layer.cout)), "conv_igemm"))
                x.mark_grad_written()
        self.on_backward(backward)
        return y

    def conv_transpose3(self, x, layer, out=None, relu=True):
        """tf.layers.conv2d_transpose(3x3, strides 2, SAME) + ReLU (Tiramisu.py:60-65) as a 3x3 SAME conv of the
        zero-stuffed input (x[i,j] placed at (2i+1, 2j+1)) with the flipped kernel (o = 2i + a)."""
        assert layer.kind == "convT3" and x.C == layer.cin
        y = out if out is not None else self.tensor(x.B, 2 * x.H, 2 * x.W, layer.cout, relu=relu)
        y.relu = relu
        ps = self.params
        lib, code = self.lib, self.code
        wp, taps, n_pad, k_pad = layer.packed("fwd")
        # bf16 / f16: the FORWARD runs as the four output-parity sub-convolutions on the input grid (9 real taps instead of 36, no stuffed tensor:
        # csrc/dd_conv_ks.hip modes 1..4); the backward still differentiates the zero-stuffed form, so a training graph keeps the stuffed copy
        parity = (self.dtype in ("bf16", "f16") and layer.cout % 4 == 0 and layer.cout <= 128 and x.ld % 8 == 0 and y.ld % 4 == 0 and x.ch0 % 8 == 0 and y.ch0 % 4 == 0
                  and os.environ.get("DD_CONVT3_PARITY", "1") != "0")
        # ... and so does the BACKWARD (round 3): with the output gradient rearranged by output parity (dd_space_to_depth2) the data gradient is a
        # 2 x 2-tap conv on the input grid (dd_conv3x3_ks mode 6) and the filter gradient one GEMM over nine shifted channel windows
        # (dd_convt3_wgrad): no zero-stuffed tensor, no 4x-redundant MACs.  DD_CONVT3_S2D_BWD=0: differentiate the zero-stuffed form instead.
        s2d_bwd = (parity and bool(getattr(self, "training", True)) and x.ld % 8 == 0 and x.ch0 % 8 == 0 and layer.cin % 4 == 0
                   and os.environ.get("DD_CONVT3_S2D_BWD", "1") != "0")      # (the s2d image only feeds the backward)
        need_z = (not parity) or (bool(getattr(self, "training", True)) and not s2d_bwd)
        z = self.tensor(x.B, 2 * x.H, 2 * x.W, x.C, requires_grad=x.requires_grad) if need_z else None
        if s2d_bwd:
            # data-gradient operand [9][cin][4 cp]: image tap (1 + (a == 2), 1 + (b == 2)), column block plane(a, b) = K[a][b][:, :] transposed
            cp = round_up(layer.cout, 16)
            s2d_n_pad, s2d_k_pad = round_up(layer.cin, 16), round_up(4 * cp, 64 // _ESZ[self.dtype])
            s2d_img = torch.zeros(9 * s2d_n_pad * s2d_k_pad, dtype=_TORCH_DT[self.dtype], device=self.device)
            for ta, tb, _, _, plane, tap_img in convt3_s2d_blocks():
                self.register_pack(layer.kernel, s2d_img, 1, layer.cin, layer.cout, s2d_n_pad, cp, 0, 1, layer.cin, 0,
                                   src_offset=(ta * 3 + tb) * layer.cout * layer.cin,
                                   dst_offset=tap_img * s2d_n_pad * s2d_k_pad + plane * cp, dst_ld=s2d_k_pad, dst_tap_stride=s2d_n_pad * s2d_k_pad)
        if need_z:
            def stuff(stream):
                L.check(lib.dd_zero_stuff(x.ptr, x.ld, z.ptr, z.ld, x.Cp, x.B, x.H, x.W, code, stream))
            self.fwd(stuff)
        if parity:
            rec = {"flops": 2.0 * x.B * x.H * x.W * 9 * layer.cin * layer.cout, "B": x.B, "H": x.H, "W": x.W, "taps": 9, "n": layer.cout, "k": layer.cin,
                   "extra_reads": 0, "flags": L.OUT_RELU if relu else 0}
            self.conv_records.append(rec)

            def ks_convt(stream, cell=[]):
                if not cell:
                    cell.append(self._conv_ks_call(x, y, layer, wp, n_pad, k_pad, L.OUT_RELU if relu else 0, mode=5))
                cell[0](stream)
            ks_convt.tag, ks_convt.info = "conv_igemm", rec
            self.fwd(ks_convt)
        else:
            self.fwd(self._defer(lambda: self._conv_call(z, wp, taps, n_pad, k_pad, ps.value_ptr(layer.bias), layer.cout, None, None, y,
                                                         z.B, z.H, z.W, L.OUT_RELU if relu else 0, nk=(layer.cout, layer.cin)), "conv_igemm"))

        def backward_s2d():
            gy = y.grad()
            self._self_mask(y, gy)
            sd = self.tensor(x.B, x.H, x.W, 4 * cp, requires_grad=False)      # the output gradient by output parity
            es = _ESZ[self.dtype]

            def s2d(stream):
                L.check(lib.dd_space_to_depth2(gy.ptr, gy.ld, sd.ptr, sd.ld, layer.cout, cp, x.B, x.H, x.W, code, stream))
            self.bwd(s2d)
            self.bwd(self._bias_grad_call(gy, layer.cout, layer.bias), grad_params=[layer.bias])
            wrec = {"flops": 2.0 * x.B * x.H * x.W * 9 * layer.cin * layer.cout, "B": x.B, "H": x.H, "W": x.W, "taps": 9, "n": layer.cin, "k": layer.cout}
            self.wgrad_records.append(wrec)

            def convt3_wgrad(stream, cell=[]):
                if not cell:      # (pointers into the parameter arena exist once the graph is finalised)
                    wa = L.ConvT3WgradArgs()
                    wa.s, wa.lds, wa.cout, wa.cp = sd.ptr, sd.ld, layer.cout, cp
                    wa.x, wa.ldx, wa.cin = x.ptr, x.ld, layer.cin
                    wa.dk = ps.grad_ptr(layer.kernel)
                    wa.B, wa.H, wa.W, wa.dtype = x.B, x.H, x.W, code
                    cell.append(wa)
                L.check(lib.dd_convt3_wgrad(C.byref(cell[0]), stream))
            convt3_wgrad.tag, convt3_wgrad.info = "conv_wgrad", wrec
            self.bwd(convt3_wgrad, grad_params=[layer.kernel])
            if x.requires_grad:
                gx = x.grad()
                rec = {"flops": 2.0 * x.B * x.H * x.W * 9 * layer.cin * layer.cout, "B": x.B, "H": x.H, "W": x.W, "taps": 9, "n": layer.cin, "k": layer.cout,
                       "extra_reads": (1 if x.relu else 0) + (1 if x.grad_written else 0), "flags": L.ACCUM if x.grad_written else 0}
                self.conv_records.append(rec)
                a = L.ConvKsArgs()
                a.x, a.ldx, a.cin = sd.ptr, sd.ld, 4 * cp
                a.wp, a.n_pad, a.k_pad = s2d_img.data_ptr(), s2d_n_pad, s2d_k_pad
                a.bias, a.nbias = None, 0
                a.y, a.ldy = gx.ptr, gx.ld
                a.mask, a.ldmask = (x.ptr, x.ld) if x.relu else (None, 0)
                a.n0, a.n = 0, round_up(layer.cin, 4)
                a.B, a.H, a.W = x.B, x.H, x.W
                a.mode, a.flags, a.dtype = 6, (L.ACCUM if x.grad_written else 0), code

                def convt3_dgrad(stream, a=a, keep=(sd.buf, gx.buf, s2d_img)):
                    L.check(lib.dd_conv3x3_ks(C.byref(a), stream))
                convt3_dgrad.tag, convt3_dgrad.info = "conv_igemm", rec
                self.bwd(convt3_dgrad)
                x.mark_grad_written()

        def backward():
            if not y.grad_written:
                return
            if s2d_bwd:
                return backward_s2d()
            gy = y.grad()
            self._self_mask(y, gy)
            # out[t][co][ci] = sum_p gy[p (+) t][co] * z[p][ci] == dKernel in TF layout [kh,kw,C_out,C_in]
            self.bwd(self._defer(lambda: self._wgrad_call(gy, layer.cout, z, layer.cin, ps.grad_ptr(layer.kernel), z.B, z.H, z.W, 9, 0,
                                                          ps.grad_ptr(layer.bias), 2), "conv_wgrad"), grad_params=[layer.kernel, layer.bias])
            if x.requires_grad:
                wd, dtaps, dn_pad, dk_pad = layer.packed("dgrad")
                gz = z.grad()
                self.bwd(self._defer(lambda: self._conv_call(gy, wd, dtaps, dn_pad, dk_pad, None, 0, None, None, gz, z.B, z.H, z.W, 0,
                                                             nk=(layer.cin, layer.cout)), "conv_igemm"))
                gx = x.grad()
                mask = x if x.relu else None
                acc = 1 if x.grad_written else 0

                def unstuff(stream):
                    L.check(lib.dd_zero_unstuff(gz.ptr, gz.ld, gx.ptr, gx.ld, mask.ptr if mask is not None else None,
                                                mask.ld if mask is not None else 0, x.Cp, x.B, x.H, x.W, acc, code, stream))
                self.bwd(unstuff)
                x.mark_grad_written()
        self.on_backward(backward)
        return y

    def maxpool(self, x, pool, stride, out=None):
        """tf.layers.max_pooling2d(padding='same') (UNet.py:42-44, Tiramisu.py:55-57)."""
        OH, OW = -(-x.H // stride), -(-x.W // stride)
        y = out if out is not None else self.tensor(x.B, OH, OW, x.C, requires_grad=x.requires_grad)
        assert (y.H, y.W, y.C) == (OH, OW, x.C)
        # the argmax plane only feeds the backward: inference graphs do not store it
        idx = torch.zeros((x.B, OH, OW, x.Cp), dtype=torch.uint8, device=self.device) if bool(getattr(self, "training", True)) else None
        lib, code = self.lib, self.code

        def run(stream):
            L.check(lib.dd_maxpool_fwd(x.ptr, x.ld, y.ptr, y.ld, idx.data_ptr() if idx is not None else None, x.Cp, x.B, x.H, x.W, pool, stride,
                                       1 if x.relu else 0, code, stream))
        self.fwd(run, "maxpool")

        def backward():
            if not (y.grad_written and x.requires_grad):
                return
            gy, gx = y.grad(), x.grad()
            mask = None          # the ReLU-backward mask of x is folded into idx by the forward kernel (relu_mask)
            acc = 1 if x.grad_written else 0

            def runb(stream):
                L.check(lib.dd_maxpool_bwd(gy.ptr, gy.ld, idx.data_ptr(), gx.ptr, gx.ld, mask.ptr if mask is not None else None,
                                           mask.ld if mask is not None else 0, x.Cp, x.B, x.H, x.W, pool, stride, acc, code, stream))
            self.bwd(runb, "maxpool")
            x.mark_grad_written()
        self.on_backward(backward)
        return y

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _defer(make, tag="pointwise"):
        """Bind a launch lazily: parameter pointers exist only after ParamStore.finalize()."""
        cell = []

        def run(stream):
            if not cell:
                cell.append(make())
                run.info = getattr(cell[0], "info", None)
            cell[0](stream)
        run.tag = tag
        run.info = None
        run.origin = getattr(make, "__qualname__", "")      # which builder queued this launch (tools/step_breakdown.py)
        return run

    def finalize(self, seed=2):
        self.params.finalize(self.device, seed)

    @staticmethod
    def stream_ptr():
        return torch.cuda.current_stream().cuda_stream

    def run(self, ops, stream=None):
        s = self.stream_ptr() if stream is None else stream
        for op in ops:
            op(s)
