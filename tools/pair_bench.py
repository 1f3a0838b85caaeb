"""Two 64-channel 3x3 convs + ReLU: one fused launch (csrc/dd_conv_pair.hip) against two register-weight launches: python tools/pair_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepdenoiser_amd.engine import Graph          # noqa: E402


def timeit(fn, iters=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for cin, B, dt in ((64, 128, "bf16"), (64, 209, "f16"), (32, 209, "f16")):
    out = []
    for fused in (False, True):
        g = Graph("cuda", dt)
        g.training = False
        x = g.tensor(B, 128, 128, cin, relu=True, requires_grad=False)
        x.buf.normal_()
        l1, l2 = g.layer("b/conv2d", 3, cin, 64), g.layer("b/conv2d_1", 3, 64, 64)
        if fused:
            g.conv_pair(x, l1, l2)
        else:
            g.conv(g.conv(x, l1, relu=True), l2, relu=True)
        g.finalize()
        s = g.stream_ptr()
        g.run(g.pack_ops)
        out.append(timeit(lambda: [op(s) for op in g.fwd_ops]))
    fl = 2.0 * B * 128 * 128 * 9 * (cin * 64 + 64 * 64)
    print("%d -> 64 -> 64 @128 B%d %s: two launches %.1f us (%.0f TF/s), fused %.1f us (%.0f TF/s): %.2fx" % (cin, B, dt, out[0], fl / out[0] / 1e6, out[1], fl / out[1] / 1e6, out[0] / out[1]), flush=True)
