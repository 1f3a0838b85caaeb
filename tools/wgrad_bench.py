"""3x3 weight-gradient launches through the engine, HIP-event timed: [DD_LIB=tools/exp/libdd_<variant>.so] python tools/wgrad_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepdenoiser_amd.engine import Graph          # noqa: E402

SHAPES = [  # cin, cout, H, W, B, in_relu
    (576, 64, 256, 256, 8, True),
    (1088, 96, 128, 128, 8, True),
    (144, 16, 256, 256, 8, True),
    (128, 128, 64, 64, 128, False),
    (192, 96, 64, 64, 128, False),
]


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


print(os.environ.get("DD_LIB", "default"))
for cin, cout, H, W, B, in_relu in SHAPES:
    g = Graph("cuda", "bf16")
    x = g.tensor(B, H, W, cin, relu=not in_relu, requires_grad=False)
    x.buf.normal_()
    lay = g.layer("b/conv2d", 3, cin, cout)
    y = g.conv(x, lay, relu=False, in_relu=in_relu)
    y.mark_grad_written()
    g.build_backward()
    g.finalize()
    y.grad().buf.normal_()
    s = g.stream_ptr()
    g.run(g.pack_ops)
    fl = 2.0 * B * H * W * 9 * cin * cout
    line = "  %4d -> %-4d @%dx%d B%d:" % (cin, cout, H, W, B)
    for op in g.bwd_ops:
        t = timeit(lambda: op(s))
        line += "  %s %7.1f us %6.1f TF/s" % (getattr(op, "tag", op.__name__), t, fl / t / 1e6)
    print(line, flush=True)
