"""1x1 layers through the engine, HIP-event timed per launch (forward, data gradient, weight gradient):
    [DD_LIB=tools/exp/libdd_<variant>.so] [DD_CONV_PW=0] python tools/pw_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepdenoiser_amd.engine import Graph          # noqa: E402

SHAPES = [  # cin, cout, H, W, B
    (704, 704, 128, 128, 8),
    (320, 320, 256, 256, 8),
    (176, 176, 128, 128, 8),
    (80, 80, 256, 256, 8),
    (640, 25, 256, 256, 8),
    (160, 25, 256, 256, 8),
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


print(os.environ.get("DD_LIB", "default"), "DD_CONV_PW=" + os.environ.get("DD_CONV_PW", "1"))
ONLY = int(os.environ.get("PW_ONLY", "-1"))
for idx, (cin, cout, H, W, B) in enumerate(SHAPES):
    if ONLY >= 0 and idx != ONLY:
        continue
    g = Graph("cuda", "bf16")
    x = g.tensor(B, H, W, cin, relu=False, requires_grad=True)
    x.buf.normal_()
    lay = g.layer("b/conv2d", 1, cin, cout)
    y = g.conv(x, lay, relu=False, in_relu=True)
    y.mark_grad_written()
    g.build_backward()
    g.finalize()
    y.grad().buf.normal_()
    s = g.stream_ptr()
    g.run(g.pack_ops)
    fl = 2.0 * B * H * W * cin * cout
    line = "  %4d -> %-4d @%dx%d B%d:" % (cin, cout, H, W, B)
    t = timeit(lambda: g.fwd_ops[-1](s))
    line += "  fwd %7.1f us %6.1f TF/s" % (t, fl / t / 1e6)
    for op in g.bwd_ops:
        t = timeit(lambda: op(s))
        line += "  %s %7.1f us %6.1f TF/s" % (getattr(op, "tag", op.__name__), t, fl / t / 1e6)
    print(line, flush=True)
