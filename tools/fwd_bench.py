"""Forward 3x3 conv launches through the engine, HIP-event timed: [DD_LIB=tools/exp/libdd_<variant>.so] python tools/fwd_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepdenoiser_amd.engine import Graph          # noqa: E402

SHAPES = [  # name, cin, cout, H, W, B, in_relu
    ("64->64 @128 B128 (rw8<2,8>)", 64, 64, 128, 128, 128, False),
    ("96->96 @64 B128 (rw8<3,12>)", 96, 96, 64, 64, 128, False),
    ("128->64 @128 B128 (rw8<4,8>)", 128, 64, 128, 128, 128, False),
    ("128->128 @32 B128 (rw8<4,8>)", 128, 128, 32, 32, 128, False),
    ("64->64 @128 B209 f16-like", 64, 64, 128, 128, 209, False),
    ("dense 576->64 @256 B8 (ks)", 576, 64, 256, 256, 8, True),
    ("dense 1088->96 @128 B8 (ks)", 1088, 96, 128, 128, 8, True),
    ("dense 144->16 @256 B8 (ks)", 144, 16, 256, 256, 8, True),
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
ONLY = int(os.environ.get("FWD_ONLY", "-1"))
for idx, (name, cin, cout, H, W, B, in_relu) in enumerate(SHAPES):
    if ONLY >= 0 and idx != ONLY:
        continue
    g = Graph("cuda", "bf16")
    x = g.tensor(B, H, W, cin, relu=not in_relu, requires_grad=False)
    x.buf.normal_()
    lay = g.layer("b/conv2d", 3, cin, cout)
    g.conv(x, lay, relu=not in_relu, in_relu=in_relu)
    g.finalize()
    s = g.stream_ptr()
    g.run(g.pack_ops)
    t = timeit(lambda: g.fwd_ops[-1](s))
    print("  %-32s %8.1f us %7.1f TF/s" % (name, t, 2.0 * B * H * W * 9 * cin * cout / t / 1e6), flush=True)
