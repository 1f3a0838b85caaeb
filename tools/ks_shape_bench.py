"""Forward time of single pre-activation 3x3 layers of the Tiramisu shapes (Tiramisu.py:26-41), per kernel choice:
    python tools/ks_shape_bench.py            (DD_CONV_KS_THIN=1 forces the K-streamed kernel on thin layers)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepdenoiser_amd.engine import Graph  # noqa: E402


def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


shapes = [(296, 16, 128, 8), (272, 24, 128, 8), (200, 24, 128, 8), (152, 24, 128, 8), (144, 16, 256, 8), (96, 16, 256, 8), (48, 16, 256, 8),
          (176, 32, 64, 8), (304, 32, 64, 8), (576, 64, 256, 8), (1088, 96, 128, 8), (320, 64, 256, 8)]
for cin, cout, H, B in shapes:
    g = Graph("cuda", "bf16")
    g.training = False
    buf = g.tensor(B, H, H, cin + cout, relu=False, requires_grad=False)
    buf.buf.normal_()
    lay = g.layer("b/conv2d", 3, cin, cout)
    g.conv(buf.view(0, cin), lay, relu=False, in_relu=True, out=buf.view(cin, cout, relu=False))
    g.finalize(); s = g.stream_ptr(); g.run(g.pack_ops)
    op = g.fwd_ops[-1]
    t = timeit(lambda: op(s))
    mb = B * H * H * (cin + cout) * 2 / 1e6
    print("%4d -> %3d  %3dx%-3d B=%d  %-12s %7.1f us  %6.1f TF/s  %6.1f MB = %5.1f us at 5 TB/s" % (
        cin, cout, H, H, B, getattr(op, "tag", getattr(op, "__name__", "?")), t, 2.0 * B * H * H * 9 * cin * cout / t / 1e6, mb, mb / 5.0), flush=True)
