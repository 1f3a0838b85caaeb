"""Per-layer microbenchmark of the MFMA kernels through the C-ABI (HIP-event timed on the launch stream).
    python tools/conv_bench.py [igemm|wgrad|all] [bf16|f32]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepdenoiser_amd import _lib as L            # noqa: E402
from deepdenoiser_amd.engine import Graph          # noqa: E402

# (name, k, cin, cout, H, W, B)  -- the flagship's layer shapes at B=32 tiles of 128x128
SHAPES = [
    ("first 32->64 @128", 3, 32, 64, 128, 128, 32),
    ("64->64 @128", 3, 64, 64, 128, 128, 32),
    ("128->64 @128", 3, 128, 64, 128, 128, 32),
    ("dgrad-like 64->128 @128", 3, 64, 128, 128, 128, 32),
    ("64->96 @64", 3, 64, 96, 64, 64, 32),
    ("96->96 @64", 3, 96, 96, 64, 64, 32),
    ("192->96 @64", 3, 192, 96, 64, 64, 32),
    ("96->128 @32", 3, 96, 128, 32, 32, 32),
    ("128->128 @32", 3, 128, 128, 32, 32, 32),
    ("1x1 64->25 @128", 1, 64, 25, 128, 128, 32),
    ("1x1 25->25 @128", 1, 25, 25, 128, 128, 32),
    ("compose 24->24 @128", 3, 24, 24, 128, 128, 32),
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
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main(which="all", dtype="bf16"):
    for name, k, cin, cout, H, W, B in SHAPES:
        g = Graph("cuda", dtype)
        x = g.tensor(B, H, W, cin, relu=True, requires_grad=True)
        x.buf.normal_()
        lay = g.layer("b/conv2d", k, cin, cout)
        y = g.conv(x, lay, relu=True)
        y.mark_grad_written()
        g.build_backward()
        g.finalize()
        y.grad().buf.normal_()
        s = g.stream_ptr()
        g.run(g.pack_ops)
        flops = 2.0 * B * H * W * k * k * cin * cout
        line = "%-26s" % name
        if which in ("igemm", "all"):
            fwd = [op for op in g.fwd_ops if getattr(op, "tag", "") == "conv_igemm"][0]
            t = timeit(lambda: fwd(s))
            line += " fwd %8.1f us %7.1f TF/s" % (t, flops / t / 1e6)
            dg = [op for op in g.bwd_ops if getattr(op, "tag", "") == "conv_igemm"][0]
            t = timeit(lambda: dg(s))
            line += " | dgrad %8.1f us %7.1f TF/s" % (t, flops / t / 1e6)
        if which in ("wgrad", "all"):
            wg = [op for op in g.bwd_ops if getattr(op, "tag", "") == "conv_wgrad"][0]
            t = timeit(lambda: wg(s))
            line += " | wgrad %8.1f us %7.1f TF/s" % (t, flops / t / 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main(*(sys.argv[1:]))
