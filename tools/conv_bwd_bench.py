"""One 3x3 layer's backward at bench size, repeated:  python tools/conv_bwd_bench.py [cin cout H B dtype reps]
Prints the average time of the backward launches (fused conv_bwd, or dgrad + wgrad with DD_FUSE_CONV_BWD=0); a small target for
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_family.py <dir> conv_bwd)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepdenoiser_amd import engine  # noqa: E402

cin, cout, H, B = [int(v) for v in (sys.argv[1:5] + ["64", "64", "128", "128"][len(sys.argv) - 1:])]
dtype = sys.argv[5] if len(sys.argv) > 5 else "bf16"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
g = engine.Graph("cuda", dtype)
x = g.tensor(B, H, H, cin, relu=True, requires_grad=True)
lay = g.layer("t/conv2d", 3, cin, cout)
y = g.conv(x, lay, relu=True)
y.mark_grad_written()
g.build_backward()
g.finalize()
x.buf.copy_(torch.relu(torch.randn(x.buf.shape, device="cuda")).to(x.buf.dtype))
g.params.values.normal_(0, 0.05)
g.run(g.pack_ops)
g.run(g.fwd_ops)
y.grad().buf.copy_(torch.randn(y.buf.shape, device="cuda").to(y.buf.dtype))
s = g.stream_ptr()
for _ in range(2):
    g.run(g.bwd_ops)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    g.run(g.bwd_ops)
e1.record()
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / reps
px = B * H * H
alg = px * 2 * (cout + 2 * cin)
print("%d->%d %dx%d B=%d %s: backward %s = %.1f us per layer; algorithmic bytes (dy + x + dx) %.0f MB -> %.2f TB/s; %.0f TFLOP/s" % (
    cin, cout, H, H, B, dtype, [getattr(op, "tag", "?") for op in g.bwd_ops], us, alg / 1e6, alg / us / 1e6, 4.0 * px * 9 * cin * cout / us / 1e6))
