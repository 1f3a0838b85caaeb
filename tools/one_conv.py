"""Runs one conv layer's fwd kernel N times (for rocprofv3 PMC collection).  args: k cin cout H W B iters [dgrad|wgrad]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from deepdenoiser_amd.engine import Graph
k, cin, cout, H, W, B, iters = [int(v) for v in sys.argv[1:8]]
which = sys.argv[8] if len(sys.argv) > 8 else "fwd"
g = Graph("cuda", "bf16")
x = g.tensor(B, H, W, cin, relu=True, requires_grad=True); x.buf.normal_()
lay = g.layer("b/conv2d", k, cin, cout)
y = g.conv(x, lay, relu=True); y.mark_grad_written(); g.build_backward(); g.finalize(); y.grad().buf.normal_()
s = g.stream_ptr(); g.run(g.pack_ops)
tag = "conv_wgrad" if which == "wgrad" else "conv_igemm"
op = [o for o in (g.fwd_ops if which == "fwd" else g.bwd_ops) if getattr(o, "tag", "") == tag][0]
for _ in range(iters): op(s)
torch.cuda.synchronize()
