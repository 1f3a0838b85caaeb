"""The weights-only mode of dd_conv3x3_bwd (dx = NULL) on one layer at bench size:  python tools/wgrad_only_bench.py [cin cout H B reps]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepdenoiser_amd import _lib as L  # noqa: E402

lib = L.load()
shapes = [(96, 96, 64, 128), (192, 96, 64, 128), (128, 128, 32, 128), (64, 64, 128, 128)] if len(sys.argv) < 5 else [tuple(int(v) for v in sys.argv[1:5])]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
for cin, cout, H, B in shapes:
    x = torch.relu(torch.randn(B, H, H, cin, device="cuda")).bfloat16()
    dy = torch.randn(B, H, H, cout, device="cuda").bfloat16()
    dw = torch.zeros(9, cin, cout, device="cuda")
    db = torch.zeros(cout, device="cuda")
    a = L.ConvBwdArgs()
    C.memset(C.byref(a), 0, C.sizeof(a))
    a.dy, a.x, a.dw, a.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr()
    a.ld_dy, a.ld_x, a.cout, a.cin, a.B, a.H, a.W, a.dtype = cout, cin, cout, cin, B, H, H, L.DD_BF16
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.check(lib.dd_conv3x3_bwd(C.byref(a), s))
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        L.check(lib.dd_conv3x3_bwd(C.byref(a), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("weights-only %d->%d %dx%d B=%d: %.1f us, %.0f TFLOP/s (%.1f %% of 2500)" % (cin, cout, H, H, B, us, 2.0 * B * H * H * 9 * cin * cout / us / 1e6, 2.0 * B * H * H * 9 * cin * cout / us / 1e6 / 25))
