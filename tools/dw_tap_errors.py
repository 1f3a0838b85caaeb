"""Per-tap / per-column error of the weights-only mode of dd_conv3x3_bwd against a float64 reference (debug aid)."""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepdenoiser_amd import _lib as L  # noqa: E402

lib = L.load()
cin, cout, H, B = [int(v) for v in (sys.argv[1:5] + ["64", "64", "32", "2"][len(sys.argv) - 1:])]
torch.manual_seed(0)
x = torch.relu(torch.randn(B, H, H, cin, device="cuda")).bfloat16()
dy = torch.randn(B, H, H, cout, device="cuda").bfloat16()
dw = torch.zeros(9, cin, cout, device="cuda")
db = torch.zeros(cout, device="cuda")
a = L.ConvBwdArgs()
C.memset(C.byref(a), 0, C.sizeof(a))
a.dy, a.x, a.dw, a.db = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr()
a.ld_dy, a.ld_x, a.cout, a.cin, a.B, a.H, a.W, a.dtype = cout, cin, cout, cin, B, H, H, L.DD_BF16
L.check(lib.dd_conv3x3_bwd(C.byref(a), torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
xd, dyd = x.double().permute(0, 3, 1, 2), dy.double().permute(0, 3, 1, 2)
xp = F.pad(xd, (1, 1, 1, 1))
ref = torch.zeros(9, cin, cout, dtype=torch.float64, device="cuda")
for ty in range(3):
    for tx in range(3):
        ref[ty * 3 + tx] = torch.einsum("bchw,bdhw->cd", xp[:, :, ty:ty + H, tx:tx + H], dyd)
got = dw.double()
for t in range(9):
    e = (got[t] - ref[t]).norm() / ref[t].norm()
    col = ((got[t] - ref[t]).norm(dim=0) / ref[t].norm(dim=0))
    print("tap %d: rel %.2e   per-co (first 20): %s" % (t, e, " ".join("%.0e" % v for v in col[:20].tolist())))
print("db rel", float((db.double() - dyd.sum((0, 2, 3))).norm() / dyd.sum((0, 2, 3)).norm()))
