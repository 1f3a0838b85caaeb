"""Times dd_compose_net_fwd directly (no Program): inference form (nothing saved) and training form (activations saved), several shapes.
    [DD_LIB=tools/exp/libdd_<variant>.so] [DD_COMPOSE_STREAM=0] python tools/compose_stream_bench.py [dtype]
Prints us per launch, ps per pixel and the fraction of the bf16 MFMA peak the four 3x3 layers' algorithmic flops reach."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepdenoiser_amd import _lib as L  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
code = {"bf16": L.DD_BF16, "f16": L.DD_F16}[dtype]
tdt = {"bf16": torch.bfloat16, "f16": torch.float16}[dtype]
lib = L.load()
dev = "cuda"
torch.manual_seed(0)
w_in = torch.randn(6, 24, device=dev) * 0.3
b_in = torch.randn(24, device=dev) * 0.1
w_res = [torch.randn(3, 3, 24, 24, device=dev) * 0.08 for _ in range(4)]
b_res = [torch.randn(24, device=dev) * 0.1 for _ in range(4)]
w_out = torch.randn(24, device=dev) * 0.3
b_out = torch.randn(1, device=dev) * 0.1


def run(N, H, W, save, reps=20):
    small = torch.randn(N, H // 2, W // 2, 3, device=dev)
    fine = torch.randn(N, H, W, 3, device=dev)
    out = torch.zeros(N, H, W, 3, device=dev)
    a = L.ComposeArgs()
    a.small, a.ld_small, a.fine, a.ld_fine, a.out, a.ld_out = small.data_ptr(), 3, fine.data_ptr(), 3, out.data_ptr(), 3
    a.w_in, a.b_in, a.w_out, a.b_out = w_in.data_ptr(), b_in.data_ptr(), w_out.data_ptr(), b_out.data_ptr()
    for i in range(4):
        a.w_res[i], a.b_res[i] = w_res[i].data_ptr(), b_res[i].data_ptr()
    keep = []
    if save:
        for i in range(5):
            t = torch.zeros(N, H, W, 24, device=dev, dtype=tdt)
            keep.append(t)
            a.save_act[i], a.ld_act[i] = t.data_ptr(), 24
        wl = torch.zeros(N, H, W, 1, device=dev, dtype=tdt)
        keep.append(wl)
        a.save_wl, a.ld_wl = wl.data_ptr(), 1
    a.N, a.H, a.W, a.dtype = N, H, W, code
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        L.check(lib.dd_compose_net_fwd(C.byref(a), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.check(lib.dd_compose_net_fwd(C.byref(a), s))
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    px = N * H * W
    flop = px * 4 * 9 * 24 * 24 * 2
    print("%-28s %s  %8.1f us  %6.1f ps/px  %5.1f %% of 2.5 PFLOP/s" % ((N, H, W), "save" if save else "infer", us, us * 1e6 / px, 100 * flop / (us * 1e-6) / 2.5e15))
    return out


shapes = [(128, 128, 128), (128, 64, 64), (209, 128, 128), (209, 64, 64), (53, 128, 128), (8, 256, 256)]
if os.environ.get("CS_SHAPES"):
    shapes = [tuple(int(v) for v in t.split(",")) for t in os.environ["CS_SHAPES"].split(";")]
for shape in shapes:
    for save in (False, True):
        run(*shape, save)
