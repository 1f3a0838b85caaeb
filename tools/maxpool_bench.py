"""Times dd_maxpool_fwd / dd_maxpool_bwd at the shapes of the bench step (set DD_MAXPOOL_GENERIC=1 for the generic kernels)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepdenoiser_amd import _lib as L
lib = L.load()
dev = "cuda"
def run(B, H, W, C, pool, acc):
    OH, OW = -(-H // 2), -(-W // 2)
    x = torch.relu(torch.randn(B, H, W, C, device=dev)).bfloat16()
    y = torch.empty(B, OH, OW, C, device=dev, dtype=torch.bfloat16)
    idx = torch.zeros(B, OH, OW, C, device=dev, dtype=torch.uint8)
    gy = torch.randn_like(y); gx = torch.zeros_like(x)
    s = torch.cuda.current_stream().cuda_stream
    def f(): L.check(lib.dd_maxpool_fwd(x.data_ptr(), C, y.data_ptr(), C, idx.data_ptr(), C, B, H, W, pool, 2, 1, L.DD_BF16, s))
    def b(): L.check(lib.dd_maxpool_bwd(gy.data_ptr(), C, idx.data_ptr(), gx.data_ptr(), C, None, 0, C, B, H, W, pool, 2, acc, L.DD_BF16, s))
    out = []
    for fn in (f, b):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20 * 1e3)
    nb_f = B * H * W * C * 2 + B * OH * OW * C * 3
    nb_b = B * OH * OW * C * 3 + B * H * W * C * 2 * (2 if acc else 1)
    print("B%d %dx%dx%d pool %d acc %d: fwd %.1f us (%.2f TB/s)  bwd %.1f us (%.2f TB/s)" % (B, H, W, C, pool, acc, out[0], nb_f / out[0] / 1e6, out[1], nb_b / out[1] / 1e6))
run(128, 128, 128, 64, 3, 1); run(128, 64, 64, 96, 3, 1); run(209, 128, 128, 64, 3, 0); run(8, 256, 256, 112, 2, 1)
