import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
def t(fn, it=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3
x = torch.randn(32, 128, 128, 64, device="cuda").bfloat16(); y = torch.empty_like(x)
us = t(lambda: y.copy_(x)); print("torch copy 67MB bf16: %.1f us  %.2f TB/s (r+w)" % (us, 2 * x.numel() * 2 / us / 1e6))
us = t(lambda: torch.relu_(y)); print("torch relu_ inplace: %.1f us %.2f TB/s" % (us, 2 * x.numel() * 2 / us / 1e6))
big = torch.randn(512 * 1024 * 1024 // 4, device="cuda"); big2 = torch.empty_like(big)
us = t(lambda: big2.copy_(big), 5); print("torch copy 512MB f32: %.1f us  %.2f TB/s (r+w)" % (us, 2 * big.numel() * 4 / us / 1e6))
