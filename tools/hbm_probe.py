"""What this box's HBM delivers to plain streaming kernels (torch elementwise ops, 1 GiB tensors): python tools/hbm_probe.py"""
import torch


def timeit(fn, iters=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


n = 1 << 29      # 2-byte elements: 1 GiB
x = torch.randn(n, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
gb = n * 2 / 1e9
t = timeit(lambda: y.copy_(x));           print("copy   (read 1 GiB + write 1 GiB): %6.1f us  %5.2f TB/s" % (t * 1e6, 2 * gb / t / 1e3))
t = timeit(lambda: y.fill_(1.0));         print("fill   (write 1 GiB):              %6.1f us  %5.2f TB/s" % (t * 1e6, gb / t / 1e3))
t = timeit(lambda: torch.relu_(y));       print("relu_  (read + write in place):    %6.1f us  %5.2f TB/s" % (t * 1e6, 2 * gb / t / 1e3))
xs = x.view(torch.int16)
t = timeit(lambda: xs.max());             print("max    (read 1 GiB):               %6.1f us  %5.2f TB/s" % (t * 1e6, gb / t / 1e3))
for mb in (64, 128, 256, 512):
    m = mb << 19
    a, b = x[:m], y[:m]
    t = timeit(lambda: b.copy_(a), 50)
    print("copy of %4d MiB (read + write):      %6.1f us  %5.2f TB/s" % (mb, t * 1e6, 2 * m * 2 / 1e9 / t / 1e3))
