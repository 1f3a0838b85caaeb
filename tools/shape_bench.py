import os, sys, torch
sys.path.insert(0, "/root/repo")
from deepdenoiser_amd.engine import Graph
def timeit(fn, iters=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for H, W, B in ((128, 128, 128), (64, 256, 128), (256, 64, 128), (512, 512, 8), (32, 32, 2048), (16, 16, 8192), (1024, 2048, 1)):
    g = Graph("cuda", "bf16")
    x = g.tensor(B, H, W, 64, relu=True, requires_grad=False); x.buf.normal_()
    lay = g.layer("b/conv2d", 3, 64, 64)
    g.conv(x, lay, relu=True); g.finalize(); s = g.stream_ptr(); g.run(g.pack_ops)
    t = timeit(lambda: g.fwd_ops[-1](s))
    print("64->64 %5dx%-5d B=%-5d (%.1f Mpx): %7.1f us %7.1f TF/s" % (H, W, B, B*H*W/1e6, t, 2.0*B*H*W*9*64*64/t/1e6), flush=True)
