import sys, time, torch
sys.path.insert(0, '.')
from deepdenoiser_amd import configs
from deepdenoiser_amd.architecture import Architecture
from deepdenoiser_amd.training import Trainer
from bench import synthetic_inputs
for name, aj, B in [("cfg3 tiramisu F=[16,24,32]", configs.cfg3_tiramisu(filters=(16, 24, 32), convs=4), 8), ("cfg3 tiramisu F=[64,96,128] (heavy)", configs.cfg3_tiramisu(filters=(64, 96, 128)), 2), ("cfg1 small unet", configs.cfg1_small_unet(), 64)]:
    arch = Architecture(aj, device="cuda", dtype="bf16", seed=2)
    H = 64 if "cfg1" in name else 256
    tr = Trainer(arch, configs.bench_training(), B, H, H)
    f, l = synthetic_inputs(arch, B, H, H, "cuda", 1)
    tr.program.set_inputs(f, l)
    for _ in range(4): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tr.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%s: B=%d %dx%d  %.2f ms/step  %.1f tiles/s  loss %.4f  params %d" % (name, B, H, H, dt * 1e3, B / dt, float(tr.program.loss_buf), arch.params.total))
