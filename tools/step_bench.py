"""cfg-2 training step (B = 32) and the heavy Tiramisu (B = 8, 256x256), ms per step, eager launches:  python tools/f32_bench.py [f32|bf16|f16]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs
from deepdenoiser_amd import configs
from deepdenoiser_amd.architecture import Architecture
from deepdenoiser_amd.training import Trainer
DT = sys.argv[1] if len(sys.argv) > 1 else "f32"
for name, aj, B, T in (("cfg-2 B=32", configs.cfg2_unet_kpcn(), 32, 128), ("cfg-3 heavy B=8 256", configs.cfg3_tiramisu(filters=(64, 96, 128)), 8, 256)):
    arch = Architecture(aj, device="cuda", dtype=DT, seed=2)
    tr = Trainer(arch, configs.bench_training(), B, T, T)
    f, l = synthetic_inputs(arch, B, T, T, "cuda", 1)
    tr.program.set_inputs(f, l)
    for _ in range(3): tr.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tr.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("%s %s: %.2f ms/step %.1f tiles/s" % (name, DT, dt * 1e3, B / dt))
    del tr, arch; torch.cuda.empty_cache()
