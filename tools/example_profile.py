"""Per-family and per-launch times of the literal ArchitectureExample.json / TrainingExample.json step (17 tuple passes per tile, B tiles):
    python tools/example_profile.py [B]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs
from deepdenoiser_amd import configs
from deepdenoiser_amd.architecture import Architecture
from deepdenoiser_amd.training import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
arch = Architecture(configs.example_architecture(), device="cuda", dtype="bf16", seed=2)
tr = Trainer(arch, configs.training(), B, 128, 128, world_size=1, use_graph=True)
f, l = synthetic_inputs(arch, B, 128, 128, "cuda", 1000)
tr.program.set_inputs(f, l)
for _ in range(3): tr.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tr.step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("%.3f ms/step  %.1f tiles/s  %.0f tuple passes/s" % (dt * 1e3, B / dt, B * tr.program.T / dt))
fam, detail = tr.program.profile_ops(repeats=3, detail=True)
for k, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print("%-16s %3d launches %8.3f ms" % (k, n, ms))
if "--detail" in sys.argv:
    for tag, info, us in detail:
        if us > 15: print("  %-16s %8.1f us" % (tag, us))
