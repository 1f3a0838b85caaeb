"""Training soak: N optimisation steps of a configuration on fixed synthetic inputs under hipGraph replay; prints the loss trajectory (it must fall
and stay finite -- a kernel racing with its neighbours under replay shows up here before it shows up anywhere else).
    python tools/soak.py [cfg2|cfg3|cfg3h] [steps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402
from deepdenoiser_amd.training import Trainer  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
aj, B, T = {"cfg2": (configs.cfg2_unet_kpcn(), 32, 128), "cfg3": (configs.cfg3_tiramisu(), 4, 256),
            "cfg3h": (configs.cfg3_tiramisu(filters=(64, 96, 128)), 2, 256)}[cfg]
arch = Architecture(aj, device="cuda", dtype="bf16", seed=2)
tr = Trainer(arch, configs.bench_training(), B, T, T)
f, l = synthetic_inputs(arch, B, T, T, "cuda", 1)
tr.program.set_inputs(f, l)
losses = []
for s in range(steps):
    lb = tr.step()
    if s % max(1, steps // 10) == 0 or s == steps - 1:
        torch.cuda.synchronize()
        losses.append((s, float(lb.sum())))
print(cfg, "B=%d" % B, " ".join("%d:%.4f" % x for x in losses))
ok = all(torch.isfinite(torch.tensor(v)) for _, v in losses) and losses[-1][1] < losses[0][1]
print("finite and falling:", bool(ok))
sys.exit(0 if ok else 1)
