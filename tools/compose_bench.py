"""Times the fused compose-net launches (csrc/dd_compose.hip) of one cfg-2 training step: forward and backward at 128^2 and 64^2.
    [DD_LIB=tools/exp/libdd_<variant>.so] python tools/compose_bench.py [B] [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
arch = Architecture(configs.cfg2_unet_kpcn(), device="cuda", dtype=dtype, seed=2)
prog = arch.program(B, 128, 128, training_json=configs.bench_training())
feats, labels = synthetic_inputs(arch, B, 128, 128, "cuda", 1)
prog.set_inputs(feats, labels)
for _ in range(2):
    prog.train_step()
_, ops = prog.profile_ops(repeats=3, detail=True)
print(os.environ.get("DD_LIB", "default"), " ".join("%.1f" % us for tag, info, us in ops if tag == "compose_net"), "(us: fwd 64^2, fwd 128^2, bwd 128^2, bwd 64^2)")
