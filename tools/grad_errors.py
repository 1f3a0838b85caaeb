"""Per-parameter gradient error of one training step against the f64 oracle (cfg-2, 128x128, B=1): python tools/grad_errors.py [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_round2 import _pair  # noqa: E402
from gpu_util import rel_l2  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from oracle import training as OT  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
aj, tj, B, H, W = configs.cfg2_unet_kpcn(), configs.bench_training(), 1, 128, 128
oracle, arch, prog, feats, labels, dev, devl, preds_o = _pair(aj, dtype, B, H, W, tj)
arch.predict(dev)
loss_o, grads_o = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
prog.train_step(dev, devl)
torch.cuda.synchronize()
rows = []
for p, go in zip(arch.params.params, grads_o):
    if float(go.norm()) > 0:
        rows.append((rel_l2(arch.params.grad(p).cpu() / prog.loss_scale, go), p.name, tuple(go.shape), float(go.abs().max())))
rows.sort()
print("median %.3e" % rows[len(rows) // 2][0])
for e, n, sh, mx in rows:
    print("%.3e  %-60s %-18s max|g| %.2e" % (e, n, sh, mx))
