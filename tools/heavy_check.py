"""Gradient error of a half-precision Tiramisu against the plain f64 oracle under the kernel switches (is a deviation a kernel's, or the net's?):
    [DD_CONV_KS=0] [DD_DENSE_GATHER=0|1] python tools/heavy_check.py [heavy|small] [bf16|f16]"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from deepdenoiser_amd import configs
from oracle import training as OT
from test_gpu_model import _pair
from gpu_util import rel_l2
which = sys.argv[1] if len(sys.argv) > 1 else "heavy"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
aj, B, H, W = (configs.cfg3_tiramisu(filters=(64, 96, 128), convs=4), 1, 64, 64) if which == "heavy" else (configs.cfg3_tiramisu(filters=(16, 24, 32), convs=2), 1, 32, 32)
tj = configs.bench_training()
oracle, arch, prog, feats, labels, dev, devl, preds_o = _pair(aj, dtype, B, H, W, tj)
loss_o, grads_o = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
loss = float(prog.train_step(dev, devl)); torch.cuda.synchronize()
errs = [(rel_l2(arch.params.grad(p).cpu() / prog.loss_scale, go), p.name) for p, go in zip(arch.params.params, grads_o) if float(go.norm()) > 0]
s = sorted(e for e, _ in errs)
print(which, dtype, "KS=%s GATHER=%s" % (os.environ.get("DD_CONV_KS", "1"), os.environ.get("DD_DENSE_GATHER", "auto")),
      "loss err %.2e  gradient median %.3f max %.3f" % (abs(loss - float(loss_o)) / abs(float(loss_o)), s[len(s) // 2], s[-1]))
