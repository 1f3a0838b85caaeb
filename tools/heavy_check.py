import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from deepdenoiser_amd import configs
from oracle import training as OT
from test_gpu_model import _pair
from gpu_util import rel_l2
aj = configs.cfg3_tiramisu(filters=(64, 96, 128), convs=4)
tj = configs.bench_training()
oracle, arch, prog, feats, labels, dev, devl, preds_o = _pair(aj, "bf16", 1, 64, 64, tj)
loss_o, grads_o = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
prog.train_step(dev, devl); torch.cuda.synchronize()
errs = [(rel_l2(arch.params.grad(p).cpu(), go), p.name) for p, go in zip(arch.params.params, grads_o) if float(go.norm()) > 0]
s = sorted(e for e, _ in errs)
print(os.environ.get("DD_CONV_KS", "1"), "median %.3f max %.3f" % (s[len(s)//2], s[-1]))
print(" ".join("%.2f" % e for e, _ in errs))
