"""GPU-side debugging aid: compares HIP-path gradients at the predictions / parameters with oracle autograd.
    python tools/debug_grads.py <case name from tests/test_gpu_model.CASES>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from deepdenoiser_amd import configs                      # noqa: E402
from deepdenoiser_amd.naming import Naming                # noqa: E402
from oracle import training as OT                         # noqa: E402
import test_gpu_model as TM                               # noqa: E402


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def main(case):
    aj, B, H, W = TM.CASES[case]
    tj = configs.bench_training() if len(aj["combined_features"]) == 1 else configs.training()
    oracle, arch, prog, feats, labels, dev, devl, _ = TM._pair(aj, "f32", B, H, W, tj)
    preds_o = oracle.predict(feats)
    loss_o = OT.model_loss(oracle, aj, tj, preds_o, labels)
    leaves = [(s, k, v) for s, d in enumerate(preds_o) for k, v in d.items() if v.requires_grad]
    gp = torch.autograd.grad(loss_o, [v for _, _, v in leaves], retain_graph=True, allow_unused=True)
    params = oracle.parameters()
    gw = torch.autograd.grad(loss_o, params, allow_unused=True)
    prog.set_inputs(dev, devl)
    prog.zero_grads()
    prog.forward()
    prog.backward()
    torch.cuda.synchronize()
    print("loss", float(prog.loss_buf), float(loss_o))
    for (s, k, v), g in zip(leaves, gp):
        name = k.split("/", 1)[1]
        f = next(x for x in prog.head if x.name == name)
        if not f.load_data or g is None:
            continue
        i = prog.head_index[name]
        # gradient at the (inverted) prediction = what the loss head wrote
        # locate the tensor the loss wrote into: predictions[s].grad()
        got = prog.predictions[s].grad().buf[i * B:(i + 1) * B][..., :f.number_of_channels]
        print("dpred scale %d %-24s rel %.3e  |g| %.3e" % (s, name, rel(got, g), float(g.norm())))
    for p, n, g in zip(arch.params.params, oracle.vs.vars.keys(), gw):
        if g is None:
            continue
        print("dparam %-50s rel %.3e  |g| %.3e" % (n, rel(arch.params.grad(p), g), float(g.norm())))


if __name__ == "__main__":
    main(sys.argv[1])
