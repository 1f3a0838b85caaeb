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
    preds_o, internals = oracle.predict(feats, return_internals=True)
    loss_o = OT.model_loss(oracle, aj, tj, preds_o, labels)
    leaves = [(s, k, v) for s, d in enumerate(preds_o) for k, v in d.items() if v.requires_grad]
    gp = torch.autograd.grad(loss_o, [v for _, _, v in leaves], retain_graph=True, allow_unused=True)
    params = oracle.parameters()
    gw = torch.autograd.grad(loss_o, params, allow_unused=True, retain_graph=True)
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
    for i, hid in enumerate(prog.post_hidden):
        gh = hid.grad().buf.double()
        print("hidden[%d] %s: |act| %.3e frac>0 %.3f  |grad| %.3e  nonzero grad frac %.3f  written=%s" % (
            i, tuple(hid.buf.shape), float(hid.buf.double().norm()), float((hid.buf > 0).double().mean()), float(gh.norm()),
            float((gh != 0).double().mean()), hid.grad_written))
    for i, ct in enumerate(prog.core_outputs):
        gc = ct.grad().buf.double()
        print("core[%d] %s ch0=%d C=%d: |grad| %.3e nonzero frac %.3f; per-image grad norms %s" % (
            i, tuple(ct.buf.shape), ct.ch0, ct.C, float(gc.norm()), float((gc != 0).double().mean()),
            [round(float(gc[b].norm()), 4) for b in range(gc.shape[0])]))
    T = len(oracle.tuples)
    n_scales = len(prog.post)
    for s in range(n_scales):
        # oracle: post[t] is largest-first per tuple; core_outputs[t] is coarsest-first
        lo = [internals["post"][t][s] for t in range(T)]
        g_lo = torch.autograd.grad(loss_o, lo, retain_graph=True)
        got = prog.post[s].grad().buf[..., :prog.post[s].C]
        print("dlogits scale %d rel %.3e" % (s, rel(got, torch.cat(g_lo, 0))))
        co = [internals["core_outputs"][t][n_scales - 1 - s] for t in range(T)]
        g_co = torch.autograd.grad(loss_o, co, retain_graph=True)
        ct = prog.core_outputs[n_scales - 1 - s]
        print("dcore  scale %d rel %.3e   (C=%d relu=%s)" % (s, rel(ct.grad().buf[..., ct.ch0:ct.ch0 + ct.C], torch.cat(g_co, 0) * (torch.cat(co, 0) > 0)), ct.C, ct.relu))
    for p, n, g in zip(arch.params.params, oracle.vs.vars.keys(), gw):
        if g is None:
            continue
        print("dparam %-50s rel %.3e  |g| %.3e" % (n, rel(arch.params.grad(p), g), float(g.norm())))


if __name__ == "__main__":
    main(sys.argv[1])
