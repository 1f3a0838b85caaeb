import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepdenoiser_amd import configs
from oracle import training as OT
import test_gpu_model as TM

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))

def run(tag, aj, B, H, W):
    tj = configs.bench_training() if len(aj["combined_features"]) == 1 else configs.training()
    oracle, arch, prog, feats, labels, dev, devl, _ = TM._pair(aj, "f32", B, H, W, tj)
    preds_o = oracle.predict(feats)
    loss_o = OT.model_loss(oracle, aj, tj, preds_o, labels)
    gw = torch.autograd.grad(loss_o, oracle.parameters(), allow_unused=True)
    prog.set_inputs(dev, devl); prog.zero_grads(); prog.forward(); prog.backward(); torch.cuda.synchronize()
    errs = {n: rel(arch.params.grad(p), g) for p, n, g in zip(arch.params.params, oracle.vs.vars.keys(), gw) if g is not None and float(g.norm()) > 0}
    worst = max(errs, key=errs.get)
    comp = {k.split("/")[-2]: round(v, 5) for k, v in errs.items() if "compose" in k and k.endswith("kernel")}
    print("%-40s loss %.5f/%.5f worst %s %.2e | compose %s" % (tag, float(prog.loss_buf), float(loss_o), worst.replace("reused_", ""), errs[worst], comp), flush=True)

base = dict(tuple_type="COMBINED", filters=(16, 16), convs=1, kernel_size=3, flag_mode="NONE")
def A(**kw):
    d = dict(base); d.update(kw); return configs.architecture(**d)
run("base 1x16x32", A(), 1, 16, 32)
run("SINGLE", A(tuple_type="SINGLE"), 1, 16, 32)
run("ks5", A(kernel_size=5), 1, 16, 32)
run("32x32", A(), 1, 32, 32)
run("B2", A(), 2, 16, 32)
run("convs2", A(convs=2), 1, 16, 32)
run("3 levels", A(filters=(16, 24, 32)), 1, 16, 32)
run("16x16", A(), 1, 16, 16)
run("32x16", A(), 1, 32, 16)
one = {"Diffuse": {"Color": "Diffuse Color", "Direct": "Diffuse Direct", "Indirect": "Diffuse Indirect"}}
run("one combined", A(combined=one), 1, 16, 32)
run("one combined SINGLE", A(combined=one, tuple_type="SINGLE"), 1, 16, 32)
