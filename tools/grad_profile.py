"""Per-parameter gradient error of a storage type against the f64 oracle (one training step, same weights).
    python tools/grad_profile.py <dtype> <B> <H> <W> [filters e.g. 16,16,24] [convs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402
from oracle import training as OT  # noqa: E402
from oracle.model import OracleArchitecture  # noqa: E402
from test_gpu_model import _inputs  # noqa: E402

dtype, B, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
filters = tuple(int(v) for v in sys.argv[5].split(",")) if len(sys.argv) > 5 else (16, 16, 24)
convs = int(sys.argv[6]) if len(sys.argv) > 6 else 1
aj = configs.architecture(filters=filters, convs=convs, flag_mode="NONE",
                          combined={"Diffuse": {"Color": "Diffuse Color", "Direct": "Diffuse Direct", "Indirect": "Diffuse Indirect"}})
tj = configs.training(image_mean=0.0)
tj["combined_image_training_settings"]["statistics"]["track_mean"] = False
oracle = OracleArchitecture(aj, dtype=torch.float64, seed=2)
feats, labels = _inputs(oracle, B, H, W)
oracle.predict(feats)
arch = Architecture(aj, device="cuda", dtype=dtype)
prog = arch.program(B, H, W, training_json=tj)
arch.params.load_list(list(oracle.vs.vars.values()))
loss = float(prog.train_step({k: v.cuda() for k, v in feats.items()}, {k: v.cuda() for k, v in labels.items()}))
torch.cuda.synchronize()
loss_o, grads_o = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
print("loss %.6f oracle %.6f" % (loss, float(loss_o)))
tot_e = tot_n = 0.0
for p, go in zip(arch.params.params, grads_o):
    g = arch.params.grad(p).double().cpu() / prog.loss_scale
    e, n = float((g - go).norm()), float(go.norm())
    tot_e += e * e
    tot_n += n * n
    print("%-60s |g| %.3e  rel-L2 %.3e" % (p.name, n, e / max(n, 1e-30)))
print("whole gradient rel-L2 %.3e" % ((tot_e / tot_n) ** 0.5))
