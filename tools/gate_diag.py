"""Where the half-precision gradient of a small network leaves the storage-emulating oracle:
    python tools/gate_diag.py [case] [dtype]
Per parameter tensor: |g_emulated|, how far storage rounding alone moves it (|g_emulated - g_plain| / |g_emulated|), the device's distance from
the emulation and from the plain oracle; plus the forward errors per prediction (a rounding flip in the forward decorrelates everything
downstream of it).  Run with DD_FUSE_COMPOSE_BWD=0 / DD_COMPOSE_STREAM_BWD=0 to see whether the compose kernel matters."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from test_gpu_model import CASES, _inputs, _with_flags  # noqa: E402
from test_gpu_round3 import _emulated_step  # noqa: E402
from gpu_util import rel_l2  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402
from oracle import training as OT  # noqa: E402
from oracle.model import OracleArchitecture  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "tiramisu_multiscale"
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
aj, B, H, W = CASES[case]
tj = configs.bench_training() if len(aj["combined_features"]) == 1 else configs.training()
plain = OracleArchitecture(aj, dtype=torch.float64, seed=2)
feats, labels = _inputs(plain, B, H, W)
feats = _with_flags(aj, feats, B, H, W)
arch = Architecture(aj, device="cuda", dtype=dtype)
prog = arch.program(B, H, W, training_json=tj)
oracle, preds_o, loss_o, grads_o = _emulated_step(aj, tj, dtype, feats, labels, prog.loss_scale)
grads_p = OT.train_step(plain, aj, tj, feats, labels, ([], []), 1)[1]
arch.params.load_list(list(oracle.vs.vars.values()))
dev, devl = {k: v.cuda() for k, v in feats.items()}, {k: v.cuda() for k, v in labels.items()}
loss = float(prog.train_step(dev, devl))
torch.cuda.synchronize()
for s, (dp, do) in enumerate(zip(prog.prediction_dictionaries(), preds_o)):
    for k in do:
        print("forward scale %d %-28s rel-L2 vs emulation %.3e" % (s, k, rel_l2(dp[k].cpu(), do[k])))
print("loss %.8f emulated %.8f" % (loss, loss_o))
print("%-58s %10s %12s %12s %12s" % ("tensor", "|g_emu|", "emu-plain", "dev-emu", "dev-plain"))
for p, go, gp in zip(arch.params.params, grads_o, grads_p):
    got = arch.params.grad(p).double().cpu() / prog.loss_scale
    n = float(go.norm())
    if n == 0:
        continue
    print("%-58s %10.3e %12.3e %12.3e %12.3e" % (p.name, n, float((go - gp).norm()) / n, float((got - go).norm()) / n, float((got - gp).norm()) / float(gp.norm())))
