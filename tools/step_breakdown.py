"""Per-launch breakdown of one training step (HIP events): which layers cost what, at what TFLOP/s.
    python tools/step_breakdown.py [bf16|f32] [B] [cfg2|cfg3|cfg3h|cfg1]      (cfg3h: the heavy Tiramisu, F = [64, 96, 128])"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
aj = {"cfg2": configs.cfg2_unet_kpcn, "cfg3": configs.cfg3_tiramisu, "cfg3h": lambda: configs.cfg3_tiramisu(filters=(64, 96, 128)),
      "cfg1": configs.cfg1_small_unet}[cfg]()
tj = configs.bench_training()
T = {"cfg2": 128, "cfg3": 256, "cfg3h": 256, "cfg1": 64}[cfg]
arch = Architecture(aj, device="cuda", dtype=dtype, seed=2)
prog = arch.program(B, T, T, training_json=tj)
feats, labels = synthetic_inputs(arch, B, T, T, "cuda", 1)
prog.set_inputs(feats, labels)
for _ in range(2):
    prog.train_step()
fam, ops = prog.profile_ops(repeats=3, detail=True)
agg = {}
for tag, info, us in ops:
    if info is None:
        key = (tag, "-")
        fl = 0.0
    else:
        key = (tag, "%4dx%-4d k=%-3d -> n=%-3d taps=%d%s" % (info["H"], info["W"], info.get("k", info.get("m")), info["n"], info["taps"],
                                                            " flags=%d" % info["flags"] if "flags" in info else ""))
        fl = info["flops"]
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += us; a[2] += fl
tot = sum(a[1] for a in agg.values())
print("%-12s %-48s %5s %9s %8s %8s" % ("family", "shape", "n", "total_us", "avg_us", "TF/s"))
for (tag, shape), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if a[1] < 0.004 * tot:
        continue
    print("%-12s %-48s %5d %9.1f %8.1f %8.1f" % (tag, shape, a[0], a[1], a[1] / a[0], a[2] / a[1] / 1e6 if a[1] else 0))
print("total %.1f us;" % tot, {k: (v[0], round(v[1], 3)) for k, v in fam.items()})
if "ops" in sys.argv:
    print("non-conv launches in program order:")
    for tag, info, us in ops:
        pass
    all_ops = list(prog.g.pack_ops) + list(prog.g.fwd_ops) + list(prog.g.bwd_ops)
    for op, (tag, info, us) in zip(all_ops, ops):
        if not tag.startswith("conv"):
            print("  %-16s %-60s %8.1f us" % (tag, getattr(op, "origin", getattr(op, "__qualname__", "?")), us))
