"""Per-launch records (family, shape, us, fraction of the bf16 MFMA peak) of one training step of BASELINE config 3:
    python tools/cfg3_launches.py [light|heavy] [B] [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402
from deepdenoiser_amd.training import Trainer  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "heavy"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
filters = (64, 96, 128) if which == "heavy" else (16, 24, 32)
arch = Architecture(configs.cfg3_tiramisu(filters=filters, convs=4), device="cuda", dtype="bf16", seed=2)
tr = Trainer(arch, configs.bench_training(), B, 256, 256, use_graph=False)
f, l = synthetic_inputs(arch, B, 256, 256, "cuda", 1)
tr.program.set_inputs(f, l)
for _ in range(2):
    tr.step()
torch.cuda.synchronize()
times, launches = tr.program.profile_ops(detail=True)
rows = []
agg = {}
for tag, info, us in launches:
    r = {"family": tag, "us": round(us, 1)}
    if info:
        r.update(info)
        if "flops" in info:
            r["frac"] = round(info["flops"] / us / 1e6 / 2500, 3)
    rows.append(r)
    key = (tag,) + tuple((k, r.get(k)) for k in ("H", "taps", "k", "n", "m", "extra_reads", "flags", "backward"))
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1; a[1] += us; a[2] += (info or {}).get("flops", 0)
tot = sum(us for _, _, us in launches)
print("%s B=%d: %d launches, %.2f ms" % (which, B, len(launches), tot / 1e3))
for key, (n, us, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%7.1f us x%2d  frac %.3f  %s" % (us / n, n, fl / us / 1e6 / 2500 if fl else 0, " ".join("%s=%s" % kv for kv in key[1:] if kv[1] is not None) + "  " + key[0]))
if len(sys.argv) > 3:
    json.dump(rows, open(sys.argv[3], "w"), indent=0)
