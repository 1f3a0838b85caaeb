"""Which engine.Graph.conv_records entries belong to no timed launch?  (3 166.1 vs 3 111.7 GFLOP)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepdenoiser_amd import configs
from deepdenoiser_amd.architecture import Architecture
from deepdenoiser_amd.training import Trainer
from bench import synthetic_inputs

B = int(os.environ.get("B", "8"))
arch = Architecture(configs.cfg2_unet_kpcn(), device="cuda:0", dtype="bf16", seed=2)
tr = Trainer(arch, configs.bench_training(), B, 128, 128, world_size=1, use_graph=False)
f, l = synthetic_inputs(arch, B, 128, 128, "cuda:0", 1)
tr.program.set_inputs(f, l)
for _ in range(2):
    tr.step()
torch.cuda.synchronize()
g = tr.program.g
ops = list(g.pack_ops) + list(tr.program.label_ops) + list(g.fwd_ops) + list(g.bwd_ops)
infos = [getattr(o, "info", None) for o in ops]
timed = sum(i["flops"] for o, i in zip(ops, infos) if i and getattr(o, "tag", "") == "conv_igemm")
print("records", len(g.conv_records), sum(r["flops"] for r in g.conv_records) / 1e9, "timed", timed / 1e9)
ids = set(id(i) for i in infos if i)
import collections
cnt = collections.Counter()
for r in g.conv_records:
    key = (r["B"], r["H"], r["taps"], r["k"], r["n"], r.get("extra_reads"))
    cnt[key] += 1
for o, i in zip(ops, infos):
    if i and getattr(o, "tag", "") == "conv_igemm":
        key = (i["B"], i["H"], i["taps"], i["k"], i["n"], i.get("extra_reads"))
        cnt[key] -= 1
for k, v in cnt.items():
    if v:
        print("unmatched", k, v)
