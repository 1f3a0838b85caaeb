import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from deepdenoiser_amd import configs
from deepdenoiser_amd.architecture import Architecture
from test_gpu_model import _inputs
from oracle.model import OracleArchitecture
aj = configs.cfg3_tiramisu(filters=(16, 24, 32), convs=2)
o = OracleArchitecture(aj, seed=2)
feats, labels = _inputs(o, 1, 32, 32)
o.predict(feats)
res = {}
for ks in ("0", "1"):
    os.environ["DD_CONV_KS"] = ks
    os.environ["DD_CONVT3_PARITY"] = ks
    arch = Architecture(aj, device="cuda", dtype="bf16")
    prog = arch.program(1, 32, 32)
    arch.params.load_list(list(o.vs.vars.values()))
    prog.set_inputs({k: v.cuda() for k, v in feats.items()})
    for rep in range(3):
        prog.forward()
        torch.cuda.synchronize()
        bufs = []
        seen = set()
        for t in prog.core_outputs:
            if id(t.buf) not in seen:
                seen.add(id(t.buf)); bufs.append(t.buf.float().cpu().clone())
        res.setdefault(ks, []).append(bufs)
for lvl in range(len(res["0"][0])):
    a = res["0"][0][lvl]
    for rep in range(3):
        b = res["1"][rep][lvl]
        d = (a - b).abs().amax(dim=(0, 1, 2))
        bad = [(c, float(d[c])) for c in range(d.numel()) if not (d[c] < 0.05 * (a[..., c].abs().max() + 1e-3))]
        print("buffer", lvl, tuple(a.shape), "rep", rep, "finite", bool(torch.isfinite(b).all()), "channels off:", [c for c, _ in bad][:40], "max", max([v for _, v in bad] or [0]))
