"""Generates tests/golden/cfg1_golden.npz: seeded inputs and float64 oracle outputs of BASELINE config 1 (small U-Net, 3-channel noisy RGB,
64x64 tile, direct prediction).  The oracle is the CPU restatement of the reference's TF-1.x graph (TensorFlow itself cannot be imported
here, so these vectors pin the RESTATEMENT, not live TF: parity stays "unpinned" in the sense of DESIGN.md).  Run from the repo root:
    python tests/golden/make_cfg1_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deepdenoiser_amd import configs          # noqa: E402
from deepdenoiser_amd.naming import Naming    # noqa: E402
from oracle import training as OT             # noqa: E402
from oracle.model import OracleArchitecture   # noqa: E402


def main():
    aj, tj = configs.cfg1_small_unet(), configs.bench_training()
    B, H, W = 1, 64, 64
    oracle = OracleArchitecture(aj, dtype=torch.float64, seed=2)
    g = torch.Generator().manual_seed(11)
    feats, labels = {}, {}
    for f in oracle.features + oracle.auxiliary:
        feats[Naming.source_feature_name(f.name, index=0)] = (torch.randn(B, H, W, f.channels, generator=g).abs()
                                                              * torch.exp(0.5 * torch.randn(B, H, W, 1, generator=g))).float()
    for f in oracle.features:
        labels[Naming.target_feature_name(f.name)] = torch.randn(B, H, W, f.channels, generator=g).abs().float()
    preds = oracle.predict(feats)
    loss, grads = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
    out = {"loss": np.float64(float(loss))}
    for k, v in feats.items():
        out["in/" + k] = v.numpy()
    for k, v in labels.items():
        out["label/" + k] = v.numpy()
    for s, d in enumerate(preds):
        for k, v in d.items():
            out["pred/%d/%s" % (s, k)] = v.detach().numpy().astype(np.float32)
    names = list(oracle.vs.vars.keys())
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([float(gr.norm()) for gr in grads])
    out["grad_first"] = grads[0].detach().numpy().astype(np.float32)          # the first conv kernel's full gradient
    out["grad_last"] = grads[-1].detach().numpy().astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "cfg1_golden.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; loss", float(loss))


if __name__ == "__main__":
    main()
