"""BASELINE config 1 (small U-Net, 3-channel RGB, 64x64) against the committed golden vectors (tests/golden/cfg1_golden.npz, made by
tests/golden/make_cfg1_golden.py from the float64 oracle): the oracle must keep reproducing them (CPU), the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest
import torch

from deepdenoiser_amd import configs
from oracle import training as OT
from oracle.model import OracleArchitecture

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_golden.npz"))


def _dicts():
    feats = {k[3:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("in/")}
    labels = {k[6:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("label/")}
    return feats, labels


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def test_oracle_reproduces_golden_vectors():
    aj, tj = configs.cfg1_small_unet(), configs.bench_training()
    feats, labels = _dicts()
    oracle = OracleArchitecture(aj, dtype=torch.float64, seed=2)
    preds = oracle.predict(feats)
    for k in G.files:
        if k.startswith("pred/"):
            _, s, name = k.split("/", 2)
            assert _rel(preds[int(s)][name].detach().numpy(), G[k]) < 1e-6, k          # stored as float32
    loss, grads = OT.train_step(oracle, aj, tj, feats, labels, ([], []), 1)
    assert abs(float(loss) - float(G["loss"])) < 1e-12
    assert list(oracle.vs.vars.keys()) == list(G["grad_names"])
    assert np.allclose([float(g.norm()) for g in grads], G["grad_norms"], rtol=1e-10)


@pytest.mark.gpu
def test_hip_path_matches_golden_vectors():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from deepdenoiser_amd.architecture import Architecture
    aj, tj = configs.cfg1_small_unet(), configs.bench_training()
    feats, labels = _dicts()
    oracle = OracleArchitecture(aj, dtype=torch.float64, seed=2)
    oracle.predict(feats)                                   # creates the (seeded) variables the golden run used
    arch = Architecture(aj, device="cuda", dtype="f32")
    prog = arch.program(1, 64, 64, training_json=tj)
    arch.params.load_list(list(oracle.vs.vars.values()))
    dev = {k: v.cuda() for k, v in feats.items()}
    preds = arch.predict(dev)
    torch.cuda.synchronize()
    for k in G.files:
        if k.startswith("pred/"):
            _, s, name = k.split("/", 2)
            assert _rel(preds[int(s)][name].cpu().numpy(), G[k]) < 1e-4, k      # the fp32 gate of BASELINE.json
    loss = prog.train_step(dev, {k: v.cuda() for k, v in labels.items()})
    torch.cuda.synchronize()
    assert abs(float(loss) - float(G["loss"])) < 2e-5 * float(G["loss"])
    p = arch.params.params
    assert _rel(arch.params.grad(p[0]).cpu().numpy(), G["grad_first"]) < 5e-3
    assert _rel(arch.params.grad(p[-1]).cpu().numpy(), G["grad_last"]) < 5e-3
    norms = np.array([float(arch.params.grad(q).norm()) for q in p])
    assert np.allclose(norms, G["grad_norms"], rtol=5e-3)
