"""-m gpu: DD_DETERMINISTIC=1 (csrc/dd_common.h: the workgroups of every launch that ends in fp32 atomics flush in index order).  The library reads
the switch once per process, so the check runs in a child: forward + loss + full backward of the cfg-2 network (every gradient kernel of the bench
step: fused 3x3 backward, weight-gradient role, transposed conv, fused head, streaming compose net, loss head) and of a small Tiramisu (dense-block
and 1x1 GEMM-tile weight gradients, transposed-conv filter gradient, column sums) twice from the same state -- the two gradient arenas and losses must
be BIT-identical.  Without the switch they differ in the last bits (printed, not asserted: a lucky run may agree)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time, torch
sys.path.insert(0, %r)
from deepdenoiser_amd import configs
from deepdenoiser_amd.architecture import Architecture
from bench import synthetic_inputs
for name, aj, B, H in (("cfg2", configs.cfg2_unet_kpcn(), 8, 128), ("tiramisu", configs.cfg3_tiramisu(filters=(16, 24, 32), convs=2), 2, 64)):
    arch = Architecture(aj, device="cuda:0", dtype="bf16", seed=2)
    prog = arch.program(B, H, H, training_json=configs.bench_training())
    feats, labels = synthetic_inputs(arch, B, H, H, "cuda:0", 3)
    prog.set_inputs(feats, labels)
    runs = []
    for i in range(3):
        prog.zero_grads(); prog.forward(pack=True); prog.backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        prog.zero_grads(); prog.forward(pack=True); prog.backward()
        torch.cuda.synchronize()
        runs.append((arch.params.grads.clone(), prog.loss_buf.clone(), time.perf_counter() - t0))
    same = all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    nz = int((runs[0][0] != 0).sum())
    print("RESULT %%s identical=%%d nonzero=%%d ms=%%.1f maxdiff=%%.3e" %% (name, same, nz, 1e3 * runs[-1][2], float((runs[0][0] - runs[1][0]).abs().max())))
"""


def _run(det):
    env = dict(os.environ)
    env.pop("DD_DETERMINISTIC", None)
    if det:
        env["DD_DETERMINISTIC"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return {ln.split()[1]: dict(kv.split("=") for kv in ln.split()[2:]) for ln in p.stdout.splitlines() if ln.startswith("RESULT")}


@pytest.mark.gpu
def test_deterministic_mode_makes_the_gradients_bit_reproducible():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    det, plain = _run(True), _run(False)
    print("deterministic:", det, " default:", plain)
    for name in ("cfg2", "tiramisu"):
        assert det[name]["identical"] == "1", (name, det[name])
        assert int(det[name]["nonzero"]) > 1000
