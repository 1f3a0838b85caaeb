"""-m gpu: does training in the half-precision storage types go where f32 training goes?

The parity gate (1e-4) applies to the f32 path; the throughput path stores activations and their gradients in bf16 (fp16 for inference,
trainable with the loss scale).  Single-step gradient error of those types is measured in tests/test_gpu_round2.py (cfg-2 at 128x128:
bf16 median 3.5e-2 per tensor); this test looks at what matters for a user: the same denoising task trained for 200 Adam steps from
the same initialisation in f32, bf16 and fp16 ends at the same loss.  The task is learnable (smooth radiance x multiplicative noise,
as in tests/test_gpu_end_to_end.py) so the loss falls by a large factor and a training run that drifted would show.
"""
import pytest
import torch

from deepdenoiser_amd import configs
from deepdenoiser_amd.naming import Naming

pytestmark = pytest.mark.gpu

STEPS, B, T = 200, 8, 64


def _task(arch, seed=0):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, T), torch.linspace(0, 1, T), indexing="ij")
    feats, labels = {}, {}
    clean = {}
    for f in arch.feature_predictions + arch.auxiliary_features:
        c = torch.rand(B, 1, 1, 3, generator=g)
        img = (c[..., 0:1] + c[..., 1:2] * yy[None, :, :, None] + c[..., 2:3] * xx[None, :, :, None]) * torch.linspace(0.5, 1.0, f.number_of_channels)
        clean[f.name] = img
        feats[Naming.source_feature_name(f.name, index=0)] = (img * (1.0 + 0.3 * torch.randn(img.shape, generator=g))).cuda()
    for f in arch.feature_predictions:
        labels[Naming.target_feature_name(f.name)] = clean[f.name].cuda()
    return feats, labels


def _train(dtype, aj=None):
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.training import Trainer
    aj, tj = aj or configs.cfg2_unet_kpcn(filters=(32, 48, 64), convs=2), configs.bench_training()
    arch = Architecture(aj, device="cuda", dtype=dtype, seed=2)
    trainer = Trainer(arch, tj, B, T, T, use_graph=True)
    feats, labels = _task(arch)
    trainer.program.set_inputs(feats, labels)
    losses = [float(trainer.step()) for _ in range(STEPS)]
    torch.cuda.synchronize()
    assert all(l == l for l in losses), "NaN loss in %s training" % dtype
    assert torch.isfinite(arch.params.values).all()
    return losses


def test_half_precision_training_ends_where_f32_training_ends():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    runs = {d: _train(d) for d in ("f32", "bf16", "f16")}
    tail = {d: sum(v[-20:]) / 20 for d, v in runs.items()}
    head = {d: v[0] for d, v in runs.items()}
    print("loss at step 1 / mean of the last 20 of %d steps: " % STEPS + ", ".join("%s %.4f / %.4f" % (d, head[d], tail[d]) for d in runs))
    assert tail["f32"] < 0.5 * head["f32"], "the task is not being learned"
    for d in ("bf16", "f16"):
        assert abs(head[d] - head["f32"]) <= 0.02 * head["f32"]
        # same destination within a few percent of the f32 run's final loss (training noise of a 200-step run included)
        assert abs(tail[d] - tail["f32"]) <= 0.08 * tail["f32"], (d, tail[d], tail["f32"])


def test_tiramisu_half_precision_training_ends_where_f32_training_ends():
    """BASELINE config 3 (Tiramisu.py:26-111 + MultiScalePrediction): the same 200-step question for the dense-block backbone -- the layer-wise
    kernel-prediction head of this config stores its 25 logits in the storage type, and its single-step half-precision gradients are gated only
    loosely (tests/test_gpu_round3.py); what a user needs is that training still goes where f32 training goes."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    aj = configs.cfg3_tiramisu(filters=(16, 24, 32), convs=4)
    runs = {d: _train(d, aj) for d in ("f32", "bf16", "f16")}
    tail = {d: sum(v[-20:]) / 20 for d, v in runs.items()}
    head = {d: v[0] for d, v in runs.items()}
    print("Tiramisu: loss at step 1 / mean of the last 20 of %d steps: " % STEPS + ", ".join("%s %.4f / %.4f" % (d, head[d], tail[d]) for d in runs))
    assert tail["f32"] < 0.5 * head["f32"], "the task is not being learned"
    for d in ("bf16", "f16"):
        assert abs(head[d] - head["f32"]) <= 0.02 * head["f32"]
        assert abs(tail[d] - tail["f32"]) <= 0.08 * tail["f32"], (d, tail[d], tail["f32"])
