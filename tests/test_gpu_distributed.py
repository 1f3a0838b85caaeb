"""-m gpu: the REAL data-parallel Trainer in two processes on one GPU (gloo transport, both ranks on device 0): bucketed
all-reduce of the gradient arena on a side stream, hipGraph-captured backward segments, the batch-global mask count all-reduce.

Asserts: the two replicas stay bit-identical, and the sharded run equals the single-process run on the concatenated
batch -- same loss, same averaged gradients, same weight trajectory (up to the summation order of the fp32 gradient atomics).
RCCL itself needs >= 2 GPUs, which this tier's test box does not have; the code path above the transport is the same
(`torch.distributed.all_reduce`), selected by bench.py's DD_DIST_BACKEND.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from deepdenoiser_amd import configs
from deepdenoiser_amd.naming import Naming
from gpu_util import rel_l2

pytestmark = pytest.mark.gpu

STEPS = 4          # hipGraphs are captured after two eager steps: steps 3 and 4 replay them
GLOBAL_B, H, W = 4, 32, 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _case(name):
    if name == "cfg2_small":
        return configs.cfg2_unet_kpcn(filters=(32, 48, 64), convs=2), configs.bench_training()
    combined = {"Diffuse": {"Color": "Diffuse Color", "Direct": "Diffuse Direct", "Indirect": "Diffuse Indirect"},
                "Glossy": {"Color": "Glossy Color", "Direct": "Glossy Direct", "Indirect": "Glossy Indirect"}}
    tj = configs.training(image_mean=0.0, masked_mean=0.8, combined_masked_mean=1.7)
    tj["combined_image_training_settings"]["statistics"]["track_mean"] = False
    return configs.architecture(filters=(16, 16), convs=1, combined=combined), tj


def _global_batch(arch):
    """The whole mini-batch (host tensors).  Every tile gets its own black target region, so the ranks' mask counts differ."""
    g = torch.Generator().manual_seed(11)
    feats, labels = {}, {}
    for f in arch.feature_predictions + arch.auxiliary_features:
        v = torch.randn(GLOBAL_B, H, W, f.number_of_channels, generator=g)
        feats[Naming.source_feature_name(f.name, index=0)] = v if f.name == "Normal" else v.abs()
    for f in arch.feature_predictions:
        t = torch.randn(GLOBAL_B, H, W, f.number_of_channels, generator=g).abs()
        for b in range(GLOBAL_B):
            t[b, 2 * b:2 * b + 5 + 3 * b, 4:9 + 5 * b] = 0.0
        labels[Naming.target_feature_name(f.name)] = t
    return feats, labels


def _run(case, world, rank, use_graph):
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.training import Trainer
    aj, tj = _case(case)
    arch = Architecture(aj, device="cuda:0", dtype="f32", seed=2)           # identical replica on every rank
    B = GLOBAL_B // world
    trainer = Trainer(arch, tj, B, H, W, world_size=world, use_graph=use_graph, n_buckets=3, force_segments=True)
    feats, labels = _global_batch(arch)
    shard = slice(rank * B, (rank + 1) * B)
    trainer.program.set_inputs({k: v[shard].cuda() for k, v in feats.items()}, {k: v[shard].cuda() for k, v in labels.items()})
    losses, grads1 = [], None
    for step in range(STEPS):
        losses.append(float(trainer.step()))
        if step == 0:
            torch.cuda.synchronize()
            grads1 = (arch.params.grads * trainer.reducer.grad_scale).cpu().clone()        # what the optimizer consumed
    torch.cuda.synchronize()
    assert len(trainer._segments) == 3
    assert (trainer._graphs is not None) == use_graph
    return {"losses": losses, "grads1": grads1, "values": arch.params.values.cpu().clone(), "masked": trainer.program.masked,
            "mask_sums": trainer.program.mask_sums.cpu().clone()}


def _worker(rank, world, port, case, use_graph, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _run(case, world, rank, use_graph)
        torch.save(res, "%s.%d" % (out, rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case,use_graph", [("cfg2_small", True), ("masked_means", True), ("masked_means", False)])
def test_two_rank_trainer_matches_the_single_process_step(case, use_graph, tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, _free_port(), case, use_graph, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["values"], r1["values"]), "replicas diverged"
    assert torch.equal(r0["grads1"], r1["grads1"])
    single = _run(case, 1, 0, use_graph)
    _, tj = _case(case)
    lr = tj["learning_rate"]
    assert r0["masked"] == (case == "masked_means")
    if r0["masked"]:
        # both ranks hold (global count) / world, and it is the single-process count / 2; the shards' own counts differ
        assert torch.equal(r0["mask_sums"], r1["mask_sums"])
        assert torch.allclose(r0["mask_sums"] * 2, single["mask_sums"], rtol=0, atol=0)
        assert float(single["mask_sums"].abs().max()) > 0
    # the full-batch loss is the mean of the shards' losses (plain means are linear; masked means divide by the global count)
    for s in range(STEPS):
        mean_loss = 0.5 * (r0["losses"][s] + r1["losses"][s])
        tol = 2e-6 if s == 0 else 2e-4
        assert abs(mean_loss - single["losses"][s]) <= tol * abs(single["losses"][s]), (s, mean_loss, single["losses"][s])
    # step-1 gradients: all-reduced mean over ranks == gradient of the concatenated batch
    assert rel_l2(r0["grads1"], single["grads1"]) < 2e-5, rel_l2(r0["grads1"], single["grads1"])
    d = (r0["values"] - single["values"]).abs()
    assert float(d.max()) <= 2 * STEPS * lr
    assert float((d > 0.5 * lr).float().mean()) < 0.02
