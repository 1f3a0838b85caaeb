"""Data-parallel path on CPU: bucket planning (pure integers) and the sharded step with gloo, world_size 2.

The GPU Trainer runs exactly this logic (`training.plan_buckets` + `training.GradientReducer`) with RCCL and a side HIP stream;
here the same two pieces drive a toy reverse program whose gradients are known in closed form, and the result is compared with
the single-process step over the whole mini-batch (TF-form Adam from the oracle).
"""
import os
import random
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepdenoiser_amd.training import GradientReducer, plan_buckets
from oracle import tf_ops as T


# ---------------------------------------------------------------------------------------------------- plan_buckets
def _random_layout(rng):
    n_params = rng.randint(1, 12)
    params, off = [], 0
    for i in range(n_params):
        size = rng.randint(1, 40)
        params.append(("p%d" % i, off, size))
        off += size
    n_ops = rng.randint(n_params, 3 * n_params + 2)
    # reverse program: later-created parameters are (mostly) written earlier; shared layers are written several times
    writers = {}
    for i, (name, _, _) in enumerate(reversed(params)):
        base = i * n_ops // n_params
        writers[name] = sorted({min(n_ops - 1, max(0, base + rng.randint(-2, 2))) for _ in range(rng.randint(1, 3))})
    return params, writers, n_ops, off


@pytest.mark.parametrize("seed", range(40))
def test_plan_buckets_properties(seed):
    rng = random.Random(seed)
    params, writers, n_ops, total = _random_layout(rng)
    n_buckets = rng.randint(1, 6)
    plan = plan_buckets(params, {k: v[-1] for k, v in writers.items()}, n_ops, total, n_buckets)
    # segments tile the op list in order
    assert plan[0][0] == 0 and plan[-1][1] == n_ops
    for a, b in zip(plan, plan[1:]):
        assert a[1] == b[0] and a[0] <= a[1]
    # slices tile the arena, tail first
    assert plan[0][3] == total and plan[-1][2] == 0
    for a, b in zip(plan, plan[1:]):
        assert b[3] == a[2] and b[2] <= b[3]
    # safety: no op that runs AFTER a segment writes into that segment's slice
    for (b, e, lo, hi) in plan:
        for name, off, size in params:
            if off < hi and off + size > lo:
                assert all(w < e for w in writers[name]), (name, writers[name], e)


def test_plan_buckets_single_bucket_and_reference_order():
    params = [("conv2d/kernel", 0, 100), ("conv2d/bias", 100, 4), ("conv2d_1/kernel", 104, 60), ("conv2d_1/bias", 164, 4)]
    lw = {"conv2d_1/kernel": 0, "conv2d_1/bias": 0, "conv2d/kernel": 1, "conv2d/bias": 1}
    assert plan_buckets(params, lw, 3, 168, 1) == [(0, 3, 0, 168)]
    assert plan_buckets(params, lw, 3, 168, 2) == [(0, 2, 84, 168), (2, 3, 0, 84)]     # kernel 0 straddles the cut -> waits for op 1


# ---------------------------------------------------------------------------------------------------- gloo, world_size 2
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class ToyProgram:
    """y = relu(x W1 + b1) W2, loss = mean over the tile shard of a SMAPE-like term; gradients written into a flat arena by a
    3-op reverse program (W2 first, then b1, then W1), mirroring how the real reverse program fills the arena from its tail."""

    def __init__(self, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.layout = [("W1", 0, 12), ("b1", 12, 4), ("W2", 16, 8)]
        self.values = torch.randn(24, generator=g, dtype=torch.float64) * 0.5
        self.grads = torch.zeros(24, dtype=torch.float64)

    def views(self, flat):
        return flat[0:12].view(3, 4), flat[12:16], flat[16:24].view(4, 2)

    def loss_and_ops(self, x, t):
        W1, b1, W2 = self.views(self.values)
        h_pre = x @ W1 + b1
        h = h_pre.clamp(min=0)
        y = h @ W2
        den = y.abs() + t.abs() + 0.01
        loss = ((y - t).abs() / den).sum(1).mean()
        dy = (torch.sign(y - t) / den - (y - t).abs() * torch.sign(y) / den ** 2) / x.shape[0]
        gW1, gb1, gW2 = self.views(self.grads)
        state = {}

        def op_w2():
            gW2.copy_(h.t() @ dy)
            state["dh"] = (dy @ W2.t()) * (h_pre > 0)

        def op_b1():
            gb1.copy_(state["dh"].sum(0))

        def op_w1():
            gW1.copy_(x.t() @ state["dh"])

        return loss, [op_w2, op_b1, op_w1], {"W2": 0, "b1": 1, "W1": 2}


def _data(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, generator=g, dtype=torch.float64), torch.randn(n, 2, generator=g, dtype=torch.float64).abs()


def _dp_worker(rank, world, port, n_buckets, steps, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prog = ToyProgram(seed=0)                            # identical replica on every rank
        x, t = _data(16, seed=5)
        shard = slice(rank * 16 // world, (rank + 1) * 16 // world)     # per-rank tile shard
        m, v = torch.zeros_like(prog.values), torch.zeros_like(prog.values)
        reducer = GradientReducer(prog.grads, world)
        for step in range(1, steps + 1):
            prog.grads.zero_()
            _, ops, lw = prog.loss_and_ops(x[shard], t[shard])
            for (b, e, lo, hi) in plan_buckets(prog.layout, lw, len(ops), 24, n_buckets):
                for op in ops[b:e]:
                    op()
                reducer.launch(lo, hi)
            reducer.wait()
            T.adam_step([prog.values], [prog.grads * reducer.grad_scale], [m], [v], step, 1e-2)
        gathered = [torch.zeros_like(prog.values) for _ in range(world)]
        dist.all_gather(gathered, prog.values)
        if rank == 0:
            torch.save(gathered, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_buckets", [1, 2, 3])
def test_sharded_step_matches_full_batch_gloo(n_buckets, tmp_path):
    steps = 3
    out = str(tmp_path / "replicas.pt")
    mp.spawn(_dp_worker, args=(2, _free_port(), n_buckets, steps, out), nprocs=2, join=True)
    replicas = torch.load(out)
    assert torch.equal(replicas[0], replicas[1])            # ranks stay bit-identical replicas
    # single process over the whole mini-batch
    prog = ToyProgram(seed=0)
    x, t = _data(16, seed=5)
    m, v = torch.zeros_like(prog.values), torch.zeros_like(prog.values)
    for step in range(1, steps + 1):
        prog.grads.zero_()
        _, ops, _ = prog.loss_and_ops(x, t)
        for op in ops:
            op()
        T.adam_step([prog.values], [prog.grads.clone()], [m], [v], step, 1e-2)
    assert torch.allclose(replicas[0], prog.values, rtol=0, atol=1e-12)


def test_toy_gradients_match_autograd():
    prog = ToyProgram(seed=0)
    x, t = _data(16, seed=5)
    vals = prog.values.clone().requires_grad_(True)
    W1, b1, W2 = vals[0:12].view(3, 4), vals[12:16], vals[16:24].view(4, 2)
    y = (x @ W1 + b1).clamp(min=0) @ W2
    loss = ((y - t).abs() / (y.abs() + t.abs() + 0.01)).sum(1).mean()
    loss.backward()
    l2, ops, _ = prog.loss_and_ops(x, t)
    for op in ops:
        op()
    assert abs(float(l2) - float(loss)) < 1e-12
    assert torch.allclose(prog.grads, vals.grad, atol=1e-12)
