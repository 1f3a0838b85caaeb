"""-m gpu: first contact with RCCL on the one-GPU box.

`torch.distributed` backend "nccl" IS RCCL on ROCm.  A communicator of ONE rank still loads librccl.so, initialises, and runs its
all-reduce kernels on the stream they are issued on -- which exercises what the 8-GPU run needs and no gloo test touches: communicator
init next to the library's own HIP context, all_reduce of arena slices on the side stream while the next hipGraph segment replays, the
`thread_local` capture mode against RCCL's watchdog thread (training.py), the mask-count all-reduce, and bench.py's barrier / max-over-ranks
path.  The multi-GPU scaling curve itself stays unmeasured until a node exists (the driver runs it)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from gpu_util import rel_l2
from test_gpu_distributed import GLOBAL_B, H, W, STEPS, _case, _global_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _train(case, force_collectives, use_graph):
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.training import Trainer
    aj, tj = _case(case)
    arch = Architecture(aj, device="cuda:0", dtype="f32", seed=2)
    trainer = Trainer(arch, tj, GLOBAL_B, H, W, world_size=1, use_graph=use_graph, n_buckets=3, force_segments=True, force_collectives=force_collectives)
    feats, labels = _global_batch(arch)
    trainer.program.set_inputs({k: v.cuda() for k, v in feats.items()}, {k: v.cuda() for k, v in labels.items()})
    losses, grads1 = [], None
    for step in range(STEPS):
        losses.append(float(trainer.step()))
        if step == 0:
            torch.cuda.synchronize()
            grads1 = arch.params.grads.cpu().clone()       # what the optimizer consumed at step 1 (all-reduced over the one rank)
    torch.cuda.synchronize()
    assert len(trainer._segments) == 3 and (trainer._graphs is not None) == use_graph
    assert trainer.reducer.active == force_collectives
    return {"losses": losses, "values": arch.params.values.cpu().clone(), "grads1": grads1, "mask_sums": trainer.program.mask_sums.cpu().clone()}


def _worker(rank, port, case, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl"
        t = torch.arange(8, dtype=torch.float32, device="cuda")
        dist.all_reduce(t)                       # sum over one rank: identity, but through RCCL's kernel
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float32))
        res = _train(case, True, True)
        dist.barrier()
        maps = open("/proc/self/maps").read()
        res["rccl_loaded"] = "librccl" in maps
        torch.save(res, out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["cfg2_small", "masked_means"])
def test_one_rank_rccl_trainer_matches_the_plain_step(case, tmp_path):
    """Trainer with the reducer forced on (3 buckets, hipGraph segments, side-stream all-reduce over a one-rank RCCL communicator) ==
    the same Trainer without collectives: losses, mask counts and the 4-step weight trajectory (up to the fp32 atomics' summation order)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = str(tmp_path / "rccl.pt")
    mp.spawn(_worker, args=(_free_port(), case, out), nprocs=1, join=True)
    r = torch.load(out)
    assert r["rccl_loaded"], "librccl.so was not mapped into the process: the nccl backend did not run on RCCL"
    plain = _train(case, False, True)
    _, tj = _case(case)
    lr = tj["learning_rate"]
    for s in range(STEPS):
        tol = 2e-6 if s == 0 else 2e-4
        assert abs(r["losses"][s] - plain["losses"][s]) <= tol * abs(plain["losses"][s]), (s, r["losses"][s], plain["losses"][s])
    assert torch.equal(r["mask_sums"], plain["mask_sums"])
    assert rel_l2(r["grads1"], plain["grads1"]) < 2e-5        # step-1 gradients: equal up to the fp32 atomics' summation order
    d = (r["values"] - plain["values"]).abs()
    assert float(d.max()) <= 2 * STEPS * lr
    assert float((d > 0.5 * lr).float().mean()) < 0.02


def test_bench_runs_its_distributed_path_over_rccl_with_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` with DD_FORCE_COLLECTIVES=1: process-group init, per-bucket
    all-reduce, barrier + max-over-ranks timing and the JSON line, argument-complete as the driver launches it for N > 1."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, DD_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "3", "--batch", "8", "--no-extras", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 1 and rec["devices"] == 1 and rec["collectives"] == "nccl"
    assert rec["steps"] == 4 and rec["value"] > 0 and rec["roofline"]["frac"] > 0
    assert rec["allreduce"]["allreduces_per_step"] >= 1 and rec["allreduce"]["allreduce_ms_per_step"] > 0 and rec["allreduce"]["exposed_ms_per_step"] >= 0


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the shape of the driver's 1-GPU command) starts two ranks itself and reports n_gpus 2.
    On the one-GPU box both ranks share device 0 over gloo (DD_FORCE_DEVICE / DD_DIST_BACKEND: test hooks); on a node the same command runs
    one rank per GPU over RCCL."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, DD_FORCE_DEVICE="0", DD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3", "--batch", "8", "--no-extras", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                      # rank 0 only
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["devices"] == 1 and rec["collectives"] == "gloo"
    assert rec["config"]["global_batch"] == 16 and rec["config"]["parallelism"] == "dp2"
    assert rec["steps"] == 3 and rec["value"] > 0
    assert rec["allreduce"] is not None and rec["allreduce"]["allreduces_per_step"] >= 2


def test_bench_refuses_more_gpus_than_visible():
    """--gpus N with fewer than N devices must fail instead of silently measuring fewer."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "DD_FORCE_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "refusing" in (p.stdout + p.stderr)
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
