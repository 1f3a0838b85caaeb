"""What the layer-wise kernel-prediction head costs where the fused one does not apply (COMBINED tuples: 3 members per tuple, 75 logits per scale):
the literal example architecture with SINGLE (17 tuple passes, fused head), SINGLE with the fused head off, and COMBINED (8 tuple passes, layer-wise head).
    python tools/combined_cost.py [B]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(tuple_type, B):
    import torch
    from bench import synthetic_inputs
    from deepdenoiser_amd import configs
    from deepdenoiser_amd.architecture import Architecture
    from deepdenoiser_amd.training import Trainer
    arch = Architecture(configs.architecture(tuple_type=tuple_type), device="cuda", dtype="bf16", seed=2)
    tr = Trainer(arch, configs.training(batch_size=B), B, 128, 128)
    f, l = synthetic_inputs(arch, B, 128, 128, "cuda", 5)
    tr.program.set_inputs(f, l)
    for _ in range(3):
        tr.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        tr.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    fam, _ = tr.program.profile_ops(repeats=2, detail=True)
    heads = {k: round(v[1], 3) for k, v in fam.items() if k in ("kpcn_head", "kpcn_apply", "conv_igemm", "conv_wgrad", "pointwise", "loss_head")}
    print("%-8s fused_head=%-5s tuples=%2d  %.2f ms/step  %.1f tiles/s  %.0f tuple passes/s   launch ms: %s"
          % (tuple_type, bool(tr.program.fused_head), tr.program.T, dt * 1e3, B / dt, B * tr.program.T / dt, heads), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        one(sys.argv[2], int(sys.argv[1]))
    else:
        B = sys.argv[1] if len(sys.argv) > 1 else "8"
        for tt, env in (("SINGLE", {}), ("SINGLE", {"DD_FUSE_HEAD": "0"}), ("COMBINED", {})):
            subprocess.run([sys.executable, os.path.abspath(__file__), B, tt], env=dict(os.environ, **env), check=False)
