"""Cycle profile of the fused head backward (workgroup 0, thread 0; csrc/dd_head.hip built with -DDD_PROFILE_PHASES):
    tools/build_variant.sh hphase dd_head.hip -DDD_PROFILE_PHASES && DD_LIB=tools/exp/libdd_hphase.so python tools/head_phases.py [B]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import synthetic_inputs  # noqa: E402
from deepdenoiser_amd import _lib as L  # noqa: E402
from deepdenoiser_amd import configs  # noqa: E402
from deepdenoiser_amd.architecture import Architecture  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = L.load()
lib.dd_debug_hphases.argtypes = [C.c_void_p, C.c_int]
arch = Architecture(configs.cfg2_unet_kpcn(), device="cuda", dtype="bf16", seed=2)
prog = arch.program(B, 128, 128, training_json=configs.bench_training())
feats, labels = synthetic_inputs(arch, B, 128, 128, "cuda", 1)
prog.set_inputs(feats, labels)
for _ in range(2):
    prog.train_step()
torch.cuda.synchronize()
bwd = [op for op in prog.g.bwd_ops if getattr(op, "tag", "") == "kpcn_head"]
s = prog.g.stream_ptr()
NAMES = ["stage weights (global -> LDS) + barrier", "forward weight fragments", "gradient fragments, publish, barrier", "walker + first x request",
         "pixel loop", "barrier (slowest wave)", "zero + LDS reduction + barrier", "global atomics (waited)"]
for k, op in enumerate(bwd):
    lib.dd_debug_hphases(None, 1)
    n = 3
    for _ in range(n):
        op(s)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.dd_debug_hphases(buf, 0)
    print("head backward launch %d: cycles (workgroup 0, thread 0)" % k)
    for i, nm in enumerate(NAMES):
        print("  %-44s %9.0f" % (nm, buf[i] / n))
    print("  %-44s %9.0f" % ("total", sum(buf[:8]) / n))
