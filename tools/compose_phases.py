"""Cycle profile of the fused compose-net kernels (workgroup 0; csrc/dd_compose.hip built with -DDD_PROFILE_PHASES by tools/build_variant.sh):
    tools/build_variant.sh cphase dd_compose.hip -DDD_PROFILE_PHASES && DD_LIB=tools/exp/libdd_cphase.so python tools/compose_phases.py [B]"""
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
lib.dd_debug_cphases.argtypes = [C.c_void_p, C.c_int]
arch = Architecture(configs.cfg2_unet_kpcn(), device="cuda", dtype="bf16", seed=2)
prog = arch.program(B, 128, 128, training_json=configs.bench_training())
feats, labels = synthetic_inputs(arch, B, 128, 128, "cuda", 1)
prog.set_inputs(feats, labels)
for _ in range(2):
    prog.train_step()
torch.cuda.synchronize()
fwd = [op for op in prog.g.fwd_ops if getattr(op, "tag", "") == "compose_net"]
bwd = [op for op in prog.g.bwd_ops if getattr(op, "tag", "") == "compose_net"]
s = prog.g.stream_ptr()
FWD = ["tile start", "L0 (input -> a1)", "barrier", "issue next x0 + save a1", "L1 conv", "barrier", "save r1", "L2 conv", "barrier", "save a2", "L3 conv",
       "barrier", "save r3", "L4 conv + 1x1 + blend", "barrier"]
BWD = ["tile start (+ issue r3 load)", "S0 (blend / 1x1 backward)", "store r3 frame", "barrier", "S1 compute (+ issue a2 load)", "barrier", "store a2 frame", "barrier",
       "S2 compute (+ issue r1 load)", "barrier", "store r1 frame", "barrier", "S3 compute (+ issue a1 load)", "barrier", "store a1 frame", "barrier",
       "S4 compute (+ issue a3 load)", "barrier", "store a3 frame", "S5 pixel part + barrier", "S5 1x1 weight gradients + barrier"]
for name, ops in (("forward 128^2", fwd[-1:]), ("backward 128^2", bwd[:1])):
    lib.dd_debug_cphases(None, 1)
    n = 3
    for _ in range(n):
        for op in ops:
            op(s)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 64)()
    lib.dd_debug_cphases(buf, 0)
    if name.startswith("forward"):
        tiles = buf[15] or 1
        print("%s: %d tiles per launch in workgroup 0; cycles per tile (thread 0):" % (name, tiles // n))
        tot = 0
        for i, nm in enumerate(FWD):
            print("  %-36s %8.0f" % (nm, buf[i] / tiles)); tot += buf[i] / tiles
        print("  %-36s %8.0f" % ("total", tot))
    else:
        tiles = (buf[15] // 1) or 1
        for role, base in (("data-gradient role (thread 0)", 16), ("weight-gradient role (thread 256)", 40)):
            print("%s, %s: cycles per tile" % (name, role))
            tot = 0
            for i, nm in enumerate(BWD):
                v = buf[base + i] / (32 * n)
                print("  %-36s %8.0f" % (nm, v)); tot += v
            print("  %-36s %8.0f" % ("total", tot))
