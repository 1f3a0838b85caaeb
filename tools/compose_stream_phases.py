"""Cycle stamps of the row-streaming compose kernel (build with -DCS_PROFILE: tools/build_variant.sh cs_prof dd_compose_stream.hip -DCS_PROFILE).
    DD_LIB=tools/exp/libdd_cs_prof.so python tools/compose_stream_phases.py
Per layer role (task slot 0 of workgroup 0), cycles per step: 0 bookkeeping -> barrier, 1 barrier wait, 2 stage 0, 3 LDS reads issued (+ waits the
compiler put there), 4 MFMA chain, 5 epilogue."""
import ctypes as C
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["CS_SHAPES"] = os.environ.get("CS_SHAPES", "128,128,128")
from deepdenoiser_amd import _lib as L  # noqa: E402

lib = L.load()
lib.dd_debug_cs_phases.argtypes = [C.c_void_p, C.c_int]
import torch  # noqa: E402
sys.argv = [sys.argv[0]]
buf = (C.c_ulonglong * 32)()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "compose_stream_bench.py")).read()
src = src.replace("for save in (False, True):", "for save in (%s,):" % os.environ.get("CS_SAVE", "False"))
torch.cuda.init()
lib.dd_debug_cs_phases(None, 1)
exec(compile(src.replace("def run(N, H, W, save, reps=20)", "def run(N, H, W, save, reps=10)"), "bench", "exec"))
lib.dd_debug_cs_phases(buf, 0)
N, H, W = [int(v) for v in os.environ["CS_SHAPES"].split(";")[0].split(",")]
o = (C.c_int * 8)()
lib.dd_compose_stream_plan(N, H, W, 256, o)
units = N * o[3] * o[6]
per = -(-units // min(units, 256))
steps = (per * o[7] + 4 * (o[1] + 1) + o[1] - 1) // o[1]
launches = 13
print("plan", list(o), "steps per launch (workgroup 0)", steps)
names = ["->barrier", "barrier", "stage0", "reads", "mfma", "epilogue", "", ""]
for l in range(4):
    vals = [buf[l * 8 + i] / (launches * steps) for i in range(6)]
    print("layer %d: " % l + "  ".join("%s %.0f" % (names[i], vals[i]) for i in range(6)) + "   total %.0f cycles/step" % sum(vals))
