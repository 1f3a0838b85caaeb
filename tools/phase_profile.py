"""Builds a DD_PROFILE_PHASES variant of the library and prints per-unit cycle counts of the conv kernel phases.
    python tools/phase_profile.py k cin cout H W B [dgrad]"""
import ctypes as C, os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "deepdenoiser_amd", "csrc")
so = os.environ.get("DD_PROF_SO", os.path.join(ROOT, "build", "libdd_prof.so"))
os.makedirs(os.path.dirname(so), exist_ok=True)
extra = [a for a in sys.argv if a.startswith("-D")]
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
prebuilt = "--prebuilt" in sys.argv
if "--prebuilt" not in sys.argv:
  subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDD_PROFILE_PHASES"] + extra + [
                       os.path.join(csrc, "dd_conv_igemm.hip"), os.path.join(csrc, "dd_conv_wgrad.hip"), os.path.join(csrc, "dd_pointwise.hip"), "-o", so])
from deepdenoiser_amd import _lib as L
L.LIB_PATH = so
lib = L.load()
lib.dd_debug_phases.argtypes = [C.c_void_p, C.c_int]
from deepdenoiser_amd.engine import Graph
args = [a for a in sys.argv[1:] if a != "--prebuilt"]
k, cin, cout, H, W, B = [int(v) for v in args[:6]]
dgrad = len(args) > 6
g = Graph("cuda", "bf16")
x = g.tensor(B, H, W, cin, relu=True, requires_grad=True); x.buf.normal_()
lay = g.layer("b/conv2d", k, cin, cout)
y = g.conv(x, lay, relu=True); y.mark_grad_written(); g.build_backward(); g.finalize(); y.grad().buf.normal_()
s = g.stream_ptr(); g.run(g.pack_ops)
wgrad = len(args) > 6 and args[6] == "wgrad"
if wgrad:
    lib.dd_debug_wphases.argtypes = [C.c_void_p, C.c_int]
    op = [o for o in g.bwd_ops if getattr(o, "tag", "") == "conv_wgrad"][0]
    op(s); torch.cuda.synchronize()
    lib.dd_debug_wphases(None, 1)
    n = 5
    for _ in range(n):
        op(s)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    lib.dd_debug_wphases(buf, 0)
    units = buf[5] or 1
    print("wgrad tiles per launch (block 0):", units / n)
    print("block 0, whole kernel: %.0f ticks, %.2f us by the 100 MHz wall clock" % (buf[14] / n, buf[15] / n / 100.0))
    for i, nm in [(0, "sync (+ tile -> LDS)"), (1, "issue next tile loads"), (2, "MFMA phase (+ DMA issue)")]:
        print("%-30s %9.0f cycles/tile" % (nm, buf[i] / units))
    print("%-30s %9.0f cycles/launch" % ("epilogue atomics", buf[3] / n))
    sys.exit(0)
op = [o for o in (g.bwd_ops if dgrad else g.fwd_ops) if getattr(o, "tag", "") == "conv_igemm"][0]
op(s); torch.cuda.synchronize()
lib.dd_debug_phases(None, 1)
n = 5
for _ in range(n):
    op(s)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
lib.dd_debug_phases(buf, 0)
units = buf[5] or 1
print("units per launch (block 0):", units / n)
if buf[15]:
    print("block 0, whole kernel: %.0f ticks, %.2f us by the 100 MHz wall clock => counter runs at %.0f MHz" % (buf[14] / n, buf[15] / n / 100.0, buf[14] / (buf[15] / 100.0)))
if buf[8] or buf[9]:
    for i, nm in [(0, "MFMA role: mma phase"), (1, "MFMA role: wait bar1"), (2, "MFMA role: write stage"), (3, "MFMA role: wait bar2"),
                  (8, "I/O role: issue patch loads"), (9, "I/O role: drain stage"), (10, "I/O role: wait bar1"), (11, "I/O role: patch -> LDS"), (12, "I/O role: wait bar2")]:
        print("%-30s %9.0f cycles/unit" % (nm, buf[i] / units))
else:
    tot = 0
    for i, nm in enumerate(["patch_load issue", "MFMA loop", "barrier+epilogue", "patch_store", "end barrier"]):
        print("%-18s %9.0f cycles/unit" % (nm, buf[i] / units)); tot += buf[i] / units
    print("%-18s %9.0f cycles/unit" % ("total", tot))
