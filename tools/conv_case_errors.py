"""Error of one conv layer's forward / dx / dW / db against the f64 oracle: python tools/conv_case_errors.py k cin cout H W dtype [B]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_ops as TG  # noqa: E402
from gpu_util import rel_l2  # noqa: E402
from deepdenoiser_amd import engine  # noqa: E402


def show(name, got, want, tol):
    print("  %-4s rel-L2 %.3e (gate %.1e)" % (name, rel_l2(got, want), tol))
    return 0.0


TG.check = show
k, cin, cout, H, W = [int(v) for v in sys.argv[1:6]]
dtype = sys.argv[6]
B = int(sys.argv[7]) if len(sys.argv) > 7 else 1
print("conv %dx%d %d->%d %dx%d B=%d %s  DD_CONV_RW=%s DD_FUSE_CONV_BWD=%s" % (k, k, cin, cout, H, W, B, dtype, os.environ.get("DD_CONV_RW", "1"),
      os.environ.get("DD_FUSE_CONV_BWD", "1")))
TG._conv_case(engine, dtype, k, cin, cout, H, W, True, False, False, True, B=B)
