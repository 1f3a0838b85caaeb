"""The kernels behind the library-internal switches.  libdd_hip.so reads its DD_* switches once per process (function-local statics), so a test
process cannot flip them: each case below runs a slice of tests/test_gpu_ops.py -- the same oracle comparisons, the same gates -- in a child
process whose environment turns a group of paths off.  What then executes is the code a default run never reaches: the plain (non wave-
specialised) conv_igemm_kernel in half precision, the register-staged wgrad_kernel, the 6 + 2 wave conv_rw_kernel forward, the plain tile walk."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # round-1 library: no wave-specialised / register-weight / LDS-DMA / GEMM-tile kernels
    "round1_kernels": ({"DD_CONV_WS": "0", "DD_CONV_RW": "0", "DD_WGRAD_DMA": "0", "DD_CONV_PW": "0"},
                       "test_conv_fwd_bwd or test_conv_non_square_batches or test_conv_grad_accumulation"),
    # the older members of the register-weight family, the plain tile walk, the mid-sized 1x1 weight gradient on the GEMM kernel
    "older_variants": ({"DD_CONV_RW8": "0", "DD_CONV_RW12": "0", "DD_CONV_XCD": "0", "DD_WGRAD_PW_MID": "1"},
                       "test_conv3x3_random_shapes_round2_kernels or test_conv_over_skip_concat or (test_conv_fwd_bwd and bf16)"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_op_parity_with_library_paths_switched_off(name):
    import torch
    if not torch.cuda.is_available():      # the child would skip every test and report none passed (plain `pytest tests` on a box without a GPU)
        pytest.skip("no GPU")
    env_extra, expr = CASES[name]
    env = dict(os.environ, **env_extra)
    env.pop("DD_PARITY_REPORT", None)
    env["DD_PARITY_FILE"] = "parity_errors_%s.txt" % name      # the child's comparisons, beside the parent's gpurun_out/parity_errors.txt
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-m", "gpu", "-x", "-q", "-k", expr,
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = "\n".join(p.stdout.strip().splitlines()[-15:])
    assert p.returncode == 0, "%s: child pytest failed\n%s\n%s" % (name, tail, p.stderr[-2000:])
    assert " passed" in tail and "no tests ran" not in tail, tail
