"""The kernels behind the library-internal switches.  libdd_hip.so reads its DD_* switches once per process (function-local statics), so a test
process cannot flip them: each case below runs a slice of tests/test_gpu_ops.py -- the same oracle comparisons, the same gates -- in a child
process whose environment turns a group of paths off.  What then executes is the code a default run never reaches: the plain (non wave-
specialised) conv_igemm_kernel in half precision, the register-staged wgrad_kernel, the 6 + 2 wave conv_rw_kernel forward, the plain tile walk."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {   # name: (environment of the child, -k expression[, test file (default tests/test_gpu_ops.py)])
    # round-1 library: no wave-specialised / register-weight / LDS-DMA / GEMM-tile kernels
    "round1_kernels": ({"DD_CONV_WS": "0", "DD_CONV_RW": "0", "DD_WGRAD_DMA": "0", "DD_CONV_PW": "0"},
                       "test_conv_fwd_bwd or test_conv_non_square_batches or test_conv_grad_accumulation"),
    # the older members of the register-weight family, the plain tile walk, the mid-sized 1x1 weight gradient on the GEMM kernel
    "older_variants": ({"DD_CONV_RW8": "0", "DD_CONV_RW12": "0", "DD_CONV_XCD": "0", "DD_WGRAD_PW_MID": "1"},
                       "test_conv3x3_random_shapes_round2_kernels or test_conv_over_skip_concat or (test_conv_fwd_bwd and bf16)"),
    # round 5 (dead-path coverage): the kernels the round-4 defaults superseded stay reachable through switches and odd shapes, so they
    # stay gated -- the 6 + 2 wave conv_rw_kernel with its own mask / residual loads (the all-wave AUX forms off), the generic max-pool backward,
    "masked_6plus2_and_generic_unpool": ({"DD_CONV_RW12_MASK": "0", "DD_CONV_RW8_MASK": "0", "DD_MAXPOOL_GENERIC": "1"},
                                         "test_conv_fwd_bwd or test_conv_over_skip_concat or test_conv_grad_accumulation or test_maxpool"),
    # ... and the 16 x 16-tile compose kernels of round 2 / 3 (csrc/dd_compose.hip) in place of the row-streaming ones, forward and backward
    "tile_compose_kernels": ({"DD_COMPOSE_STREAM": "0", "DD_COMPOSE_STREAM_BWD": "0"},
                             "test_fused_compose_net_matches_the_layerwise_path", "test_gpu_round2.py"),
    # round 5: the loss launch before its per-pixel kernels (one thread re-reading features per term, global read-modify-write gradients) and with
    # the inverse standardization as launches of its own -- the path that still carries the variation terms
    "older_loss_kernel_unfused": ({"DD_LOSS_SIMPLE": "0", "DD_LOSS_GENERAL": "0", "DD_FUSE_LOSS_INVERT": "0"},
                                  "(test_training_step_parity_f32 and (example_json or cfg2)) or test_masked_mean", "test_gpu_model.py"),
    # ... and the launches round 5 merged, apart again: one weight-gradient launch per conv of a dense block and per 128-channel layer, one head
    # backward per scale, the per-channel-lane column sums, thin layers K-streamed
    "round5_merges_off": ({"DD_WGRAD_STACK": "0", "DD_WGRAD_MULTI": "0", "DD_HEAD_BWD_MULTI": "0", "DD_COLSUM_VEC": "0", "DD_CONV_KS_THIN": "1"},
                          "test_small_networks_half_precision or test_backward_of_the_fused or test_cfg2_full_size_half", "test_gpu_round3.py"),
    # round 6: the fused backward of the 65 - 96-channel level (csrc/dd_conv_bwd96.hip) off -> the two launches it replaced: the 12-wave masked
    # register-weight data gradient + the weight-gradient role per 64 x 64 block pair; and the engine-level paths of rounds 1 / 2 that no
    # default configuration reaches any more (unfused input assembly, layer-wise transposed conv, one-launch conv over the skip concat, the
    # two-launch backward of every 3 x 3 layer), on the half-precision networks that would otherwise never run them
    "two_launch_backward_of_the_96_channel_level": ({"DD_CONV_BWD96": "0"},
                                                    "test_conv3x3_random_shapes_round2_kernels or test_conv_grad_accumulation or test_conv_over_skip_concat"),
    "engine_paths_of_rounds_1_and_2": ({"DD_FUSE_INPUT": "0", "DD_CONVT_STREAM": "0", "DD_CONV_SPLIT_CONCAT": "0", "DD_FUSE_CONV_BWD": "0"},
                                       "test_small_networks_half_precision and (cfg2 or example_json or ragged)", "test_gpu_round3.py"),
    "tile_compose_kernels_bit_faithful": ({"DD_COMPOSE_STREAM": "0", "DD_COMPOSE_STREAM_BWD": "0"},
                                          "test_backward_of_the_fused_head_and_compose_kernels_is_bit_faithful", "test_gpu_round3.py"),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_op_parity_with_library_paths_switched_off(name):
    import torch
    if not torch.cuda.is_available():      # the child would skip every test and report none passed (plain `pytest tests` on a box without a GPU)
        pytest.skip("no GPU")
    env_extra, expr = CASES[name][:2]
    test_file = CASES[name][2] if len(CASES[name]) > 2 else "test_gpu_ops.py"
    env = dict(os.environ, **env_extra)
    env.pop("DD_PARITY_REPORT", None)
    env["DD_PARITY_FILE"] = "parity_errors_%s.txt" % name      # the child's comparisons, beside the parent's gpurun_out/parity_errors.txt
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", test_file), "-m", "gpu", "-x", "-q", "-k", expr,
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = "\n".join(p.stdout.strip().splitlines()[-15:])
    assert p.returncode == 0, "%s: child pytest failed\n%s\n%s" % (name, tail, p.stderr[-2000:])
    assert " passed" in tail and "no tests ran" not in tail, tail
