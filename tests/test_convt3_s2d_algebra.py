"""The index algebra behind the transposed conv's backward on the space-to-depth output gradient (engine.convt3_s2d_blocks, dd_space_to_depth2,
dd_conv3x3_ks mode 6, dd_convt3_wgrad), checked on the CPU against autograd of the oracle's tf.layers.conv2d_transpose(3x3, strides 2, 'same')
(oracle/tf_ops.py, Tiramisu.py:60-65).  No device code runs here: the GPU tests check the kernels, this pins what they are asked to compute."""
import pytest
import torch

from oracle import tf_ops as T


def _blocks():
    import os
    import re
    # engine.py loads the HIP library at Graph construction only, but importing it pulls torch + ctypes bindings in: read the helper's source instead
    src = open(os.path.join(os.path.dirname(__file__), "..", "deepdenoiser_amd", "engine.py")).read()
    m = re.search(r"def convt3_s2d_blocks\(\):.*?\n    return out\n", src, re.S)
    ns = {}
    exec(m.group(0), ns)
    return ns["convt3_s2d_blocks"]()


def _space_to_depth(dy, cp):
    B, H2, W2, C = dy.shape
    H, W = H2 // 2, W2 // 2
    s = torch.zeros(B, H, W, 4 * cp, dtype=dy.dtype)
    for py in range(2):
        for px in range(2):
            s[..., (py * 2 + px) * cp:(py * 2 + px) * cp + C] = dy[:, py::2, px::2, :]
    return s


@pytest.mark.parametrize("B,H,W,cin,cout", [(1, 3, 4, 5, 2), (2, 5, 3, 4, 7), (1, 1, 1, 3, 3), (1, 6, 6, 8, 16)])
def test_data_and_filter_gradient_from_the_space_to_depth_gradient(B, H, W, cin, cout):
    gen = torch.Generator().manual_seed(B * 1000 + H * 100 + W * 10 + cin)
    x = torch.randn(B, H, W, cin, generator=gen, dtype=torch.float64, requires_grad=True)
    K = torch.randn(3, 3, cout, cin, generator=gen, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(B, 2 * H, 2 * W, cout, generator=gen, dtype=torch.float64)
    y = T.conv2d_transpose_s2(x, K)
    dx_ref, dk_ref = torch.autograd.grad((y * dy).sum(), [x, K])

    cp = (cout + 15) // 16 * 16
    s = _space_to_depth(dy, cp)
    sp = torch.zeros(B, H + 1, W + 1, 4 * cp, dtype=torch.float64)      # zero beyond the grid: what the kernels' bounds checks supply
    sp[:, :H, :W] = s
    dx = torch.zeros_like(dx_ref)
    dk = torch.zeros_like(dk_ref)
    seen_image_taps = set()
    for a, b, di, dj, plane, image_tap in _blocks():
        win = sp[:, di:di + H, dj:dj + W, plane * cp:plane * cp + cout]            # [B,H,W,cout]
        dx += torch.einsum("bhwo,oi->bhwi", win, K[a, b].detach())
        dk[a, b] = torch.einsum("bhwo,bhwi->oi", win, x.detach())
        assert image_tap == (1 + di) * 3 + (1 + dj) and image_tap in (4, 5, 7, 8)
        seen_image_taps.add((image_tap, plane))
    assert len(seen_image_taps) == 9                                                  # nine distinct (image tap, column block) cells
    assert torch.allclose(dx, dx_ref, rtol=1e-12, atol=1e-12)
    assert torch.allclose(dk, dk_ref, rtol=1e-12, atol=1e-12)
