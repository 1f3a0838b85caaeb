"""Helpers shared by the -m gpu tests (HIP path vs the CPU oracle)."""
import torch

TOL = {"f32": 3e-5, "bf16": 2.5e-2, "f16": 3.5e-3}


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check(name, got, want, tol):
    assert tuple(got.shape) == tuple(want.shape), (name, tuple(got.shape), tuple(want.shape))
    assert torch.isfinite(got.double()).all(), name + ": non-finite values"
    e = rel_l2(got, want)
    assert e <= tol, "%s: rel-L2 %.3e > %.1e (max abs diff %.3e, |want| max %.3e)" % (
        name, e, tol, float((got.double().cpu() - want.double().cpu()).abs().max()), float(want.double().abs().max()))
    return e


def representable(t, dtype):
    """Round values so that storing them in the graph dtype is lossless."""
    if dtype == "bf16":
        return t.to(torch.bfloat16).to(torch.float64)
    if dtype == "f16":
        return t.to(torch.float16).to(torch.float64)
    return t.to(torch.float32).to(torch.float64)


def fill(dt, values):
    """values: [B,H,W,C] float tensor -> device tensor channels (pad channels stay zero)."""
    dt.buf[..., dt.ch0:dt.ch0 + dt.C] = values.to(dt.buf.dtype).to(dt.buf.device)


def read(dt):
    return dt.buf[..., dt.ch0:dt.ch0 + dt.C].double().cpu()


def set_param(ps, p, value):
    ps.value(p).copy_(value.to(torch.float32))
