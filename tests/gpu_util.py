"""Helpers shared by the -m gpu tests (HIP path vs the CPU oracle)."""
import torch

import os

# Gates by what an output IS, not by which path produced it:
#   TOL    a whole half-precision network against the PLAIN f64 oracle: the storage type's own accumulated error;
#   ROUND  a storage-type output (y, dx) computed from inputs that are representable in the storage type, or any tensor compared with the
#          storage-emulating oracle (oracle.model, storage=...): one rounding is rel-L2 ~ 2^-9/sqrt(3) = 1.1e-3 (bf16), 2^-12/sqrt(3) =
#          1.4e-4 (fp16) -- gated at <= 2 roundings' worth so that a 1-ulp flip on a rounding boundary passes and a dropped channel, tap or
#          halo column (>= 1/192 of the terms = 5e-3 relative at the very least) does not;
#   ACC32  an fp32 output (dW, db, kernel-prediction output) computed from representable inputs: bf16 x bf16 products are exact in fp32, so
#          only the fp32 summation order differs from the f64 oracle: 5e-6 (a dropped term out of ~10^4..10^6 would still show at ~1e-3).
TOL = {"f32": 3e-5, "bf16": 2.5e-2, "f16": 3.5e-3}
# measured over the whole op suite (profiles/r03_parity_errors.txt): one rounding = 1.67e-3 (bf16) / 2.08e-4 (fp16), worst storage-type
# output 2.5e-3 / 3.1e-4 (two roundings: a conv over a split skip concat); fp32 outputs <= 3e-7 in every storage type; f32 path <= 7e-7
ROUND = {"f32": 5e-6, "bf16": 4e-3, "f16": 5e-4}
ACC32 = {"f32": 5e-6, "bf16": 5e-6, "f16": 5e-6}

# every comparison of the session: (test id, name, measured rel-L2, gate); tests/conftest.py writes them to gpurun_out/parity_errors.txt
RECORDS = []
REPORT_ONLY = os.environ.get("DD_PARITY_REPORT", "0") != "0"      # measure everything, fail nothing (how the gates were set)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def check(name, got, want, tol):
    assert tuple(got.shape) == tuple(want.shape), (name, tuple(got.shape), tuple(want.shape))
    assert torch.isfinite(got.double()).all(), name + ": non-finite values"
    e = rel_l2(got, want)
    RECORDS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], name, e, tol))
    if REPORT_ONLY:
        return e
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


def gate(name, value, bound):
    """A scalar bound (loss error, median / max over tensors) recorded like check()."""
    RECORDS.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], name, float(value), float(bound)))
    if not REPORT_ONLY:
        assert value <= bound, "%s: %.3e > %.1e" % (name, value, bound)
    return value
