"""Builds libdd_hip.so (the C-ABI library of hand-written gfx950 kernels) in-tree with hipcc.

    python -m deepdenoiser_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["dd_conv_igemm.hip", "dd_conv_wgrad.hip", "dd_pointwise.hip", "dd_compose.hip", "dd_head.hip", "dd_conv_bwd.hip", "dd_convt.hip", "dd_conv_rw.hip", "dd_conv_wgrad96.hip"]
HEADERS = ["dd_common.h", os.path.join("..", "..", "include", "dd_hip.h")]
LIB = os.path.join(HERE, "libdd_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Kernels whose long-lived accumulators are updated by in-place inline-asm MFMAs (no hazard recogniser sees inside the asm): they are only
# correct while hipcc neither spills nor otherwise touches those registers, so a build in which one of them uses scratch memory is refused.
NO_SCRATCH = {"dd_conv_bwd.hip": ("conv_bwd_kernel",), "dd_convt.hip": ("convt_bwd_kernel",), "dd_conv_wgrad96.hip": ("wgrad96_kernel",)}


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _check_no_scratch(src, names, remarks):
    """Parse hipcc's -Rpass-analysis=kernel-resource-usage remarks: every kernel whose name contains one of `names` must use no scratch."""
    import re
    seen = 0
    for block in remarks.split("Function Name: ")[1:]:
        fn = block.split()[0]
        if not any(n in fn for n in names):
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", block)
        if m is None:
            continue
        seen += 1
        if int(m.group(1)) != 0:
            raise RuntimeError("%s: kernel %s uses %s bytes of scratch per lane; its in-place asm MFMA accumulators require a spill-free build" % (src, fn, m.group(1)))
    if seen == 0:
        raise RuntimeError("%s: no resource-usage remark found for %s" % (src, ", ".join(names)))


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            guarded = src in NO_SCRATCH
            cmd = [hipcc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if guarded else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, src, o, subprocess.Popen(cmd, stderr=subprocess.PIPE if guarded else None, text=True if guarded else None)))
    for cmd, src, o, p in procs:
        err = p.communicate()[1] if src in NO_SCRATCH else None
        if p.wait() != 0:
            if err:
                sys.stderr.write(err)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if src in NO_SCRATCH:
            try:
                _check_no_scratch(src, NO_SCRATCH[src], err)
            except RuntimeError:
                os.remove(o)
                raise
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
