"""Builds libdd_hip.so (the C-ABI library of hand-written gfx950 kernels) in-tree with hipcc.

    python -m deepdenoiser_amd.build [--force] [--resources]

hipcc cross-compiles for gfx950 without a GPU; the built .so travels to the GPU box with the repo snapshot.  The library carries the hash
of the sources it was built from (dd_version(), csrc/dd_version.hip); _lib.load() compares it with the sources next to it.
"""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["dd_conv_igemm.hip", "dd_conv_wgrad.hip", "dd_pointwise.hip", "dd_compose.hip", "dd_head.hip", "dd_conv_bwd.hip", "dd_convt.hip",
           "dd_conv_rw.hip", "dd_conv_ks.hip", "dd_conv_pw.hip", "dd_conv_bwd96.hip", "dd_compose_stream.hip", "dd_compose_stream_bwd.hip"]
VERSION_SRC = "dd_version.hip"
HEADERS = ["dd_common.h", "dd_compose_stream.h", os.path.join("..", "..", "include", "dd_hip.h")]
LIB = os.path.join(HERE, "libdd_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# Kernels that must not use scratch memory.  (1) Kernels whose long-lived accumulators are updated by in-place inline-asm MFMAs (no hazard
# recogniser sees inside the asm) are only CORRECT while hipcc neither spills nor otherwise touches those registers.  (2) The fused head /
# compose / register-weight / transposed-conv kernels are HBM- or latency-bound: a spill there is silent HBM traffic (round 2 shipped a
# compose backward that wrote 15x its algorithmic bytes to scratch).  A build in which one of them reports a non-zero ScratchSize is refused.
NO_SCRATCH = {
    "dd_conv_bwd.hip": ("conv_bwd_kernel", "conv_bwd_multi_kernel"),
    "dd_conv_bwd96.hip": ("conv_bwd96_kernel",),
    "dd_pointwise.hip": ("kpcn_fwd_kernel", "kpcn_bwd_kernel", "assemble_input_kernel"),
    "dd_convt.hip": ("convt_bwd_kernel", "convt_fwd_kernel"),
    "dd_conv_rw.hip": ("conv_rw_kernel", "conv_rw8_kernel"),
    "dd_compose.hip": ("compose_fwd_kernel", "compose_bwd_kernel"),
    "dd_head.hip": ("head_fwd_kernel", "head_bwd_kernel", "head_bwd_multi_kernel"),
    "dd_conv_ks.hip": ("conv_ks_kernel",),
    "dd_conv_pw.hip": ("conv_pw_kernel", "wgrad_pw_kernel"),
    "dd_compose_stream.hip": ("compose_stream_fwd_kernel",),
    "dd_compose_stream_bwd.hip": ("compose_stream_wgrad_kernel",),
}


# Sources with "request now, wait later" inline-asm loads: (request marker, wait marker).  Between the two hipcc believes the destination
# registers defined although the load may not have landed; the build compiles the file to ISA and refuses it if any instruction in between
# names one of them (correctness must not depend on the current register allocation).
ASYNC_LOADS = {"dd_convt.hip": ("dd_accum_request", "dd_accum_wait")}
_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    out = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_async_loads(asm, request, wait):
    """asm: device ISA text (hipcc -S).  Returns the number of request groups checked; raises if an instruction between a request and the
    following wait touches a register a pending request writes."""
    pending, groups, fn = {}, 0, "?"
    for ln, line in enumerate(asm.splitlines(), 1):
        code = line.split(";", 1)[0].strip()
        if line and not line[0].isspace() and line.rstrip().endswith(":") and not line.startswith(".") and not line.startswith(";"):
            fn = line.rstrip()[:-1]
            if pending:
                raise RuntimeError("%s: request at line %d never reached its wait" % (fn, min(pending.values())))
        if request in line:
            dst = code.split(None, 1)[1].split(",", 1)[0]
            if not pending:
                groups += 1
            # its address operand must not be the destination of an EARLIER request (its own destination may overlap it: the address is read
            # when the instruction issues)
            hit = _vregs(code.split(",", 1)[1]) & set(pending)
            for r in _vregs(dst):
                pending[r] = ln
        elif wait in line:
            pending = {}
            continue
        else:
            hit = _vregs(code) & set(pending) if (pending and code and not code.startswith(".")) else set()
        if hit:
            raise RuntimeError("%s: line %d `%s` touches v%s while its load (line %d) is still in flight"
                               % (fn, ln, code, sorted(hit), pending[sorted(hit)[0]]))
    if pending:
        raise RuntimeError("%s: request never reached its wait" % fn)
    return groups


def source_hash():
    """First 16 hex digits of sha256 over every kernel source and header (name + content, sorted by name)."""
    h = hashlib.sha256()
    for name in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(os.path.basename(name).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def parse_resources(remarks):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {kernel name: {"vgprs", "agprs", "spill", "scratch", "occupancy", "lds"}}."""
    out = {}
    for block in remarks.split("Function Name: ")[1:]:
        fn = block.split()[0]

        def num(pat):
            m = re.search(pat + r": (\d+)", block)
            return int(m.group(1)) if m else None
        out[fn] = {"vgprs": num(r"\bVGPRs"), "agprs": num(r"AGPRs"), "spill": num(r"VGPRs Spill"), "scratch": num(r"ScratchSize \[bytes/lane\]"),
                   "occupancy": num(r"Occupancy \[waves/SIMD\]"), "lds": num(r"LDS Size \[bytes/block\]"), "sgprs": num(r"\bSGPRs")}
    return out


def _check_no_scratch(src, names, remarks):
    """Every kernel whose name contains one of `names` must use no scratch."""
    res = parse_resources(remarks)
    seen = 0
    for fn, r in res.items():
        if not any(n in fn for n in names) or r["scratch"] is None:
            continue
        seen += 1
        if r["scratch"] != 0:
            raise RuntimeError("%s: kernel %s uses %d bytes of scratch per lane (%s spilled VGPRs); this kernel requires a spill-free build"
                               % (src, fn, r["scratch"], r["spill"]))
    if seen == 0:
        raise RuntimeError("%s: no resource-usage remark found for %s" % (src, ", ".join(names)))


def build(force=False, verbose=True, resources=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    report = {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or resources or _stale(o, [s] + headers):
            guarded = src in NO_SCRATCH or resources
            cmd = [hipcc] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if guarded else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, src, o, guarded, subprocess.Popen(cmd, stderr=subprocess.PIPE if guarded else None, text=True if guarded else None)))
    failures = []
    isa = []
    for cmd, src, o, guarded, p in procs:
        if src in ASYNC_LOADS:
            c2 = [hipcc] + [f for f in FLAGS if f != "-fPIC"] + ["-Wno-unused-command-line-argument", "--cuda-device-only", "-S", "-o", "-", os.path.join(CSRC, src)]
            isa.append((src, subprocess.Popen(c2, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)))
    for cmd, src, o, guarded, p in procs:
        err = p.communicate()[1] if guarded else None
        if p.wait() != 0:
            if err:
                sys.stderr.write(err)
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if guarded:
            report[src] = parse_resources(err)
        if src in NO_SCRATCH:
            try:
                _check_no_scratch(src, NO_SCRATCH[src], err)
            except RuntimeError as e:
                os.remove(o)
                failures.append(str(e))
    for src, p in isa:
        text = p.communicate()[0]
        try:
            if p.returncode != 0:
                raise RuntimeError("%s: hipcc -S failed" % src)
            if check_async_loads(text, *ASYNC_LOADS[src]) == 0:
                raise RuntimeError("%s: no `%s` found in the ISA" % (src, ASYNC_LOADS[src][0]))
        except RuntimeError as e:
            o = os.path.join(CSRC, src.replace(".hip", ".o"))
            if os.path.exists(o):
                os.remove(o)
            failures.append(str(e))
    if failures:
        raise RuntimeError("\n".join(failures))
    # dd_version.o carries the source hash: rebuilt whenever the hash it was compiled with differs
    digest = source_hash()
    vo, stamp = os.path.join(CSRC, VERSION_SRC.replace(".hip", ".o")), os.path.join(CSRC, ".source_hash")
    old = open(stamp).read().strip() if os.path.exists(stamp) else ""
    version_rebuilt = False
    if force or old != digest or _stale(vo, [os.path.join(CSRC, VERSION_SRC)]):
        cmd = [hipcc] + FLAGS + ['-DDD_SOURCE_HASH="%s"' % digest, "-c", os.path.join(CSRC, VERSION_SRC), "-o", vo]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, "w") as f:
            f.write(digest + "\n")
        version_rebuilt = True
    objs.append(vo)
    if force or procs or version_rebuilt or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if resources:
        return report
    return LIB


def print_resources(report, only_spills=False):
    for src in sorted(report):
        for fn, r in sorted(report[src].items()):
            if only_spills and not r["scratch"]:
                continue
            print("%-22s %-110s vgpr %3s agpr %3s spill %3s scratch %4s occ %s lds %s" % (src, fn[:110], r["vgprs"], r["agprs"], r["spill"], r["scratch"],
                                                                                          r["occupancy"], r["lds"]))


if __name__ == "__main__":
    if "--resources" in sys.argv:
        print_resources(build(resources=True, verbose=False), only_spills="--spills" in sys.argv)
    else:
        print(build(force="--force" in sys.argv))
