"""A/B of the cfg-2 training step under environment variants, one bench.py child process each (same box, same call):
    python tools/ab_bench.py "name:VAR=1,VAR2=0" "other:" ...   [--inference] [--repeat N]
prints ms per step and the per-family milliseconds of bench.py's profiling pass."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
repeat = int(sys.argv[sys.argv.index("--repeat") + 1]) if "--repeat" in sys.argv else 1
mode = ["--mode", "inference"] if "--inference" in sys.argv else []
if "--repeat" in sys.argv:
    args = [a for a in args if a != sys.argv[sys.argv.index("--repeat") + 1]]
for rep in range(repeat):
    for spec in args:
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline"] + mode, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-2000:])
            continue
        d = json.loads(line[-1])
        r = d.get("roofline", {})
        fam = dict(r.get("other_kernels_ms_per_step", {}))
        for k, v in r.get("mfma_families", {}).items():
            fam[k] = v["ms_per_step"]
        print("%-24s %8.3f ms  %9.1f %s  %s" % (name, d.get("ms_per_step", d.get("ms_per_frame", 0.0)), d["value"], d["unit"],
                                                 " ".join("%s=%.3f" % kv for kv in sorted(fam.items()))), flush=True)
