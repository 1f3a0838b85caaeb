"""Copy the artefacts of one tools/refresh_profiles.sh run from gpurun_out/<tag>/ into profiles/ (tracked) and derive the two JSON
summaries bench.py and DESIGN.md cite:  <tag>_hbm_traffic.json (FETCH_SIZE / WRITE_SIZE per launch, gfx950 correction applied) and
<tag>_mfma_busy.json (SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CU_CYCLES per kernel family).
    python tools/collect_profiles.py <tag> "<one-line description of the state>" """
import json
import os
import re
import shutil
import sys

tag, note = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
for name in ("bench.json", "kernel_stats.txt", "inference.json", "host_inputs.txt", "pmc_FETCH_SIZE.txt", "pmc_WRITE_SIZE.txt",
             "pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt", "two_ranks_one_gpu.txt", "one_rank_rccl.txt", "inference_kernel_stats.txt", "cfg3_heavy_kernel_stats.txt",
             "cfg3_light_kernel_stats.txt", "cfg3_light_pmc_FETCH_SIZE.txt", "cfg3_light_pmc_WRITE_SIZE.txt", "cfg3_light_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt",
             "cfg3_heavy_pmc_FETCH_SIZE.txt", "cfg3_heavy_pmc_WRITE_SIZE.txt", "cfg3_heavy_pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt", "conv_launches.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        out = os.path.join(dst, "%s_%s" % (tag, name.replace("pmc_SQ_VALU_MFMA_BUSY_CYCLES", "pmc_MFMA_BUSY")))
        if name == "kernel_stats.txt":
            bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
            with open(out, "w") as f:
                f.write("# %s: rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-extras` (cfg-2, bf16, B=128 tiles of 128x128 per GPU, hipGraph;\n"
                        "#   5 warm-up + 20 timed steps + 4 per-op timing passes = 29 step-equivalents).  %s\n"
                        "# bench line of the same build without profiler: profiles/%s_bench.json (%.0f tiles/s)\n" % (tag, note, tag, bench["value"]))
                f.write(open(p).read())
        else:
            shutil.copy(p, out)


# every oracle comparison of the round's -m gpu session with its measured error and its gate (tests/conftest.py writes them to gpurun_out/),
# the child-process sessions of tests/test_gpu_fallbacks.py beside it (the round's parity log belongs in the tracked tree)
import glob
for pth in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "parity_errors*.txt"))):
    shutil.copy(pth, os.path.join(dst, "%s_%s" % (tag.split("_")[0], os.path.basename(pth))))
for name in ("rw_loop_ubench.txt", "lds_probe.txt", "sq_table.txt", "wgrad_only_knockouts.txt", "deterministic.txt", "example_profile.txt", "ab_round5.txt",
             "cfg3_light_launches.txt", "cfg3_heavy_launches.txt", "sq_table_cfg3_heavy.txt", "sq_table_cfg3_light.txt", "ks_shape_bench.txt",
             "ab_round6.txt", "b96_knockouts.txt", "b96_hbm.txt", "gate_diag.txt", "smoke.txt"):
    pth = os.path.join(src, name)
    if os.path.exists(pth):
        shutil.copy(pth, os.path.join(dst, "%s_%s" % (tag, name)))


def family(path):
    fam = {}
    for line in open(path):
        m = re.match(r"(\S+)\s+(\S+)\s+mean\s+([\d.]+)\s+sum\s+([\d.]+)\s+dispatches\s+(\d+)", line)
        if m:
            fam.setdefault(m.group(1), {})[m.group(2)] = (float(m.group(3)), int(m.group(5)))
    return fam


fetch, write = family(os.path.join(src, "pmc_FETCH_SIZE.txt")), family(os.path.join(src, "pmc_WRITE_SIZE.txt"))
traffic = {"how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (no trace options) over `python bench.py --no-cpu-baseline "
                  "--no-extras --no-graph --steps 2 --warmup 1` (7 step-equivalents); mean per dispatch of the kernel family (tools/pmc_family.py; "
                  "raw: profiles/%s_pmc_*.txt, values in kB).  FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); "
                  "WRITE_SIZE uncorrected (calibrated in round 1 on a 64->64 forward launch: 67.3 MB counted for 67.1 MB written)." % tag,
           "state": note, "tiles_per_gpu_per_step": 128, "dtype": "bf16"}
for k in fetch:
    kk = {"compose_fwd": "compose_fwd", "compose_bwd": "compose_bwd", "head_fwd": "head_fwd", "head_bwd": "head_bwd"}.get(k, k)
    traffic[kk] = {"fetch_MB_per_launch": round(2 * fetch[k]["FETCH_SIZE"][0] / 1e3, 1), "write_MB_per_launch": round(write[k]["WRITE_SIZE"][0] / 1e3, 1),
                   "dispatches": fetch[k]["FETCH_SIZE"][1]}
json.dump(traffic, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)

busy = family(os.path.join(src, "pmc_SQ_VALU_MFMA_BUSY_CYCLES.txt"))
out = {"how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE (one pass, no trace options) over the same "
              "command; per kernel family.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the fraction of the cycles a CU "
              "had waves resident during which a SIMD's matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, 16 per "
              "v_mfma_f32_16x16x32_bf16).  Kernels run serialised and at lower clocks under the profiler: ratios, not absolute times.",
       "state": note}
for k, v in busy.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CU_CYCLES" in v:
        out[k] = {"mfma_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (4.0 * v["SQ_BUSY_CU_CYCLES"][0]), 4),
                  "SQ_VALU_MFMA_BUSY_CYCLES_per_launch": v["SQ_VALU_MFMA_BUSY_CYCLES"][0], "SQ_BUSY_CU_CYCLES_per_launch": v["SQ_BUSY_CU_CYCLES"][0],
                  "mfma_instructions_per_launch": v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / 16.0, "dispatches": v["SQ_VALU_MFMA_BUSY_CYCLES"][1]}
json.dump(out, open(os.path.join(dst, tag + "_mfma_busy.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if isinstance(v, dict)}, indent=1))
