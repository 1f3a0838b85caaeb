"""Issue-level decomposition of the hot kernels from two rocprofv3 --pmc passes (tools/pmc_sq_table.sh): per kernel (template arguments kept),
mean per dispatch of the SQ wave-state counters and what the guide's disjoint decomposition says (MI355X_MICROARCH.md, "rocprofv3 PMC slots"):
    WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA pipe / operand hazards) + ACTIVE_INST_ANY ~ WAVE_CYCLES.
    python tools/pmc_sq_table.py <dir of pass 1> <dir of pass 2> [name substrings...]"""
import collections
import csv
import glob
import re
import sys


def load(d):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[-1])):
        name = re.sub(r"^void ", "", r["Kernel_Name"]).replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*$", "", name)
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


a, b = load(sys.argv[1]), load(sys.argv[2])
pats = sys.argv[3:] or ["conv_rw", "conv_bwd", "compose", "conv_igemm_ws", "head_", "wgrad"]
print("%-58s %5s %9s | %6s %6s %6s %6s | %7s %7s %7s | %8s %8s" % ("kernel", "calls", "wave-cyc", "parked", "stall", "active", "stlLDS", "VALU/w", "SALU/w", "LDS/w", "MFMAbusy", "LDSconfl") + " %8s" % "LDSbusy")
for name in sorted(a):
    if not any(p in name for p in pats):
        continue
    A, B = a[name], b.get(name, {})
    m = lambda d, k: (sum(d[k]) / len(d[k])) if k in d and d[k] else float("nan")
    wc = m(A, "SQ_WAVE_CYCLES")
    waves = m(B, "SQ_WAVES")
    busy = m(B, "SQ_BUSY_CU_CYCLES")
    print("%-58s %5d %9.3g | %5.1f%% %5.1f%% %5.1f%% %5.1f%% | %7.0f %7.0f %7.0f | %7.1f%% %7.1f%% %7.1f%%" % (
        name[:58], len(A["SQ_WAVE_CYCLES"]), wc,
        100 * m(A, "SQ_WAIT_ANY") / wc, 100 * m(A, "SQ_WAIT_INST_ANY") / wc, 100 * m(A, "SQ_ACTIVE_INST_ANY") / wc, 100 * m(A, "SQ_WAIT_INST_LDS") / wc,
        m(B, "SQ_INSTS_VALU") / waves, m(B, "SQ_INSTS_SALU") / waves, m(B, "SQ_INSTS_LDS") / waves,
        100 * m(B, "SQ_VALU_MFMA_BUSY_CYCLES") / (4 * busy) if busy == busy else float("nan"),
        100 * m(B, "SQ_LDS_BANK_CONFLICT") / m(B, "SQ_LDS_IDX_ACTIVE") if m(B, "SQ_LDS_IDX_ACTIVE") else float("nan"),
        100 * m(B, "SQ_LDS_IDX_ACTIVE") / busy if busy == busy else float("nan")))
print("columns: parked = SQ_WAIT_ANY, stall = SQ_WAIT_INST_ANY, active = SQ_ACTIVE_INST_ANY, stlLDS = SQ_WAIT_INST_LDS, each / SQ_WAVE_CYCLES; per-wave instruction counts;")
print("         MFMAbusy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES); LDSconfl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; LDSbusy = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES (LDS-array cycles per CU cycle)")
