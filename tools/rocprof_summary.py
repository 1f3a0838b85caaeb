"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into a per-kernel table (calls, total/avg/min/max us, %).
    python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db > profiles/<name>.txt"""
import re
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name) if len(name) > 120 else name
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values()) or 1
    print("%-100s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-100s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (name[:100], a[0], a[1] / 1e3, a[1] / 1e3 / a[0], a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
    print("TOTAL kernel time: %.1f us over %d dispatches" % (total / 1e3, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
