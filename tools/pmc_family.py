"""Aggregate one rocprofv3 --pmc CSV per kernel family: mean counter value per dispatch.
    python tools/pmc_family.py <dir> <substring> [<substring> ...]"""
import collections
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
if not f:
    raise SystemExit("no counter_collection.csv under " + sys.argv[1])
for pat in sys.argv[2:]:
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[-1])):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-28s %-24s mean %14.1f  sum %16.1f  dispatches %d" % (pat, k, sum(v) / len(v), sum(v), len(v)))
