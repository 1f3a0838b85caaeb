import csv, glob, collections, sys
f = sorted(glob.glob(sys.argv[1] + "/*counter_collection.csv"))
if not f:
    print("no csv"); raise SystemExit
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[-1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-32s %14.0f  (per dispatch, %d dispatches)" % (k, sum(v) / len(v), len(v)))
