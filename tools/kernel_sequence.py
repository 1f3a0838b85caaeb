import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
pat = sys.argv[2]
seq = [(re.sub(r"\(anonymous namespace\)::", "", n)[:60], (e - s) / 1e3) for n, s, e in rows if re.search(pat, n)]
print(len(seq))
for n, d in seq[-24:]:
    print("%-62s %8.1f us" % (n, d))
