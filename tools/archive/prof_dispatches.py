"""List per-dispatch durations of kernels matching a pattern from a rocprofv3 .db, averaged in chunks."""
import sqlite3, sys, glob
path, pat, chunk = sys.argv[1], sys.argv[2], int(sys.argv[3])
dbp = glob.glob(path + "/**/*.db", recursive=True)[0]
db = sqlite3.connect(dbp)
cols = [d[1] for d in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, end from kernels order by start"))
d = [(e - s) / 1e3 for n, s, e in rows if pat in n]
for i in range(0, len(d), chunk):
    c = d[i:i + chunk]
    print(f"launches {i:5d}-{i+len(c):5d}: avg {sum(c)/len(c):8.2f} us  min {min(c):8.2f}  max {max(c):8.2f}")
