"""Print per-kernel stats from a rocprofv3 results .db (top_kernels view) as CSV."""
import sqlite3, sys, re, glob
path = sys.argv[1]
dbs = glob.glob(path + "/**/*.db", recursive=True) if not path.endswith(".db") else [path]
for dbp in dbs:
    db = sqlite3.connect(dbp)
    print("name,calls,total_us,avg_us,percent")
    for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        nm = re.sub(r"\(anonymous namespace\)::", "", r[0])
        nm = re.sub(r"void at::native::(\w+).*", r"at::native::\1<...>", nm)
        print('"%s",%d,%.1f,%.3f,%.2f' % (nm[:110], r[1], r[2], r[3], r[4]))
