"""Timeline of the LAST `n` kernel dispatches of a rocprofv3 --kernel-trace .db: per kernel name the mean duration and the mean idle
gap in front of it (start - end of the previous dispatch), plus the totals -- where a chain of short dependent launches loses time
between kernels rather than inside them.  usage: prof_gaps.py <dir-or-db> <n> [name-filter-of-the-window's-first-kernel]"""
import collections, glob, re, sqlite3, sys
path, n = sys.argv[1], int(sys.argv[2])
dbp = path if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)[0]
db = sqlite3.connect(dbp)
rows = list(db.execute("select name, start, end from kernels order by start"))[-n:]
agg = collections.OrderedDict()
tk = tg = 0.0
for i, (name, s, e) in enumerate(rows):
    nm = re.sub(r"\(anonymous namespace\)::", "", name)
    nm = re.sub(r"void at::native::(\w+).*", r"at::native::\1<...>", nm)[:90]
    gap = (s - rows[i - 1][2]) / 1e3 if i else 0.0
    a = agg.setdefault(nm, [0, 0.0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3; a[2] += gap
    tk += (e - s) / 1e3; tg += gap
print("name,calls,avg_us,avg_gap_before_us")
for nm, (c, d, g) in agg.items():
    print(f'"{nm}",{c},{d / c:.2f},{g / c:.2f}')
print(f"# window of {len(rows)} dispatches: kernels {tk:.1f} us, gaps {tg:.1f} us, span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us")
