"""Timeline of one decode step from a rocprofv3 kernel trace (.db under the given directory): start offset, duration and
name of every launch between two consecutive embed launches, time with >= 1 and >= 2 kernels active."""
import glob
import sqlite3
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -3
dbp = glob.glob(path + "/**/*.db", recursive=True)[0]
db = sqlite3.connect(dbp)
cols = [d[1] for d in db.execute("pragma table_info(kernels)")]
extra = [c for c in ("queue_id", "stream_id", "stream") if c in cols]
rows = list(db.execute(f"select name, start, end{''.join(', ' + c for c in extra)} from kernels order by start"))
emb = [i for i, r in enumerate(rows) if "embed_gather" in r[0]]
print("columns:", cols)
print("kernels", len(rows), "embed launches", len(emb))
a, b = emb[which], emb[which + 1]
step = rows[a:b]
t0 = step[0][1]
print(f"step of {len(step)} launches, {(rows[b][1] - t0) / 1e3:.1f} us between embed launches")
ev = []
for r in step:
    ev += [(r[1], 1), (r[2], -1)]
ev.sort()
act, last, busy1, busy2 = 0, t0, 0, 0
for t, dlt in ev:
    if act >= 1: busy1 += t - last
    if act >= 2: busy2 += t - last
    act += dlt; last = t
print(f">=1 active {busy1 / 1e3:.1f} us, >=2 active {busy2 / 1e3:.1f} us")
short = lambda n: n.replace("void ", "").split("(")[0][:60]
for i, r in enumerate(step[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]):
    print(f"{(r[1] - t0) / 1e3:9.2f} +{(r[2] - r[1]) / 1e3:7.2f} us  {' '.join(str(x) for x in r[3:])}  {short(r[0])}")
print("...")
for r in step[-8:]:
    print(f"{(r[1] - t0) / 1e3:9.2f} +{(r[2] - r[1]) / 1e3:7.2f} us  {' '.join(str(x) for x in r[3:])}  {short(r[0])}")
