"""In-situ A/B of kernel choices by KERNEL TIME (rocprofv3 --kernel-trace), not by wall clock: the gpurun boxes' host CPUs differ by
2x in launch rate, which a chain of 5-40 us launches sees directly, and a device timer around the calls sees the host too.
For every variant (a set of environment variables) the tool is run under rocprofv3; the per-call kernel time is the sum of all
libpcy kernel durations divided by the count of an anchor kernel that runs once per call.

usage: insitu_ab.py esm|prefill  "LABEL:ENV=VAL ENV2=VAL2"  "LABEL2:..."  ...
"""
import glob, os, re, shutil, sqlite3, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
what = sys.argv[1]
tool, anchor = {"esm": ("tools/bench_esm1.py", "esm_embed_kernel"), "prefill": ("tools/bench_prefill1.py", "copy_rows_kernel")}[what]
for spec in sys.argv[2:]:
    label, _, envs = spec.partition(":")
    env = dict(os.environ, TMPDIR="/tmp")
    for kv in envs.split():
        k, _, v = kv.partition("=")
        env[k] = v
    out = f"/tmp/insitu_{label}"
    shutil.rmtree(out, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--", sys.executable, os.path.join(ROOT, tool)], env=env, cwd="/tmp",
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    dbs = glob.glob(out + "/**/*.db", recursive=True)
    if not dbs:
        print(f"{label}: no trace"); continue
    db = sqlite3.connect(dbs[0])
    rows = list(db.execute("select name, start, end from kernels order by start"))
    calls = sum(1 for n, _, _ in rows if anchor in n)
    agg = {}
    for n, s, e in rows:
        if "at::native" in n or "rocclr" in n:
            continue
        nm = re.sub(r"\(anonymous namespace\)::", "", n)
        nm = re.sub(r"\(.*", "", nm).replace("void ", "")
        a = agg.setdefault(nm, [0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    print(f"=== {label} [{envs}]: {tot / max(calls, 1) / 1e3:.3f} ms of kernels per call ({calls} calls)")
    for nm, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:9]:
        print(f"    {nm[:70]:70s} {c / max(calls, 1):6.1f}/call  avg {t / c:7.2f} us  {t / max(calls, 1):8.1f} us/call")
    shutil.rmtree(out, ignore_errors=True)
