"""HBM bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counter_collection CSVs):
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- on gfx950 FETCH_SIZE reports half the bytes of a 16-B/lane coalesced stream
(MI355X_MICROARCH.md, HBM section).  usage: pmc_hbm_summary.py <dir fetch pass> <dir write pass> <kernel substring> <algorithmic bytes> <what>"""
import csv, glob, json, re, sys
from collections import defaultdict


def mean_counter(path, counter, sub):
    vals = defaultdict(float)
    name = None
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
                name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    v = sorted(vals.values())
    v = v[len(v) // 10: len(v) - len(v) // 10] if len(v) > 20 else v   # drop warm-up / outlier dispatches
    return name, sum(v) / len(v), len(vals)


def summarise(fetch_dir, write_dir, sub, alg, what):
    name, fetch, n1 = mean_counter(fetch_dir, "FETCH_SIZE", sub)
    _, write, n2 = mean_counter(write_dir, "WRITE_SIZE", sub)
    hbm = int((2 * fetch + write) * 1024)
    return {"note": "rocprofv3 --pmc FETCH_SIZE (pass 1) / --pmc WRITE_SIZE (pass 2) --kernel-trace -- python tools/pmc_decode.py; "
                    "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE on gfx950 reports half the bytes of a 16-B/lane coalesced stream "
                    "(MI355X_MICROARCH.md, HBM section)",
            "kernels": {name: {"what": what, "launches": [n1, n2], "FETCH_SIZE_KiB": round(fetch, 1), "WRITE_SIZE_KiB": round(write, 1),
                               "hbm_bytes_per_launch": hbm, "algorithmic_bytes": alg, "ratio": round(hbm / alg, 4)}}}


if __name__ == "__main__":
    print(json.dumps(summarise(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]), indent=1))
