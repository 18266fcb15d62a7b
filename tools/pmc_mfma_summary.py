"""Summarise a rocprofv3 --pmc run (CSV counter_collection) per kernel: launches, mean duration, mean counter values, and --
when SQ_INSTS_VALU_MFMA_MOPS_BF16 was collected -- the hardware-counted MFMA rate (MOPS x 512 FLOP / duration) and its share
of the 2.5 PFLOP/s dense bf16 peak; with SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE the matrix-pipe busy share
(busy cycles / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs): GRBM_GUI_ACTIVE arrives summed over the 8 XCDs -- 9.66 M per 543 us launch
= 8 x 2.22 GHz)."""
import csv, glob, json, re, sys
from collections import defaultdict
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for f in files:
    for r in csv.DictReader(open(f)):
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        if "gemm" not in nm and "attn" not in nm:
            continue
        nm = nm[:80]
        acc[nm][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[nm][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {}
for nm, cs in acc.items():
    d = sum(dur[nm].values()) / len(dur[nm])
    row = {"launches": len(dur[nm]), "avg_us": round(d, 1)}
    for c, v in cs.items():
        row[c] = sum(v) / len(v)
    if "SQ_INSTS_VALU_MFMA_MOPS_BF16" in row:
        tf = row["SQ_INSTS_VALU_MFMA_MOPS_BF16"] * 512 / (d * 1e-6) / 1e12
        row["mfma_TFLOPs_counted"] = round(tf, 1)
        row["mfma_share_of_2.5PF"] = round(tf / 2500, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row and row["GRBM_GUI_ACTIVE"] > 0:
        row["mfma_pipe_busy_share"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["GRBM_GUI_ACTIVE"] / 8 * 1024), 3)
        row["shader_clock_GHz"] = round(row["GRBM_GUI_ACTIVE"] / 8 / d / 1e3, 2)
    out[nm] = row
print(json.dumps(out, indent=1))
