"""Condense rocprofv3 --pmc passes (counter_collection CSVs) into profiles/<tag>_pmc_fetch_write.csv:
mean FETCH_SIZE / WRITE_SIZE per launch and kernel (KB, raw counter units).
usage: summarize_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <out.csv>"""
import csv, glob, os, re, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def collect(d, counter):
    per = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch -> value (summed over counter instances)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] == counter:
                per[short(row["Kernel_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: (len(v), sum(v.values()) / len(v)) for k, v in per.items()}


f, w = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
with open(sys.argv[3], "w") as out:
    out.write("kernel,dispatches,FETCH_SIZE_mean_KB_raw,WRITE_SIZE_mean_KB_raw\n")
    for k in sorted(f):
        out.write(f"{k},{f[k][0]},{f[k][1]:.3f},{w.get(k, (0, 0.0))[1]:.3f}\n")
print(open(sys.argv[3]).read())
