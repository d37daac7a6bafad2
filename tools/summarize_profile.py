#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile.sh into two small tables:
<dir>/kernel_stats.csv (calls, average duration) and <dir>/pmc_summary.csv (per-kernel means of
HBM bytes -- FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, both in KB -- and SQ activity)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
stats = glob.glob(os.path.join(out, "trace", "*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,avg_us,total_ms,percent\n")
        for r in rows:
            f.write('"%s",%s,%.2f,%.3f,%s\n' % (r["Name"][:160], r["Calls"], float(r["AverageNs"]) / 1e3,
                                               float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:160]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
with open(os.path.join(out, "pmc_summary.csv"), "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + ",hbm_bytes_per_launch\n")
    for k, v in agg.items():
        n = max(len(x) for x in v.values())
        if n < 2:
            continue
        mean = {c: (sum(x) / len(x)) for c, x in v.items()}
        hbm = (2 * mean.get("FETCH_SIZE", 0) + mean.get("WRITE_SIZE", 0)) * 1024
        f.write('"%s",%d,%s,%.0f\n' % (k, n, ",".join("%.1f" % mean.get(c, float("nan")) for c in names), hbm))
print(open(os.path.join(out, "pmc_summary.csv")).read()[:3000])

# the sources these counters were taken from (bench.py withholds roofline.traffic / valu_insts when they differ)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:
    from bench import csrc_sha256
    open(os.path.join(out, "csrc_sha256.txt"), "w").write(csrc_sha256() + "\n")
except Exception as exc:                                     # noqa: BLE001
    print("csrc hash not written:", exc)
