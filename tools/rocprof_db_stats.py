#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 run that wrote its default rocpd database (``-d DIR -o NAME`` -> DIR/NAME_results.db):
   python tools/rocprof_db_stats.py gpurun_out/imgfit_prof/imgfit_results.db [rows]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
agg = collections.defaultdict(lambda: [0, 0])
n = 0
for name, start, end in db.execute("select name, start, end from kernels"):
    agg[name][0] += 1
    agg[name][1] += end - start
    n += 1
total = sum(v[1] for v in agg.values())
print(f"total kernel time {total / 1e6:.2f} ms in {n} launches")
print("kernel,calls,avg_us,total_ms,percent")
for name, (calls, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"\"{name[:150]}\",{calls},{ns / calls / 1e3:.2f},{ns / 1e6:.3f},{100.0 * ns / total:.2f}")
