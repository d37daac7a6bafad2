#!/bin/bash
# PMC passes for one bench workload: bash tools/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ... -- [bench args]
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; shift
PASSES=()
while [ "$1" != "--" ] && [ -n "$1" ]; do PASSES+=("$1"); shift; done
[ "$1" == "--" ] && shift
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG/p$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_$TAG/p$i.json 2> gpurun_out/pmc_$TAG/p$i.err || { tail -5 gpurun_out/pmc_$TAG/p$i.err; exit 1; }
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_$TAG/p*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if any(len(x)>=2 for x in v.values()) and ("tfk" in k or "Cijk" in k):
        print(k)
        for c,x in sorted(v.items()): print(f"   {c:28s} mean={sum(x)/len(x):16.1f} n={len(x)}")
PY
