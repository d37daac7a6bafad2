#!/bin/bash
# where do the LDS bank-conflict cycles of the first checkerboard launch come from?  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
# with one phase removed at a time (TFK_GLOW_SKIP: 1 S0, 2 conv blocks, 4 Linear + transform):  bash tools/glow_conflicts.sh [step] [rows]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
STEP=${1:-0}; ROWS=${2:-65536}
for skip in 0 1 2 4 6 7; do
  out=gpurun_out/glowconf_$skip
  rm -rf $out; mkdir -p $out
  TFK_GLOW_SKIP=$skip TORCHFLOWS_AMD_GLOW_LEVELS=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --kernel-trace --output-format csv -d $out -o p -- python tools/glow_step_bench.py $STEP $ROWS 3 > $out/log.txt 2> $out/err.txt || { tail -3 $out/err.txt; exit 1; }
  python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("$out/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "glow" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k,v in agg.items()}
print("skip=$skip", {k: round(v/1e6,2) for k,v in sorted(m.items())}, "conflict share %.3f" % (m.get("SQ_LDS_BANK_CONFLICT",0)/max(m.get("SQ_LDS_IDX_ACTIVE",1),1)))
PY
  tail -1 $out/log.txt
  rm -rf $out
done
