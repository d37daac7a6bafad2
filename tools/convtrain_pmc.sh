#!/bin/bash
# PMC passes over the ConvNet training launches (tools/convtrain_bench.py N): bash tools/convtrain_pmc.sh <tag> <N>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
TAG=$1; N=${2:-8192}
mkdir -p gpurun_out/ctpmc_$TAG
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" \
         "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d gpurun_out/ctpmc_$TAG/p$i -o p -- python tools/convtrain_bench.py $N > gpurun_out/ctpmc_$TAG/p$i.log 2> gpurun_out/ctpmc_$TAG/p$i.err || { tail -5 gpurun_out/ctpmc_$TAG/p$i.err; exit 1; }
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/ctpmc_$TAG/p*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    if "k_ct_" in k:
        print(k)
        for c,x in sorted(v.items()): print(f"   {c:28s} mean={sum(x)/len(x):16.1f} n={len(x)}")
PY
cat gpurun_out/ctpmc_$TAG/p1.log | grep -v amdgpu.ids
rm -rf gpurun_out/ctpmc_$TAG/p[0-9]
