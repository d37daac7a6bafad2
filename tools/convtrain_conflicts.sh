cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pm -o p -- python tools/convtrain_bench.py 8192 > /tmp/pm.log 2>&1
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    if "k_ct_" in k and "bwd" in k:
        m={c:sum(x)/len(x) for c,x in v.items()}
        print(k[:55], "bank %.0f%%" % (100*m["SQ_LDS_BANK_CONFLICT"]/max(m["SQ_LDS_IDX_ACTIVE"],1)), "conflict cycles %.1fM of wave cycles %.0fM" % (m["SQ_LDS_BANK_CONFLICT"]/1e6, m["SQ_WAVE_CYCLES"]/1e6))
PY
