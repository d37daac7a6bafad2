#!/bin/bash
# bench.py (timed legs only) once per libtfk variant: bash tools/variant_bench.sh <out dir> [bench args] -- name1 name2 ...
OUT=$1; shift
ARGS=()
while [ "$1" != "--" ] && [ -n "$1" ]; do ARGS+=("$1"); shift; done
shift
mkdir -p $OUT
for V in base "$@"; do
  if [ "$V" == "base" ]; then unset TORCHFLOWS_AMD_LIB; else export TORCHFLOWS_AMD_LIB=$PWD/torchflows_amd/lib/variants/libtfk_$V.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-sample --no-train --steps 50 --warmup 5 "${ARGS[@]}" > $OUT/$V.json 2> $OUT/$V.err || { echo "$V failed"; tail -3 $OUT/$V.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$V.json"))
print("$V", "value %.4e"%d["value"], "median_ms", d["step_stats"]["median_ms"], "kernel avg_us", d["roofline"]["avg_us"], "parity n/a")
PY
done
