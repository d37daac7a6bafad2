#!/bin/bash
rows=${1:-65536}
for cfg in "0 0" "128 0" "128 1" "128 2" "256 1" "256 2" "256 4"; do
  set -- $cfg
  echo "== BLOCK=$1 SLOTS=$2"
  TORCHFLOWS_AMD_GLOW_LEVELS=0 TORCHFLOWS_AMD_DEBUG=glow_block=$1,glow_slots=$2 python tools/glow_fused_probe.py $rows 3 2>&1 | grep -E "sum of|step  0|step  3|step  6|step  9|step 12|step 15" | sed -e "s/cg 4\/4//"
done
