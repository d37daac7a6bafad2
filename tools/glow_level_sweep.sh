#!/bin/bash
# launch-shape sweep of the level launches:  tools/glow_level_sweep.sh [rows]
rows=${1:-65536}
for blk in 256 512 1024; do
  for cg in "4 4" "2 2" "8 4"; do
    set -- $cg
    echo "== BLOCK=$blk CG1=$1 CG2=$2"
    TORCHFLOWS_AMD_GLOW_LEVELS=1 TORCHFLOWS_AMD_DEBUG=glow_level_block=$blk,glow_cg1=$1,glow_cg2=$2 python tools/glow_level_probe.py 65536 3 2>&1 | grep -E "level of|sum of" | sed -e "s/{'samples'.*'D_level': [0-9]*}//"
  done
done
