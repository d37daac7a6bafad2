#!/bin/bash
# per-phase ablation of the level launches (TFK_GLOW_SKIP bits: 1 S0, 2 conv blocks, 4 Linear + transform, 8 background cells,
# 16 S4); the blob is packed with the flag, so every run recompiles the program:  tools/glow_level_ablate.sh [rows]
rows=${1:-65536}
for skip in 0 1 2 4 8 16 31; do
  echo "== TFK_GLOW_SKIP=$skip"
  TFK_GLOW_SKIP=$skip python tools/glow_level_probe.py $rows 3 2>&1 | grep "level of"
done
