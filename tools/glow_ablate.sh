# timing ablations of tfk_glow_coupling (TFK_GLOW_SKIP: 1 no S0, 2 no conv blocks, 4 no Linear / transform)
N=${1:-65536}
for s in 0 6 5 3 7 1 2 4; do echo "== TFK_GLOW_SKIP=$s"; TFK_GLOW_SKIP=$s timeout -k 10 200 python tools/glow_fused_probe.py $N 2 2>&1 | grep -E "step  0|step  4|step 15|sum of"; done
