#!/usr/bin/env python3
"""Same-box A/B of two libtfk builds on the 19 coupling launches of AffineGlow((3,32,32)) (box-to-box spread on this pool is
larger than most kernel changes): python tools/ab_glow.py <libA.so> <libB.so> [rows] -- the image program is compiled once per
library in a child process each, alternating twice."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs, rows = sys.argv[1:3], (sys.argv[3] if len(sys.argv) > 3 else "65536")
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, TORCHFLOWS_AMD_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "glow_fused_probe.py"), rows, "3"], env=env,
                             capture_output=True, text=True, timeout=600).stdout
        print(os.path.basename(lib), [ln.strip() for ln in out.splitlines() if "sum of" in ln or "step  0" in ln or "step  3" in ln])
