#!/usr/bin/env python3
"""gpurun_out/prof_r04_* (tools/profile_r04.sh on the GPU box) -> profiles/r04/ (what is committed) + profiles/traffic.json.

Keeps, per workload: csrc_sha256.txt, the first 24 rows of the kernel trace summary, the libtfk / hipBLASLt rows of the
PMC summary; the training probe's line; the two config-5 coupling launches under the SQ / TCC counters, from which the
glow32 entries of traffic.json are taken (2 x FETCH_SIZE + WRITE_SIZE, KB)."""
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles", "r04")


def head(src, dst, n):
    with open(src) as f:
        lines = f.readlines()[:n]
    with open(dst, "w") as f:
        f.writelines(lines)


def main():
    for name, pmc in (("lean", True), ("nsf", True), ("rnvp256", True), ("glow32", True), ("sample", False), ("train", False),
                      ("train_glow", False)):
        s, d = os.path.join(SRC, f"prof_r04_{name}"), os.path.join(DST, name)
        os.makedirs(d, exist_ok=True)
        shutil.copy(os.path.join(s, "csrc_sha256.txt"), os.path.join(d, "csrc_sha256.txt"))
        head(os.path.join(s, "kernel_stats.csv"), os.path.join(d, "kernel_stats.csv"), 25)
        if pmc:
            with open(os.path.join(s, "pmc_summary.csv")) as f:
                lines = f.readlines()
            keep = [lines[0]] + [ln for ln in lines[1:] if "tfk::" in ln or "Cijk" in ln]
            with open(os.path.join(d, "pmc_summary.csv"), "w") as f:
                f.writelines(keep)
    shutil.copy(os.path.join(SRC, "prof_r04_train", "probe.txt"), os.path.join(DST, "train", "probe.txt"))
    for f in ("probe.txt", "convtrain_bench.txt", "pmc_8192.txt", "pmc_1024.txt"):
        with open(os.path.join(SRC, "prof_r04_train_glow", f)) as fh:
            keep = [ln for ln in fh if "amdgpu.ids" not in ln]
        with open(os.path.join(DST, "train_glow", f), "w") as fh:
            fh.writelines(keep)
    glow = {}
    for step, label in (("step0", "affine 3x16x32"), ("step3", "conv1x1 6x16x16")):
        src = os.path.join(SRC, "prof_r04_glow32", f"{step}_pmc.txt")
        dst = os.path.join(DST, "glow32", f"{step}_{label.replace(' ', '_')}_pmc.txt")
        shutil.copy(src, dst)
        vals = {}
        for ln in open(src):
            m = re.match(r"\s+(\w+)\s+mean=\s*([\d.]+)", ln)
            if m:
                vals[m.group(1)] = float(m.group(2))
        key = f"glow_coupling[{label}]"
        glow[key] = int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
        glow[key + ":valu_insts"] = vals["SQ_INSTS_VALU"]
        glow[key + ":mfma_insts"] = vals["SQ_INSTS_VALU_MFMA_F32"]
        glow[key + ":grbm_gui_active"] = round(vals["GRBM_GUI_ACTIVE"], 1)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "update_traffic.py"), "realnvp64=profiles/r04/lean",
                    "nsf64=profiles/r04/nsf", "realnvp256=profiles/r04/rnvp256"], check=True, cwd=ROOT)
    p = os.path.join(ROOT, "profiles", "traffic.json")
    db = json.load(open(p))
    db["glow32"] = glow
    db["source"]["glow32"] = ("profiles/r04/glow32/step0_affine_3x16x32_pmc.txt, step3_conv1x1_6x16x16_pmc.txt "
                              "(tools/glow_pmc.sh; 131 072-row launches = bench.py's chunk): 2 x FETCH_SIZE + WRITE_SIZE (KB)")
    json.dump(db, open(p, "w"), indent=1)
    print("traffic.json:", db["csrc_sha256"][:16], {k: v for k, v in glow.items() if ":" not in k})


if __name__ == "__main__":
    main()
