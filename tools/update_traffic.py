#!/usr/bin/env python3
"""profiles/traffic.json from the pmc_summary.csv files of tools/profile.sh runs:

    python tools/update_traffic.py realnvp64=profiles/r02/mfma nsf64=profiles/r02/nsf_mfma ...

Per workload and libtfk entry point (bench.py's kernel names): HBM bytes per launch (FETCH_SIZE x 2 gfx950 correction
+ WRITE_SIZE), SQ_INSTS_VALU and -- for the matrix-core flow programs -- the measured MFMA instruction count and
matrix-pipe cycles.  The file carries the sha256 of the kernel sources the PMC runs were taken from; all directories
given must carry the same csrc_sha256.txt (written by tools/summarize_profile.py), and bench.py ignores the file when it
differs from the sources in the tree."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = [("k_flow_chain", "flow_run_mfma"), ("k_flow_rqs_chain", "flow_run_mfma"), ("k_flow_run_mfma", "flow_run_mfma"), ("k_flow_run<", "flow_run"), ("k_affine_half_v4", "affine_coupling[inplace]"),
           ("k_rqs_coupling_dma", "rqs_coupling[inplace]"), ("k_elementwise_affine<", "elementwise_affine"),
           ("k_permute", "permute"), ("k_diag_gauss<", "diag_gauss_logprob"),
           ("k_conv3x3_relu_pool_affine", "conv3x3_relu_pool_affine")]


def main():
    db = {"_comment": "per launch, from rocprofv3 PMC passes (tools/profile.sh): HBM bytes (FETCH_SIZE x2 gfx950 correction + "
                      "WRITE_SIZE, KB units); '<kernel>:valu_insts' SQ_INSTS_VALU; ':mfma_insts' SQ_INSTS_VALU_MFMA_F32; "
                      "':mfma_busy_cycles' SQ_VALU_MFMA_BUSY_CYCLES; ':mfma_coexec_cycles' SQ_VALU_MFMA_COEXEC_CYCLES; "
                      "':sq_busy_cycles' SQ_BUSY_CYCLES; default rows per GPU", "source": {}}
    shas = set()
    for arg in sys.argv[1:]:
        wl, d = arg.split("=")
        shas.add(open(os.path.join(ROOT, d, "csrc_sha256.txt")).read().strip())
        db["source"][wl] = d + "/pmc_summary.csv"
        rows = list(csv.DictReader(open(os.path.join(ROOT, d, "pmc_summary.csv"))))
        ent = db.setdefault(wl, {})
        for r in rows:
            name = next((v for k, v in KERNELS if k in r["kernel"]), None)
            if name is None:
                continue
            # several instantiations of one kernel may appear: keep the one with the most dispatches
            if name in ent and int(r["dispatches"]) <= ent.get("_n:" + name, 0):
                continue
            ent["_n:" + name] = int(r["dispatches"])
            ent[name] = int(float(r["hbm_bytes_per_launch"]))
            for col, key in (("SQ_INSTS_VALU", "valu_insts"), ("SQ_INSTS_VALU_MFMA_F32", "mfma_insts"), ("SQ_INSTS_VALU_MFMA_BF16", "mfma_bf16_insts"),
                             ("SQ_VALU_MFMA_BUSY_CYCLES", "mfma_busy_cycles"),
                             ("SQ_VALU_MFMA_COEXEC_CYCLES", "mfma_coexec_cycles"), ("SQ_BUSY_CYCLES", "sq_busy_cycles"),
                             ("GRBM_GUI_ACTIVE", "grbm_gui_active"), ("SQ_INSTS_VALU_TRANS_F32", "trans_insts")):
                v = r.get(col)
                if v not in (None, "", "nan"):
                    ent[f"{name}:{key}"] = float(v)
        for k in [k for k in ent if k.startswith("_n:")]:
            del ent[k]
    assert len(shas) == 1, f"PMC runs from different kernel sources: {shas}"
    db["csrc_sha256"] = shas.pop()
    json.dump(db, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(db, indent=1)[:2000])


if __name__ == "__main__":
    main()
