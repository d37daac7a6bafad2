#!/usr/bin/env python3
"""Instruction histogram of one kernel from a hipcc --save-temps .s file, per basic block and in total:

    python tools/isa_hist.py <file.s> <kernel-name-substring> [min block size]

Classes: mfma, trans (v_exp / v_log / v_rcp / v_rsq / v_sqrt), mov (v_mov / v_accvgpr), sel (v_cndmask / v_cmp),
valu (every other v_*), lds, vmem, salu, wait (s_waitcnt / s_nop), branch."""
import collections
import re
import sys


def cls(i):
    if "mfma" in i:
        return "mfma"
    if re.match(r"v_(exp|log|rcp|rsq|sqrt)_", i):
        return "trans"
    if i.startswith("v_mov") or i.startswith("v_accvgpr"):
        return "mov"
    if i.startswith("v_cndmask") or i.startswith("v_cmp"):
        return "sel"
    if i.startswith("v_"):
        return "valu"
    if i.startswith("ds_"):
        return "lds"
    if i.startswith("global_") or i.startswith("buffer_") or i.startswith("flat_") or i.startswith("scratch_"):
        return "vmem"
    if i.startswith("s_waitcnt") or i.startswith("s_nop"):
        return "wait"
    if i.startswith("s_cbranch") or i == "s_branch":
        return "branch"
    if i.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    min_block = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], ("entry", [])
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = (m.group(1), [])
        else:
            t = l.strip()
            if t and not t.startswith(";") and not t.startswith("."):
                cur[1].append(t.split()[0])
    blocks.append(cur)
    total = collections.Counter()
    print(lines[start].split(":")[0])
    for bname, ins in blocks:
        c = collections.Counter(cls(i) for i in ins)
        total.update(c)
        if len(ins) >= min_block:
            print(f"  {bname:12s} {len(ins):5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
    print("  total        %5d  " % sum(total.values()) + "  ".join(f"{k}={v}" for k, v in sorted(total.items())))
    for l in lines[end:end + 40]:
        if re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize)", l):
            print("  " + l[2:])


if __name__ == "__main__":
    main()
