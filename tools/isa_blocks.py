#!/usr/bin/env python3
"""Splits a kernel's disassembly (tools/disasm_scene.py output) into basic blocks and prints, per block, its address
range, instruction-class counts and where it branches -- enough to find the hot loop and count its VALU work offline.
    python tools/disasm_scene.py sphere dcsg_sweep_sp_f > /tmp/k.s && python tools/isa_blocks.py /tmp/k.s"""
import re
import sys


def main(path):
    ins = []
    for line in open(path):
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), line))
    if not ins:
        sys.exit("no instructions")
    base = ins[0][0]
    targets = set()
    for a, op, args, line in ins:
        if op.startswith(("s_cbranch", "s_branch")):
            t = re.search(r"<[^>+]+\+0x([0-9a-f]+)>", line)
            if t:
                targets.add(base + int(t.group(1), 16))
    blocks, cur = [], []
    for i, (a, op, args, line) in enumerate(ins):
        if a in targets and cur:
            blocks.append(cur)
            cur = []
        cur.append((a, op, args, line))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_swappc", "s_setpc")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    for b in blocks:
        n = {"v": 0, "s": 0, "ds": 0, "mem": 0, "trans": 0, "f64": 0}
        for a, op, args, line in b:
            if op.startswith("v_"):
                n["v"] += 1
                if re.match(r"v_(sqrt|rcp|rsq|exp|log|sin|cos)_", op):
                    n["trans"] += 1
                if "_f64" in op:
                    n["f64"] += 1
            elif op.startswith("s_"):
                n["s"] += 1
            elif op.startswith("ds_"):
                n["ds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                n["mem"] += 1
        last = b[-1]
        t = re.search(r"<[^>+]+\+0x([0-9a-f]+)>", last[3])
        dest = (" -> +0x%x" % int(t.group(1), 16)) if t and last[1].startswith(("s_cbranch", "s_branch")) else ""
        print("+0x%05x..+0x%05x  %4d instr  valu %4d (trans %2d, f64 %2d)  salu %3d  lds %2d  vmem %2d  %s%s"
              % (b[0][0] - base, last[0] - base, len(b), n["v"], n["trans"], n["f64"], n["s"], n["ds"], n["mem"], last[1], dest))


if __name__ == "__main__":
    main(sys.argv[1])
