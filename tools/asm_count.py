#!/usr/bin/env python3
"""Instruction mix of the kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only): how many MFMA, plain VALU,
packed VALU, transcendental, LDS, global-memory and scalar instructions each kernel holds, in total and inside its
innermost loops.  Usage: tools/asm_count.py file.s [substring of the kernel's mangled name]."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "valu_packed"
    if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)_", op):
        return "valu_trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier"):
        return "sync"
    if op.startswith("s_"):
        return "salu"
    return None


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        c = collections.Counter()
        for line in body.split("\n"):
            mm = re.match(r"\s+([a-z_0-9]+)\b", line)
            if mm:
                k = classify(mm.group(1))
                if k:
                    c[k] += 1
        print(name[:100])
        print("   ", dict(sorted(c.items())))


if __name__ == "__main__":
    main()
