#!/usr/bin/env python3
"""Map of one kernel in a hipcc -S listing: where the barriers, branches, labels, scratch (spill) accesses and waits sit,
and the instruction mix between consecutive barriers.  usage: tools/asm_map.py file.s <substring of the kernel symbol>"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith('_ZN') and key in l and re.match(r'_ZN[^ ]*:', l))
end = next(i for i in range(start, len(txt)) if txt[i].startswith('.Lfunc_end'))
body = txt[start:end]
print(f"{txt[start]}  {len(body)} lines")
marks = []
for i, l in enumerate(body):
    t = l.strip()
    if re.match(r'\.LBB\d+_\d+:', t) or t.startswith('s_barrier') or t.startswith('s_cbranch') or t.startswith('s_branch'):
        marks.append((i, t.split(';')[0].strip()))
prev = 0
def mix(a, b):
    seg = [x.strip() for x in body[a:b]]
    c = lambda pat: sum(1 for x in seg if re.match(pat, x))
    return (f"mfma {c('v_mfma')} valu {c('v_(?!mfma)')} ds_read {c('ds_read')} ds_write {c('ds_write')} gload {c('global_load(?!_lds)')} "
            f"glds {c('global_load_lds')} gstore {c('global_store')} scratch_ld {c('scratch_load')} scratch_st {c('scratch_store')} "
            f"accvgpr {c('v_accvgpr')} waitcnt {c('s_waitcnt')} nop {c('s_nop')}")
for i, t in marks:
    if i - prev > 40:
        print(f"   [{prev}-{i}] {mix(prev, i)}")
    print(f"{i:6d} {t}")
    prev = i
print(f"   [{prev}-{len(body)}] {mix(prev, len(body))}")
