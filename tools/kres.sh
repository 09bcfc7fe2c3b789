#!/bin/bash
# kernel resource usage (VGPRs / AGPRs / scratch / LDS / occupancy) of one HIP source: tools/kres.sh sais_amd/csrc/gemm_row.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /tmp/kres.o 2>&1 | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line: print(line.rstrip())
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur = {"fn": t.split(":",1)[1].strip()[:110]}
    elif ":" in t:
        k, v = t.split(":", 1)
        k = k.strip()
        if k in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"):
            cur[k.split(" ")[0]] = v.strip()
if cur: print(cur)
'
