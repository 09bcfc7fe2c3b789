#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 PMC passes over bench.py (HBM bytes per launch, per kernel).

On the GPU box (separate passes, counters only with --kernel-trace, as MI355X_MICROARCH.md prescribes):
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
then here:  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write
FETCH_SIZE / WRITE_SIZE count KiB; gfx950 reports half the bytes of wide coalesced reads, so FETCH_SIZE is doubled
(the guide's gfx950 correction)."""
import collections
import csv
import glob
import json
import os
import re
import sys

NT_NAMES = {0: "bias_bf16", 1: "relu_bf16", 2: "f32", 3: "resid_f32", 4: "gelu_bf16", 5: "dgelu_bf16", 6: "drelu_bf16",
            7: "patch_f32", 8: "relu_f32", 9: "drelu_f32", 10: "gelu_grad_bf16", 11: "mul_bf16"}


def timer_name(kernel):
    m = re.search(r"gemm_nt(?:_a3|_w8)?_kernel<(\d+)>|gemm_nt(?:_a3|_w8)?_kernelILi(\d+)E", kernel)
    if m:
        return "gemm_nt<%s>" % NT_NAMES[int(m.group(1) or m.group(2))]
    for key, name in (("gemm_tn_wide_kernel", "gemm_tn_grouped"), ("gemm_tn_grouped_kernel", "gemm_tn_grouped"),
                      ("attn_fwd_kernel", "vit_attn_fwd"), ("attn_bwd_dq_kernel", "vit_attn_bwd_dq"),
                      ("attn_bwd_dkv_kernel", "vit_attn_bwd_dkv"), ("ln_fwd_kernel", "ln_fwd"),
                      ("ln_bwd_kernel", "ln_bwd")):
        if key in kernel:
            return name
    return None


def collect(folder, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(folder, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                n = timer_name(r["Kernel_Name"])
                if n:
                    agg[n].append(float(r["Counter_Value"]))
    return agg


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    out = {}
    for n in sorted(set(fetch) & set(write)):
        f = sum(fetch[n]) / len(fetch[n])
        w = sum(write[n]) / len(write[n])
        out[n] = {"hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_kb_raw": int(f), "write_kb": int(w),
                  "launches_sampled": len(fetch[n]),
                  "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled (gfx950 "
                          "reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md HBM section)"}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    for n, v in out.items():
        print(f"{n:28s} {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB/launch  ({v['launches_sampled']} launches)")


if __name__ == "__main__":
    main()
