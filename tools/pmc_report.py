#!/usr/bin/env python3
"""profiles/pmc_mfma.json and profiles/pmc_traffic.json from rocprofv3 PMC passes over bench.py.

On the GPU box (counters in their own passes, program directly after `--`, as MI355X_MICROARCH.md prescribes):
    cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph"
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
              SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- $B
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- $B
then here:  python tools/pmc_report.py gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write

Units (guide, 'Per-instruction cycle constants'): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed
over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES count cycles summed over CUs; GRBM_GUI_ACTIVE is summed over
the 8 XCDs.  FETCH_SIZE / WRITE_SIZE count KiB; gfx950 reports half the bytes of wide coalesced reads, so FETCH_SIZE is
doubled (the guide's gfx950 correction)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NT_NAMES = {0: "bias_bf16", 1: "relu_bf16", 2: "f32", 3: "resid_f32", 4: "gelu_bf16", 5: "dgelu_bf16", 6: "drelu_bf16",
            7: "patch_f32", 8: "relu_f32", 9: "drelu_f32", 10: "gelu_grad_bf16", 11: "mul_bf16", 12: "raw_slabs_f32",
            13: "gelu_gradq_bf16", 14: "mulq_bf16"}
ROW_NAMES = {0: "gemm_nt<bias_bf16>(row)", 1: "gemm_nt<resid_f32>(row)", 2: "gemm_ln_fwd", 3: "gemm_ln_bwd"}


# the row kernel runs two shapes per family and the name does not carry K: dispatches of a family follow the training
# step's launch order (vit.py), so K is read off the position inside the step
# (round 4: the last block runs on the CLS rows only — no row-kernel launches forward, only dX qkv + norm1' backward)
ROW_K_PATTERN = {"gemm_ln_fwd": [384, 1536] * 11,                  # proj + norm2, fc2 + next norm1 of blocks 0..10
                 "gemm_ln_bwd": [1152] + [1536, 1152] * 11}        # block 11: dX qkv + norm1'; then dX fc1 + norm2', dX qkv + norm1'


def timer_name(kernel):
    m = re.search(r"gemm_nt_row_kernel<(\d+)[,>]|gemm_nt_row_kernel<(\d+)>|gemm_nt_row_kernelILi(\d+)E", kernel)
    if m:
        return ROW_NAMES[int(m.group(1) or m.group(2) or m.group(3))]
    m = re.search(r"gemm_nt(?:_w8p)?_kernel<(\d+)>|gemm_nt(?:_w8p)?_kernelILi(\d+)E", kernel)
    if m:
        return "gemm_nt<%s>" % NT_NAMES[int(m.group(1) or m.group(2))]
    for key, name in (("gemm_tn_xl_kernel", "gemm_tn_grouped[xl]"), ("xl_finish_kernel", "gemm_tn_grouped[finish]"),
                      ("tn_slab_finish_kernel", "gemm_tn_grouped[finish]"),
                      ("gemm_tn_pp_kernel", "gemm_tn_grouped"), ("gemm_tn_wide_kernel", "gemm_tn_grouped"), ("gemm_tn_grouped_kernel", "gemm_tn_grouped[compact]"),
                      ("attn_cls_fwd_kernel", "vit_attn_cls_fwd"), ("attn_cls_bwd_kernel", "vit_attn_cls_bwd"),
                      ("attn_fwd_kernel", "vit_attn_fwd"), ("attn_bwd_dq_kernel", "vit_attn_bwd_dq"),
                      ("attn_bwd_dkv_kernel", "vit_attn_bwd_dkv"), ("attn_bwd_kernel", "vit_attn_bwd"),
                      ("tln_fwd_kernel", "temporal_ln_fwd"), ("tln_bwd_kernel", "temporal_ln_bwd"), ("tgemm_kernel", "tgemm"),
                      ("tattn_fwd_kernel", "temporal_attn_fwd"), ("tattn_bwd_kernel", "temporal_attn_bwd"),
                      ("gemm_tn_grouped_f32_kernel", "gemm_tn_grouped_f32"),
                      ("ln_fwd_kernel", "ln_fwd"), ("ln_bwd_kernel", "ln_bwd"), ("sgd_kernel", "sgd")):
        if key in kernel:
            return name
    return None


def collect(folder):
    """{kernel family: {counter: [values per dispatch]}}; the LayerNorm-fused row GEMMs are split by K ("gemm_ln_fwd[K384]")."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(folder, "**", "*_counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        seen = collections.defaultdict(dict)                # family -> {dispatch id: ordinal}
        # round 6: the four-GEMM launch of a block runs on the 192 x 384 kernel (+ its finish); the 128 x 384 kernel is left with
        # the k / v-only launch of the CLS-only block
        xl = any("gemm_tn_xl_kernel" in r["Kernel_Name"] for r in rows)
        for r in rows:
            n = timer_name(r["Kernel_Name"])
            if not n:
                continue
            if xl and n == "gemm_tn_grouped":
                n = "gemm_tn_grouped[qkv only]"
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
                continue
            if n == "gemm_tn_grouped[xl]":
                # launches of different sizes share the kernel (R6.8: blocks 10..1 + the k / v gradient in one launch of 244
                # workgroups, block 0 in one of 24 tiles x 10 splits): families by workgroup count, as bench.py keys them
                try:
                    n = "gemm_tn_grouped[%d wg]" % (int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
                except (KeyError, ValueError, ZeroDivisionError):
                    n = "gemm_tn_grouped"
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
                continue
            if n in ROW_K_PATTERN:
                ordinal = seen[n].setdefault(r["Dispatch_Id"], len(seen[n]))
                pat = ROW_K_PATTERN[n]
                n = "%s[K%d]" % (n, pat[ordinal % len(pat)])
            elif n == "gemm_tn_grouped":
                # per step: the last block's qkv-only launch first, then the four-GEMM launch of blocks 10..0 (round 4)
                ordinal = seen[n].setdefault(r["Dispatch_Id"], len(seen[n]))
                if ordinal % 12 == 0:
                    n = "gemm_tn_grouped[qkv only]"
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else 0.0


def provenance():
    """Which tree the counters describe: the files outlive kernel changes, so they carry the commit they were collected on
    (the gpurun box has no .git: tools/gpu_validate.sh writes gpurun_out/<tag>/HEAD before the passes; else `git rev-parse`)."""
    import datetime
    import subprocess
    head = None
    for d in sys.argv[1:4]:
        f = os.path.join(os.path.dirname(os.path.abspath(d)), "HEAD")
        if os.path.exists(f):
            head = open(f).read().strip()
            break
    dirty = None
    if head is None:
        try:
            head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
            dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "--", "sais_amd"], text=True).strip())
        except Exception:
            head = "unknown"
    sys.path.insert(0, ROOT)
    import bench
    return {"head": head, "kernel_sources_modified_since_head": dirty, "kernel_source_hash": bench.kernel_source_hash(ROOT),
            "collected": datetime.date.today().isoformat(), "passes": [os.path.relpath(os.path.abspath(d), ROOT) for d in sys.argv[1:4]]}


def main():
    sq, fetch, write = collect(sys.argv[1]), collect(sys.argv[2]), collect(sys.argv[3])
    mf = {}
    for n, c in sorted(sq.items()):
        g = {k: mean(v) for k, v in c.items()}
        wave = g.get("SQ_WAVE_CYCLES", 0.0)
        busy = g.get("SQ_BUSY_CU_CYCLES", 0.0)
        mf[n] = {
            "launches_sampled": len(next(iter(c.values()))),
            "counters_per_launch": {k: int(v) for k, v in g.items()},
            # one MFMA pipe per SIMD, 4 SIMDs per CU: busy cycles per CU-cycle-with-work
            "mfma_busy_frac_of_cu_busy": round(g.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / busy / 4, 4) if busy else None,
            "wait_any_frac_of_wave_cycles": round(g.get("SQ_WAIT_ANY", 0.0) / wave, 4) if wave else None,
            "wait_inst_any_frac_of_wave_cycles": round(g.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4) if wave else None,
            "active_inst_frac_of_wave_cycles": round(g.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4) if wave else None,
            "lds_bank_conflict_ratio": round(g.get("SQ_LDS_BANK_CONFLICT", 0.0) / g["SQ_LDS_IDX_ACTIVE"], 4)
            if g.get("SQ_LDS_IDX_ACTIVE") else None,
        }
    prov = provenance()
    json.dump({"collected_with": "rocprofv3 --pmc (one SQ pass) -- python3 bench.py --steps 2 --warmup 1 --no-graph",
               "_provenance": prov, "kernels": mf}, open(os.path.join(ROOT, "profiles", "pmc_mfma.json"), "w"), indent=1)
    tr = {}
    for n in sorted(set(fetch) & set(write)):
        f, w = mean(fetch[n].get("FETCH_SIZE", [])), mean(write[n].get("WRITE_SIZE", []))
        tr[n] = {"hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_kb_raw": int(f), "write_kb": int(w),
                 "launches_sampled": len(fetch[n].get("FETCH_SIZE", [])),
                 "note": "separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled (gfx950 correction)"}
    # the launches that cut M into splits (240 workgroups = 24 tiles x 10, 48 x 5, ...) are followed by a finish launch, and bench.py
    # times the PAIR under one tag: report the pair's bytes there too
    pair = next((k for k in ("gemm_tn_grouped[240 wg]", "gemm_tn_grouped") if k in tr), None)
    if "gemm_tn_grouped[finish]" in tr and pair:
        tr[pair]["hbm_bytes_per_launch_kernel_only"] = tr[pair]["hbm_bytes_per_launch"]
        tr[pair]["hbm_bytes_per_launch"] += tr["gemm_tn_grouped[finish]"]["hbm_bytes_per_launch"]
        tr[pair]["note"] += "; = 192 x 384 kernel + its finish launch (the pair bench.py times under this tag)"
    # whole-step HBM bytes by counters: launches per step x bytes per launch over every family that was matched (steps in the
    # profiled run = launches of the CLS-only block's compact dW, one per step)
    nsteps = max(1, len(fetch.get("gemm_tn_grouped[compact]", {}).get("FETCH_SIZE", [])))
    per_step = {n: round(v["launches_sampled"] / nsteps, 2) for n, v in tr.items()}
    step_bytes = sum(tr[n].get("hbm_bytes_per_launch_kernel_only", tr[n]["hbm_bytes_per_launch"]) * per_step[n] for n in tr)
    tr["_step"] = {"step_hbm_bytes": int(step_bytes), "steps_profiled": nsteps, "launches_per_step": per_step,
                   "note": "sum over the matched kernel families (all GEMM / attention / LayerNorm / SGD kernels; the "
                           "temporal-encoder kernels and elementwise torch kernels are not matched)"}
    tr["_provenance"] = prov
    json.dump(tr, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print("step_hbm_bytes %.2f GB over %d steps" % (step_bytes / 1e9, nsteps))
    for n, v in mf.items():
        t = tr.get(n, {}).get("hbm_bytes_per_launch", 0) / 1e6
        print(f"{n:28s} mfma {v['mfma_busy_frac_of_cu_busy']}  wait_any {v['wait_any_frac_of_wave_cycles']}  "
              f"wait_inst {v['wait_inst_any_frac_of_wave_cycles']}  lds_conf {v['lds_bank_conflict_ratio']}  hbm {t:8.1f} MB")


if __name__ == "__main__":
    main()
