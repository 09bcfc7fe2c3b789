#!/usr/bin/env python3
"""profiles/<round>_pmc_tcp.json (name from SAIS_TCP_OUT, default r06_pmc_tcp.json): vector-L1 (TCP) and L2 request counters per kernel family of the training step, from rocprofv3 --pmc
passes over `bench.py --steps 2 --warmup 1 --no-graph` (tools/scratch/gpu_r5m.sh).  usage: pmc_tcp_report.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_report import ROOT, timer_name  # noqa: E402

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = timer_name(r["Kernel_Name"])
            if n == "gemm_tn_grouped[xl]":       # launches of different sizes share the kernel: families by workgroup count (pmc_report.py)
                try:
                    n = "gemm_tn_grouped[%d wg]" % (int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
                except (KeyError, ValueError, ZeroDivisionError):
                    pass
            if n:
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for n, c in sorted(agg.items()):
    g = {k: sum(v) / len(v) for k, v in c.items()}
    gate = g.get("TCP_GATE_EN1_sum", 0.0)
    rd, wr = g.get("TCP_TCC_READ_REQ_sum", 0.0), g.get("TCP_TCC_WRITE_REQ_sum", 0.0)
    out[n] = {"launches_sampled": max(len(v) for v in c.values()),
              "tcp_pending_stall_frac_of_gated": round(g.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / gate, 4) if gate else None,
              "tcp_ta_data_stall_frac_of_gated": round(g.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0.0) / gate, 4) if gate else None,
              "tcp_gated_cycles_per_tcp": int(gate / 256),
              "l2_read_requests": int(rd), "l2_read_latency_cycles": round(g.get("TCP_TCC_READ_REQ_LATENCY_sum", 0.0) / rd, 1) if rd else None,
              "l2_write_requests": int(wr), "l2_write_latency_cycles": round(g.get("TCP_TCC_WRITE_REQ_LATENCY_sum", 0.0) / wr, 1) if wr else None,
              "tcc_ea_wrreq_stall": int(g.get("TCC_EA0_WRREQ_STALL_sum", 0.0)), "tcc_busy_sum": int(g.get("TCC_BUSY_sum", 0.0))}
json.dump({"collected_with": "rocprofv3 --pmc (three passes: TCP stalls, TCP<->TCC requests and latencies, TCC) -- python3 bench.py --steps 2 "
                             "--warmup 1 --no-graph --no-cpu-baseline --sustain-seconds 0 --parity-clips 0 --no-variants",
           "head": open(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "HEAD")).read().strip()
           if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "HEAD")) else "unknown",
           "kernels": out}, open(os.path.join(ROOT, "profiles", os.environ.get("SAIS_TCP_OUT", "r06_pmc_tcp.json")), "w"), indent=1)
for n, v in out.items():
    print(f"{n:28s} pend {v['tcp_pending_stall_frac_of_gated']}  ta_stall {v['tcp_ta_data_stall_frac_of_gated']}  rd {v['l2_read_requests']:>9d} @ {v['l2_read_latency_cycles']}  "
          f"wr {v['l2_write_requests']:>9d} @ {v['l2_write_latency_cycles']}  ea_wr_stall {v['tcc_ea_wrreq_stall']}")
