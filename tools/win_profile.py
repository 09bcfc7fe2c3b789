#!/usr/bin/env python3
"""Where the sliding-window half of a 512-frame video's inference goes: host collate, the encoder pass, the copies to the host."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import synth
from sais_amd.inference import collate_windows_tta, gesture_windows, run_windows
from sais_amd.temporal import fullModel
dev = "cuda"
model = fullModel('reps', 2, 'in_vs_out', 384, 'ViT', modalities='RGB-Flow').to(dev).eval()
reps = synth.reps(seed=1, B=1, T=512)[0, 0].to(dev); freps = synth.reps(seed=2, B=1, T=34)[0, 0].to(dev)
wins = gesture_windows(512)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
c = collate_windows_tta(reps, freps, wins)
with torch.no_grad():
    print("collate            %.3f ms" % t(lambda: collate_windows_tta(reps, freps, wins)))
    print("model (merged)     %.3f ms" % t(lambda: model(c["x"], c["f"], c["xlens"], c["flens"], 'Prototypes', c["xpad"], c["fpad"], None)))
    print("run_windows        %.3f ms" % t(lambda: run_windows(model, reps, freps, videoname="v", batch_size=2)))
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        for _ in range(5):
            model(c["x"], c["f"], c["xlens"], c["flens"], 'Prototypes', c["xpad"], c["fpad"], None)
        torch.cuda.synchronize()
    ev = [e for e in prof.key_averages() if e.device_time_total > 0]
    tot = sum(e.device_time_total for e in ev) / 5
    print("device time per model call: %.1f us in %d kernels" % (tot, sum(e.count for e in ev) / 5))
    for e in sorted(ev, key=lambda e: -e.device_time_total)[:14]:
        print("  %-60s %6.1f us x %d" % (e.key[:60], e.device_time_total / e.count, e.count / 5))
