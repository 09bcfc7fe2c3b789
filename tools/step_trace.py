"""One training step of bench.py as the sequence of its kernel dispatches (rocprofv3 --kernel-trace CSV): which kernels sit next
to the blit copies / fills, how many of each per step.  Usage (on the GPU box, after
  rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustain-seconds 0 --parity-clips 0 --no-variants):
  python tools/step_trace.py <dir> [pattern]"""
import csv
import glob
import sys
from collections import Counter

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2] if len(sys.argv) > 2 else "copyBuffer"
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
marks = [i for i, r in enumerate(rows) if "embed_bwd_kernel" in r["Kernel_Name"]]          # once per step (patch-embed backward)
print("dispatches", len(rows), "steps seen", len(marks), "total", pat, sum(pat in r["Kernel_Name"] for r in rows))
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    n = sum(pat in r["Kernel_Name"] for r in seg)
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg if pat in r["Kernel_Name"])
    span = int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    print(f"step of {len(seg)} dispatches, {span / 1e6:.3f} ms: {n} x {pat} = {dur / 1e3:.1f} us; queues {Counter(r.get('Queue_Id') for r in seg)}")
if len(marks) >= 2:
    seg = rows[marks[-2]:marks[-1]]
    prev = Counter()
    for i, r in enumerate(seg):
        if pat in r["Kernel_Name"]:
            prev[(short(seg[i - 1]["Kernel_Name"]) if i else "-", short(seg[i + 1]["Kernel_Name"]) if i + 1 < len(seg) else "-")] += 1
    for k, v in prev.most_common(25):
        print(v, "after", k[0], "| before", k[1])
    # where the matched dispatches sit on the time line of the step
    t0 = int(seg[0]["Start_Timestamp"])
    pos = [(int(r["Start_Timestamp"]) - t0) / 1e6 for r in seg if pat in r["Kernel_Name"]]
    print("first / last at ms", pos[:3], pos[-3:])
