import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import bench, synth
from sais_amd.parallel import GradSync
dev = torch.device("cuda", 0)
B, T, C = 8, 32, 2
vit, model, protos, opt = bench.build(dev, B, T, C, lr=0.1)
frames = synth.clips(seed=0, B=B, T=T).view(B * T, 3, 224, 224).to(dev)
pad = synth.padding_mask([T] * B).to(dev)
labels = synth.labels(seed=0, B=B, nclasses=C)
step = bench.make_step(vit, model, protos, opt, GradSync(1), frames, pad, labels, B, T, 1)
vit(frames[:2]); model._engine(dev)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"eager: CPU issue {1e3 * (t1 - t0) / 10:.2f} ms/step, wall {1e3 * (t2 - t0) / 10:.2f} ms/step")
