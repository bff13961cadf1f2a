"""Which torch (non-ctvae) device kernels does one training step launch, and from which op?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from ctvae_amd import filler, native
from ctvae_amd.optim import FlatAdam
native.load()
dev = torch.device("cuda")
model = bench.build_model("VanillaVAE", dev, 1265)
opt = FlatAdam(model, lr=0.005)
x = filler.synthetic_batch(1, 256)[0].to(dev)
def step():
    model.zero_grad()
    out = model(x)
    l = model.loss_function(*out, M_N=0.00025)
    l["loss"].backward()
    opt.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
# device kernels with their launching CPU op
rows = []
for e in evs:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        continue
for e in prof.key_averages():
    if e.device_time_total > 0 and not e.key.startswith("ctvae") and "ctvae::" not in e.key:
        rows.append((e.key, e.count, e.device_time_total))
for k, c, t in sorted(rows, key=lambda r: -r[2])[:40]:
    print(f"{c:4d}  {t:9.1f} us  {k[:100]}")
