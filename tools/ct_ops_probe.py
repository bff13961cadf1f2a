"""Where does a CTMCQVAE action-mode step spend its GPU time (torch ops of the causal-transition layer vs ctvae kernels)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from ctvae_amd import filler, native
native.load()
dev = torch.device("cuda")
B = int(os.environ.get("PB", 128))
model = bench.build_model("CTMCQVAE", dev, 1250)
x, y, a = filler.synthetic_pairs(1250, B, 12)
x, y, a = x.to(dev), y.to(dev), a.to(dev)
kw = {"mode": ["action"] * B, "input_y": y, "action": a}
def step():
    model.zero_grad()
    out = model(x, **kw)
    l = model.loss_function(*out, M_N=0.00025)
    l["loss"].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=70))
