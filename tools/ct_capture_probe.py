"""Does one CT-MCQ-VAE mode survive hipGraph capture?  python tools/ct_capture_probe.py <mode> [B]  (run each mode in its own process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import faulthandler
faulthandler.enable()
import torch
import yaml
from ctvae_amd import filler, kernels as K
from ctvae_amd.models import vae_models
from ctvae_amd.optim import FlatAdam

mode, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
cfg = yaml.safe_load(open("configs/ct_mcq_vae.yaml"))["model_params"]
m = vae_models["CTMCQVAE"](**cfg).to(dev).train()
opt = FlatAdam(m, lr=5e-4, params_slice=m.flat_range("ct_layer"))
x, y, a = filler.synthetic_pairs(1, B, 12)
x, y, a = x.to(dev), y.to(dev), a.to(dev)
kw = {"mode": [mode] * B}
if mode != "base":
    kw.update(input_y=y, action=a)


def body():
    m.zero_grad()
    out = m(x, **kw)
    l = m.loss_function(*out, M_N=0.00025)
    K.backward(l["loss"])
    opt.step()
    return l["loss"]


for _ in range(3):
    body()
torch.cuda.synchronize()
print("eager ok", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = body()
print("captured", flush=True)
g.replay()
torch.cuda.synchronize()
print("replayed", float(loss), flush=True)
