"""CPU: the tap-GEMM geometry (ct-vae_amd/csrc/geom.hpp: tap tables, output-parity classes, per-tap weight
transposition, packed [tap][Ci][Co] weight layout) emulated on the host with naive loops and compared with
torch's own conv / conv_transpose forward, input-gradient and weight-gradient for every layer shape on the path."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_emul", "geom_emul.cpp")
SO = os.path.join(HERE, "host_emul", "_geom_emul.so")


@pytest.fixture(scope="module")
def emul():
    hdr = os.path.join(os.path.dirname(HERE), "ct-vae_amd", "csrc", "geom.hpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(SRC), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-std=c++17", "-o", SO, SRC])
    lib = ctypes.CDLL(SO)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.emul_tapgemm.argtypes = [ctypes.c_int, fp, fp, fp] + [ctypes.c_int] * 9
    lib.emul_wgrad.argtypes = [ctypes.c_int, fp, fp, fp] + [ctypes.c_int] * 9
    return lib


def ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def pack_weight(w, transposed):
    """PyTorch layout -> packed [kh*kw][Ci][Co]."""
    if transposed:   # ConvTranspose2d [Ci,Co,kh,kw]
        return w.permute(2, 3, 0, 1).contiguous()
    return w.permute(2, 3, 1, 0).contiguous()   # Conv2d [Co,Ci,kh,kw]


CASES = [
    # (transposed, Ci, Co, H, k, s, p, op)
    (False, 3, 8, 16, 3, 2, 1, 0),      # vanilla encoder.0 shape family
    (False, 8, 16, 8, 3, 2, 1, 0),
    (False, 8, 3, 8, 3, 1, 1, 0),       # final conv 32->3 family
    (False, 3, 8, 16, 4, 2, 1, 0),      # mcq encoder.0
    (False, 8, 8, 8, 4, 2, 1, 0),
    (False, 8, 8, 4, 3, 1, 1, 0),       # residual 3x3
    (False, 8, 4, 4, 1, 1, 0, 0),       # 1x1
    (False, 32, 16, 1, 1, 1, 0, 0),     # linear
    (True, 8, 4, 4, 3, 2, 1, 1),        # vanilla decoder
    (True, 8, 4, 4, 4, 2, 1, 0),        # mcq decoder
    (True, 8, 3, 8, 4, 2, 1, 0),        # mcq final
]


@pytest.mark.parametrize("case", CASES)
def test_geometry_matches_torch(emul, case):
    tr, Ci, Co, H, k, s, p, op = case
    B = 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Ci, H, H, generator=g).requires_grad_(True)
    w = torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), generator=g).requires_grad_(True)
    y = F.conv_transpose2d(x, w, None, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, None, stride=s, padding=p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xn = x.detach().permute(0, 2, 3, 1).contiguous().numpy()
    wn = pack_weight(w.detach(), tr).numpy()
    gyn = gy.permute(0, 2, 3, 1).contiguous().numpy()
    Ho = y.shape[2]
    args = (B, H, H, Ci, Co, k, s, p, op)
    # forward
    out = np.zeros((B, Ho, Ho, Co), np.float32)
    assert emul.emul_tapgemm(1 if tr else 0, ptr(xn), ptr(wn), ptr(out), *args) == 0
    np.testing.assert_allclose(out, y.detach().permute(0, 2, 3, 1).numpy(), atol=2e-4, rtol=1e-4)
    # dgrad
    dx = np.zeros((B, H, H, Ci), np.float32)
    assert emul.emul_tapgemm(3 if tr else 2, ptr(gyn), ptr(wn), ptr(dx), *args) == 0
    np.testing.assert_allclose(dx, x.grad.permute(0, 2, 3, 1).numpy(), atol=2e-4, rtol=1e-4)
    # wgrad
    dw = np.zeros_like(wn)
    assert emul.emul_wgrad(1 if tr else 0, ptr(xn), ptr(gyn), ptr(dw), *args) == 0
    np.testing.assert_allclose(dw, pack_weight(w.grad, tr).numpy(), atol=5e-4, rtol=1e-4)


FLAT_CASES = [
    # (C, k, stride, out): nn.Linear(C*k*k, out) over torch.flatten(h [B,C,k,k], 1), run as a k x k convolution of the NHWC tensor
    (8, 2, 2, 12),      # vanilla_vae.py:36-37 family (fc_mu | fc_var over [B,512,2,2])
    (8, 2, 1, 12),
    (4, 4, 1, 6),       # betatc_vae.py family (Linear over a 4x4 map)
]


@pytest.mark.parametrize("case", FLAT_CASES)
def test_linear_over_flatten_as_convolution(emul, case):
    """CTVAE_W_CI_TAP (include/ctvae_hip.h): the Linear layer's [in][out] block read as [Ci][tap][Co]."""
    C, k, s, out = case
    B, FLAG = 3, 0x100
    g = torch.Generator().manual_seed(5)
    h = torch.randn(B, C, k, k, generator=g).requires_grad_(True)
    w = torch.randn(out, C * k * k, generator=g).requires_grad_(True)
    y = F.linear(torch.flatten(h, start_dim=1), w)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    hn = h.detach().permute(0, 2, 3, 1).contiguous().numpy()
    wn = w.detach().t().contiguous().numpy()                      # packing.PackedLinear memory: [in][out]
    gyn = gy.contiguous().numpy()
    args = (B, k, k, C, out, k, s, 0, 0)
    o = np.zeros((B, 1, 1, out), np.float32)
    assert emul.emul_tapgemm(0 | FLAG, ptr(hn), ptr(wn), ptr(o), *args) == 0
    np.testing.assert_allclose(o.reshape(B, out), y.detach().numpy(), atol=2e-4, rtol=1e-4)
    dx = np.zeros((B, k, k, C), np.float32)
    assert emul.emul_tapgemm(2 | FLAG, ptr(gyn), ptr(wn), ptr(dx), *args) == 0
    np.testing.assert_allclose(dx, h.grad.permute(0, 2, 3, 1).numpy(), atol=2e-4, rtol=1e-4)
    dw = np.zeros_like(wn)
    assert emul.emul_wgrad(0 | FLAG, ptr(hn), ptr(gyn), ptr(dw), *args) == 0
    np.testing.assert_allclose(dw, w.grad.t().numpy(), atol=5e-4, rtol=1e-4)
