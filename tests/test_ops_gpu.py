"""GPU: every HIP entry point, called through the C ABI, against torch fp32 CPU arithmetic of the same op
(tolerance 1e-4 absolute on O(1) data, the north_star bound)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ctvae_amd import kernels
    from ctvae_amd import native
    native.load()
    return kernels


def pack(w, transposed):
    return (w.permute(2, 3, 0, 1) if transposed else w.permute(2, 3, 1, 0)).contiguous()


def as_param(packed, transposed):
    """packed [kh,kw,Ci,Co] device tensor -> parameter with PyTorch logical shape over the same memory"""
    v = packed.permute(2, 3, 0, 1) if transposed else packed.permute(3, 2, 0, 1)
    return torch.nn.Parameter(v)


CASES = [
    # transposed, Ci, Co, H, k, s, p, op, B
    (False, 3, 32, 64, 3, 2, 1, 0, 3),     # vanilla encoder.0
    (False, 32, 64, 32, 3, 2, 1, 0, 3),
    (False, 256, 512, 4, 3, 2, 1, 0, 5),   # vanilla encoder.4 (small M)
    (False, 32, 3, 64, 3, 1, 1, 0, 2),     # final conv 32->3
    (False, 3, 64, 64, 4, 2, 1, 0, 2),     # mcq encoder.0
    (False, 64, 128, 32, 4, 2, 1, 0, 2),
    (False, 256, 256, 8, 3, 1, 1, 0, 3),   # residual 3x3
    (False, 256, 128, 8, 1, 1, 0, 0, 3),   # 1x1
    (False, 2048, 256, 1, 1, 1, 0, 0, 7),  # fc heads as one GEMM
    (False, 128, 2048, 1, 1, 1, 0, 0, 7),  # decoder_input
    (True, 512, 256, 2, 3, 2, 1, 1, 3),    # vanilla decoder.0
    (True, 32, 32, 32, 3, 2, 1, 1, 2),     # final_layer.0
    (True, 256, 128, 8, 4, 2, 1, 0, 3),    # mcq decoder.8
    (True, 64, 3, 32, 4, 2, 1, 0, 2),      # mcq final
]


@pytest.mark.parametrize("case", CASES)
def test_conv_family(K, case):
    tr, Ci, Co, H, k, s, p, op, B = case
    g = torch.Generator().manual_seed(100 + Ci + Co + k)
    x = torch.randn(B, Ci, H, H, generator=g).requires_grad_(True)
    fan = Ci * k * k / (s * s if tr else 1)
    w = (torch.randn((Ci, Co, k, k) if tr else (Co, Ci, k, k), generator=g) / fan ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=s, padding=p, output_padding=op) if tr else F.conv2d(x, w, b, stride=s, padding=p)
    ya = F.leaky_relu(y, 0.01)
    gy = torch.randn(ya.shape, generator=g)
    ya.backward(gy)

    dev = torch.device("cuda")
    spec = K.ConvSpec(K.CONVT if tr else K.CONV, Ci, Co, k, s, p, op, K.ACT_LRELU)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wp = as_param(pack(w.detach(), tr).to(dev), tr)
    bp = torch.nn.Parameter(b.detach().to(dev))
    out = K.ConvAct.apply(xd, wp, bp, None, spec)
    out.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.detach().cpu().permute(0, 3, 1, 2).numpy(), ya.detach().numpy(), atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(xd.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy(), atol=TOL, rtol=1e-4)
    scale = max(1.0, float(w.grad.abs().max()))
    np.testing.assert_allclose(wp.grad.cpu().numpy(), w.grad.numpy(), atol=TOL * scale, rtol=1e-4)
    np.testing.assert_allclose(bp.grad.cpu().numpy(), b.grad.numpy(), atol=TOL * max(1.0, float(b.grad.abs().max())), rtol=1e-4)
    # accumulate semantics + determinism: a second backward doubles the gradient bit-reproducibly
    g1 = wp.grad.clone()
    out2 = K.ConvAct.apply(xd, wp, bp, None, spec)
    out2.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(wp.grad.cpu().numpy(), 2 * g1.cpu().numpy(), atol=TOL * scale, rtol=1e-4)


@pytest.mark.parametrize("C,R_shape", [(32, (3, 32, 32)), (512, (5, 2, 2)), (64, (2, 16, 16))])
def test_conv_bn_lrelu(K, C, R_shape):
    """ConvBNAct against conv2d + F.batch_norm(training) + leaky_relu incl. running stats and all gradients."""
    B, H, W = R_shape
    Ci = 32
    g = torch.Generator().manual_seed(C)
    x = torch.randn(B, Ci, H * 2, W * 2, generator=g).requires_grad_(True)
    w = (torch.randn(C, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(C, generator=g).requires_grad_(True)
    gamma = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(C, generator=g) - 0.5).requires_grad_(True)
    rm, rv = torch.rand(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = F.conv2d(x, w, b, stride=2, padding=1)
    a = F.leaky_relu(F.batch_norm(y, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5), 0.01)
    ga = torch.randn(a.shape, generator=g)
    a.backward(ga)

    dev = torch.device("cuda")
    spec = K.ConvSpec(K.CONV, Ci, C, 3, 2, 1, 0, K.ACT_NONE)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wp = as_param(pack(w.detach(), False).to(dev), False)
    bp, gp, btp = (torch.nn.Parameter(t.detach().to(dev)) for t in (b, gamma, beta))
    rmd, rvd = rm.to(dev), rv.to(dev)
    nbt = torch.zeros((), dtype=torch.long, device=dev)
    out = K.ConvBNAct.apply(xd, wp, bp, gp, btp, rmd, rvd, True, spec, K.ACT_LRELU, nbt)
    out.backward(ga.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.detach().cpu().permute(0, 3, 1, 2).numpy(), a.detach().numpy(), atol=TOL, rtol=1e-4)
    assert int(nbt.item()) == 1
    np.testing.assert_allclose(rmd.cpu().numpy(), rm_ref.numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv_ref.numpy(), atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(xd.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy(), atol=TOL, rtol=1e-3)
    np.testing.assert_allclose(gp.grad.cpu().numpy(), gamma.grad.numpy(), atol=TOL * 10, rtol=1e-3)
    np.testing.assert_allclose(btp.grad.cpu().numpy(), beta.grad.numpy(), atol=TOL * 10, rtol=1e-3)
    np.testing.assert_allclose(wp.grad.cpu().numpy(), w.grad.numpy(), atol=TOL * 10, rtol=1e-3)
    np.testing.assert_allclose(bp.grad.cpu().numpy(), b.grad.numpy(), atol=TOL * 10, rtol=0)   # analytically zero


def test_reparam_and_loss(K):
    g = torch.Generator().manual_seed(5)
    B, L = 6, 128
    mu = torch.randn(B, L, generator=g).requires_grad_(True)
    lv = (0.5 * torch.randn(B, L, generator=g)).requires_grad_(True)
    eps = torch.randn(B, L, generator=g)
    r = torch.rand(B, 64, 64, 3, generator=g).requires_grad_(True)
    x = torch.rand(B, 64, 64, 3, generator=g)
    z = eps * torch.exp(0.5 * lv) + mu
    mse = F.mse_loss(r, x)
    kld = torch.mean(-0.5 * torch.sum(1 + lv - mu ** 2 - lv.exp(), dim=1), dim=0)
    loss = mse + 0.00025 * kld + (z * z).sum() * 1e-3
    loss.backward()
    dev = torch.device("cuda")
    md, ld, rd = (t.detach().to(dev).requires_grad_(True) for t in (mu, lv, r))
    zd = K.Reparameterize.apply(md, ld, eps.to(dev))
    out = K.VAELoss.apply(rd, x.to(dev), md, ld, None, 0.00025)
    (out[0] + (zd * zd).sum() * 1e-3).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(zd.detach().cpu().numpy(), z.detach().numpy(), atol=1e-5, rtol=1e-5)
    assert abs(out[1].item() - mse.item()) < 1e-6 and abs(out[2].item() - kld.item()) < 1e-4 * abs(kld.item())
    assert abs(out[3].item() + kld.item()) < 1e-4 * abs(kld.item())
    np.testing.assert_allclose(rd.grad.cpu().numpy(), r.grad.numpy(), atol=1e-9, rtol=1e-4)
    np.testing.assert_allclose(md.grad.cpu().numpy(), mu.grad.numpy(), atol=1e-7, rtol=1e-4)
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lv.grad.numpy(), atol=1e-7, rtol=1e-4)


def test_permute_roundtrip(K):
    x = torch.randn(3, 5, 6, 7)
    xd = x.cuda()
    n = K.to_nhwc(xd)
    assert torch.equal(n.cpu(), x.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(K._ToNCHW.apply(n).cpu(), x)
    cl = xd.contiguous(memory_format=torch.channels_last)
    assert K.to_nhwc(cl).data_ptr() == cl.data_ptr()          # channels_last input is consumed zero-copy


def test_adam_matches_torch(K):
    g = torch.Generator().manual_seed(9)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=0.005, weight_decay=0.01)
    pd = p0.cuda()
    m, v = torch.zeros_like(pd), torch.zeros_like(pd)
    state = torch.tensor([0.0, 0.005, 0.9, 0.999, 1e-8, 0.01, 1.0, 1.0], device="cuda")
    for step in range(4):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        K.adam_step(pd, (2.0 * gr).cuda(), m, v, state, grad_scale=0.5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=1e-5)
    assert state[0].item() == 4.0
