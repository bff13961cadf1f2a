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
    # image-side / transposed-conv kernels at their smallest tile grids, and neighbours that must fall back to tap-GEMM
    (False, 32, 3, 32, 3, 1, 1, 0, 1),     # image.hip forward/wgrad/dgrad: 4 x 1 tiles, B = 1
    (False, 32, 3, 24, 3, 1, 1, 0, 2),     # W % 32 != 0 -> masked tap-GEMM
    (False, 3, 32, 48, 3, 2, 1, 0, 2),     # encoder.0-like with a 24 x 24 output -> masked tap-GEMM
    (True, 32, 32, 12, 3, 2, 1, 1, 2),     # final_layer.0-like with H % 8 != 0 -> tap-GEMM classes
    (True, 32, 32, 64, 3, 2, 1, 1, 1),     # upconv.hip with 8 x 2 tiles, B = 1
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


@pytest.mark.parametrize("C,k,s,out,B", [(512, 2, 2, 256, 7), (512, 2, 2, 256, 256), (64, 4, 1, 32, 5), (32, 2, 1, 12, 3)])
def test_linear_over_flatten_as_convolution(K, C, k, s, out, B):
    """CONV_FLAT (CTVAE_W_CI_TAP): nn.Linear over torch.flatten(NCHW) as a k x k convolution of the NHWC tensor, reading and
    writing the Linear layer's own [in][out] weight block; forward, data gradient, weight / bias gradient vs torch."""
    g = torch.Generator().manual_seed(C + k + out)
    h = torch.randn(B, C, k, k, generator=g).requires_grad_(True)
    w = (torch.randn(out, C * k * k, generator=g) / (C * k * k) ** 0.5).requires_grad_(True)
    b = torch.randn(out, generator=g).requires_grad_(True)
    y = F.linear(torch.flatten(h, start_dim=1), w, b)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dev = torch.device("cuda")
    spec = K.ConvSpec(K.CONV_FLAT, C, out, k, s, 0)
    hd = h.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wp = torch.nn.Parameter(w.detach().t().contiguous().to(dev).t())          # logical [out, in] over memory [in][out]
    bp = torch.nn.Parameter(b.detach().to(dev))
    o = K.ConvAct.apply(hd, wp, bp, None, spec)
    assert tuple(o.shape) == (B, 1, 1, out)
    o.backward(gy.view(B, 1, 1, out).to(dev))
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.detach().cpu().view(B, out).numpy(), y.detach().numpy(), atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(hd.grad.cpu().permute(0, 3, 1, 2).numpy(), h.grad.numpy(), atol=TOL, rtol=1e-4)
    scale = max(1.0, float(w.grad.abs().max()))
    np.testing.assert_allclose(wp.grad.cpu().numpy(), w.grad.numpy(), atol=TOL * scale, rtol=1e-4)
    np.testing.assert_allclose(bp.grad.cpu().numpy(), b.grad.numpy(), atol=TOL * max(1.0, float(b.grad.abs().max())), rtol=1e-4)


@pytest.mark.parametrize("L,C,h,B", [(128, 512, 2, 64), (128, 512, 2, 256), (128, 512, 2, 7), (64, 32, 3, 130)])
def test_linear_into_nhwc_view(K, L, C, h, B):
    """ctvae_linear_pixmajor_forward: nn.Linear(L, C*h*h) + .view(-1, C, h, h) (vanilla_vae.py:101-102) written as the NHWC tensor by
    the GEMM's epilogue; forward and every gradient vs torch, and bit-equal to the two-launch path (ConvAct + layout change)."""
    g = torch.Generator().manual_seed(L + C + B)
    z = torch.randn(B, L, generator=g).requires_grad_(True)
    w = (torch.randn(C * h * h, L, generator=g) / L ** 0.5).requires_grad_(True)
    b = torch.randn(C * h * h, generator=g).requires_grad_(True)
    y = F.linear(z, w, b).view(B, C, h, h)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    dev = torch.device("cuda")
    spec = K.ConvSpec(K.CONV, L, C * h * h, 1)
    assert K.LinearToNHWC.supported(B, L, C, h * h, dev)
    gy_nhwc = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    outs = []
    for fused in (True, False):
        zd = z.detach().view(B, 1, 1, L).to(dev).requires_grad_(True)
        wp = torch.nn.Parameter(w.detach().t().contiguous().to(dev).t())          # logical [out, in] over memory [in][out]
        bp = torch.nn.Parameter(b.detach().to(dev))
        if fused:
            o = K.LinearToNHWC.apply(zd, wp, bp, spec, C, h, h)
        else:
            o = K._ToNHWC.apply(K.ConvAct.apply(zd, wp, bp, None, spec).view(B, C, h, h))
        assert tuple(o.shape) == (B, h, h, C)
        o.backward(gy_nhwc)
        torch.cuda.synchronize()
        outs.append((o.detach().cpu(), zd.grad.cpu(), wp.grad.cpu(), bp.grad.cpu()))
    o, gz, gw, gb = outs[0]
    np.testing.assert_allclose(o.permute(0, 3, 1, 2).numpy(), y.detach().numpy(), atol=TOL, rtol=1e-4)
    np.testing.assert_allclose(gz.view(B, L).numpy(), z.grad.numpy(), atol=TOL * max(1.0, float(z.grad.abs().max())), rtol=1e-4)
    np.testing.assert_allclose(gw.numpy(), w.grad.numpy(), atol=TOL * max(1.0, float(w.grad.abs().max())), rtol=1e-4)
    np.testing.assert_allclose(gb.numpy(), b.grad.numpy(), atol=TOL * max(1.0, float(b.grad.abs().max())), rtol=1e-4)
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)


def test_linear_into_nhwc_view_says_no_where_the_tile_kernel_does_not_run(K):
    dev = torch.device("cuda")
    assert not K.LinearToNHWC.supported(4, 20, 512, 4, dev)          # in_features not a multiple of the 32-wide K chunk


@pytest.mark.parametrize("B,C,P", [(5, 3, 64 * 64), (2, 3, 36), (3, 3, 35), (4, 512, 4), (2, 7, 10)])
def test_permute_both_ways(K, B, C, P):
    """ctvae_permute (NCHW <-> NHWC copies): the 3-channel quad kernel, and the element-wise one for everything else."""
    dev = torch.device("cuda")
    x = torch.randn(B, C, P, generator=torch.Generator().manual_seed(P)).to(dev)
    nhwc = K.permute_raw(x, B, C, P, True).view(B, P, C)
    assert torch.equal(nhwc, x.permute(0, 2, 1).contiguous())
    back = K.permute_raw(nhwc.contiguous(), B, C, P, False).view(B, C, P)
    assert torch.equal(back, x)


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


@pytest.mark.parametrize("rows,Q", [(2 * 64, 40), (33, 10), (7, 1), (19, 64), (5, 100), (3, 200), (4096 * 3 + 5, 40)])
def test_gumbel_softmax_and_categorical_kl(K, rows, Q):
    """CategoricalVAE's latent kernels (catlatent.hip) against the torch expressions of cat_vae.py:125-130,147,160-167:
    group sizes 1..64 lanes, 1/2/4 categories per lane, ragged row counts, grid-stride tail."""
    g = torch.Generator().manual_seed(rows * 1000 + Q)
    B = 3 if rows % 3 == 0 else 1
    z = (2.0 * torch.randn(rows, Q, generator=g)).view(B, rows // B, Q).requires_grad_(True)
    u = torch.rand(B, rows // B, Q, generator=g)
    wgt = torch.randn(B, rows // B, Q, generator=g)
    temp, eps = 0.5, 1e-7
    gum = -torch.log(-torch.log(u + eps) + eps)
    s = F.softmax((z + gum) / temp, dim=-1)
    q_p = F.softmax(z, dim=-1)
    kld = torch.mean(torch.sum(q_p * torch.log(q_p + eps) - q_p * np.log(1.0 / Q + eps), dim=(1, 2)), dim=0)
    ((s * wgt).sum() + 0.7 * kld).backward()
    zd = z.detach().cuda().requires_grad_(True)
    sd = K.GumbelSoftmax.apply(zd, u.cuda(), temp, eps)
    kd = K.CatKL.apply(zd, eps)
    ((sd * wgt.cuda()).sum() + 0.7 * kd).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(sd.detach().cpu().numpy(), s.detach().numpy(), atol=2e-6, rtol=1e-5)
    assert abs(kd.item() - kld.item()) <= 1e-5 * max(1.0, abs(kld.item()))
    np.testing.assert_allclose(zd.grad.cpu().numpy(), z.grad.numpy(), atol=2e-6, rtol=1e-4)


@pytest.mark.parametrize("B,lead,n,L", [(3, (4,), 1024, 16), (2, (3, 5), 12288, 128), (5, (1,), 64, 300), (1, (2, 7), 256, 1)])
def test_importance_weighted_loss(K, B, lead, n, L):
    """csrc/iwloss.hip against the torch expressions of iwae.py:139-160 / miwae.py:146-163 (weights not detached)."""
    from oracle import vae_cpu as O
    g = torch.Generator().manual_seed(B * 100 + n)
    R = B * int(np.prod(lead))
    r = torch.rand((B,) + lead + (n // 4, 2, 2), generator=g).requires_grad_(True)
    x = torch.rand(B, n // 4, 2, 2, generator=g)
    mu = torch.randn((B,) + lead + (L,), generator=g).requires_grad_(True)
    lv = (0.3 * torch.randn((B,) + lead + (L,), generator=g)).requires_grad_(True)
    want = O.iw_loss(r, x, mu, lv, 0.01)
    want["loss"].backward()
    rd, md, ld = (t.detach().cuda().requires_grad_(True) for t in (r, mu, lv))
    out = K.IWLoss.apply(rd.reshape(R, -1), x.cuda().reshape(B, -1), md.reshape(R, L), ld.reshape(R, L), lead[-1], 0.01)
    out[0].backward()
    torch.cuda.synchronize()
    for i, k in ((0, "loss"), (1, "Reconstruction_Loss"), (3, "KLD")):
        assert abs(out[i].item() - want[k].item()) <= 1e-5 * max(1.0, abs(want[k].item())), k
    np.testing.assert_allclose(rd.grad.cpu().numpy(), r.grad.numpy(), atol=1e-9, rtol=1e-3)
    np.testing.assert_allclose(md.grad.cpu().numpy(), mu.grad.numpy(), atol=1e-8, rtol=1e-3)
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lv.grad.numpy(), atol=1e-8, rtol=1e-3)


@pytest.mark.parametrize("N,D,kind", [(4, 128, "imq"), (37, 128, "rbf"), (256, 128, "imq"), (9, 20, "imq"), (5, 300, "rbf"), (2, 512, "imq")])
def test_mmd_kernel(K, N, D, kind):
    """csrc/mmd.hip against the torch expressions of wae_mmd.py:120-203: all three kernel sums, mmd and d mmd / d z."""
    from oracle import vae_cpu as O
    g = torch.Generator().manual_seed(N * 7 + D)
    z = (1.5 * torch.randn(N, D, generator=g)).requires_grad_(True)
    p = torch.randn(N, D, generator=g)
    w = (0.3, 0.7, 1.1)
    pp, zz, pz = O.mmd_terms(z, p, kind, 2.0)
    (w[0] * pp + w[1] * zz - 2 * w[2] * pz).backward()
    zd = z.detach().cuda().requires_grad_(True)
    out = K.MMD.apply(zd, p.cuda(), kind, 2.0 * D * 2.0, *w)
    (out[0] * 1.0).backward()
    torch.cuda.synchronize()
    for got, want in zip([o.item() for o in out[1:]], (pp.item(), zz.item(), pz.item())):
        assert abs(got - want) <= 2e-5 * max(1.0, abs(want)), (got, want)
    np.testing.assert_allclose(zd.grad.cpu().numpy(), z.grad.numpy(), atol=1e-7, rtol=2e-4)


@pytest.mark.parametrize("case", [
    # transposed, Ci, Co, H, k, s, p, op, B, bias    (shapes whose two backward GEMMs take the 64x64 tile kernels -> ONE launch)
    (False, 64, 128, 16, 3, 2, 1, 0, 16, True),      # encoder.2-like: data gradient in 4 parity classes
    (True, 256, 128, 4, 3, 2, 1, 1, 32, True),       # decoder.1-like: split-K data gradient (finish rides with the slab reduction)
    (False, 2048, 256, 1, 1, 1, 0, 0, 64, True),     # fc heads
    (False, 256, 256, 8, 1, 1, 0, 0, 8, False),      # 1x1 conv of a residual block
    (False, 32, 3, 32, 3, 1, 1, 0, 2, True),         # picture-side layer: neither side pairs, same entry point
    (False, 96, 160, 5, 3, 1, 1, 0, 3, True),        # ragged tiles on every side: M = 75, N = 96 / 160, K = 864
    (True, 160, 96, 3, 4, 2, 1, 0, 5, False),        # k4 s2 transposed conv, channels not multiples of 64
    (False, 64, 64, 6, 3, 2, 1, 0, 1, True),         # B = 1: 3 x 3 output, one partial tile per class (odd sizes are refused by the stride-2 data-gradient geometry)
    (False, 128, 32, 9, 1, 1, 0, 0, 7, True),        # N = 32 on the weight-gradient side (128 x 32 tiles: only the data gradient records)
])
def test_paired_backward_matches_separate_launches(K, case):
    """ctvae_conv_backward (data + weight gradient in one launch, shared finishing launch) against the separate
    ctvae_conv_wgrad / ctvae_conv_dgrad calls on the same inputs: the weight gradient runs the same arithmetic in the same
    order (bit-identical); the data gradient may be split over fewer K slices when it shares the launch (summation order
    of the slices differs: a few ulp)."""
    transposed, Ci, Co, H, k, s, p, op, B, bias = case
    g = torch.Generator().manual_seed(Ci * 7 + Co)
    spec = K.ConvSpec(K.CONVT if transposed else K.CONV, Ci, Co, k, s, p, op, K.ACT_NONE)
    ho, wo = spec.out_hw(H, H)
    x = torch.randn(B, H, H, Ci, generator=g).cuda()
    dy = torch.randn(B, ho, wo, Co, generator=g).cuda()
    w = torch.randn((Ci, Co, k, k) if transposed else (Co, Ci, k, k), generator=g) * 0.05
    res = []
    for paired in (True, False):
        wp = as_param(pack(w, transposed).cuda(), transposed)
        bp = torch.nn.Parameter(torch.zeros(Co).cuda()) if bias else None
        if paired:
            dx = K.conv_backward_raw(x, dy, wp, bp, spec)
        else:
            K.conv_wgrad_raw(x, dy, wp, bp, spec)
            dx = K.conv_dgrad_raw(dy, wp, spec, (H, H))
        torch.cuda.synchronize()
        res.append((dx.clone(), wp.grad.clone(), bp.grad.clone() if bias else None))
    assert torch.equal(res[0][1], res[1][1])
    np.testing.assert_allclose(res[0][0].cpu().numpy(), res[1][0].cpu().numpy(), rtol=2e-5, atol=2e-5 * float(res[1][0].abs().max()))
    if bias:
        assert torch.equal(res[0][2], res[1][2])


@pytest.mark.parametrize("B,D", [(4, 128), (256, 128), (7, 20), (300, 40), (3, 300)])
def test_dip_regulariser(K, B, D):
    """csrc/dip.hip against the torch expression of dip_vae.py:147-159 (B < D and B > D: the scalar variance term takes the
    main diagonal of the [B,D] matrix), value and both gradients."""
    g = torch.Generator().manual_seed(B * 31 + D)
    mu = torch.randn(B, D, generator=g).requires_grad_(True)
    lv = (0.3 * torch.randn(B, D, generator=g)).requires_grad_(True)
    centered = mu - mu.mean(dim=1, keepdim=True)
    cov_z = centered.t().matmul(centered) + torch.mean(torch.diagonal((2. * lv).exp(), dim1=0), dim=0)
    cd = torch.diag(cov_z)
    want = 0.1 * torch.sum((cov_z - torch.diag(cd)) ** 2) + 0.05 * torch.sum((cd - 1) ** 2)
    (want * 0.5).backward()
    md, ld = (t.detach().cuda().requires_grad_(True) for t in (mu, lv))
    got = K.DIPLoss.apply(md, ld, 0.05, 0.1)
    (got * 0.5).backward()
    torch.cuda.synchronize()
    assert abs(got.item() - want.item()) <= 2e-5 * max(1.0, abs(want.item()))
    sc = float(mu.grad.abs().max())
    np.testing.assert_allclose(md.grad.cpu().numpy(), mu.grad.numpy(), atol=2e-5 * sc, rtol=2e-4)
    np.testing.assert_allclose(ld.grad.cpu().numpy(), lv.grad.numpy(), atol=2e-5 * max(1e-6, float(lv.grad.abs().max())), rtol=2e-4)


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
    state = K.adam_state([0.0, 0.005, 0.9, 0.999, 1e-8, 0.01, 1.0, 1.0], "cuda")
    for step in range(4):
        gr = torch.randn(n, generator=g)
        ref.grad = gr.clone()
        opt.step()
        K.adam_step(pd, (2.0 * gr).cuda(), m, v, state, grad_scale=0.5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(pd.cpu().numpy(), ref.detach().numpy(), atol=2e-6, rtol=1e-5)
    assert state[0].item() == 4.0


@pytest.mark.parametrize("transposed,B,H", [(False, 32, 32), (True, 128, 16), (False, 4, 16), (True, 3, 8)])
def test_bn_backward_sums_fused_into_dgrad(K, transposed, B, H):
    """Two chained ConvBNAct blocks: the second block's dgrad emits the first BatchNorm's backward sums
    (ctvae_conv_dgrad_bn).  All gradients against torch autograd, and the fused path must really have run."""
    from ctvae_amd import native
    from ctvae_amd.models import blocks
    g = torch.Generator().manual_seed(31 + B)
    C0, C1, C2 = 32, 64, 32
    x = torch.randn(B, C0, H, H, generator=g).requires_grad_(True)
    mk = lambda ci, co: (torch.randn((ci, co, 3, 3) if transposed else (co, ci, 3, 3), generator=g) / (ci * 9) ** 0.5).requires_grad_(True)
    w1, w2 = mk(C0, C1), mk(C1, C2)
    b1, b2 = torch.randn(C1, generator=g).requires_grad_(True), torch.randn(C2, generator=g).requires_grad_(True)
    gm1, bt1 = (torch.rand(C1, generator=g) + 0.5).requires_grad_(True), (torch.rand(C1, generator=g) - 0.5).requires_grad_(True)
    gm2, bt2 = (torch.rand(C2, generator=g) + 0.5).requires_grad_(True), (torch.rand(C2, generator=g) - 0.5).requires_grad_(True)

    def conv(t, w, b):
        return F.conv_transpose2d(t, w, b, stride=2, padding=1, output_padding=1) if transposed else F.conv2d(t, w, b, stride=1, padding=1)

    def bn(t, gm, bt):
        C = t.shape[1]
        return F.leaky_relu(F.batch_norm(t, torch.zeros(C), torch.ones(C), gm, bt, True, 0.1, 1e-5), 0.01)

    out = bn(conv(bn(conv(x, w1, b1), gm1, bt1), w2, b2), gm2, bt2)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)

    dev = torch.device("cuda")
    kind = K.CONVT if transposed else K.CONV
    s, op = (2, 1) if transposed else (1, 0)
    sp1 = K.ConvSpec(kind, C0, C1, 3, s, 1, op, K.ACT_NONE)
    sp2 = K.ConvSpec(kind, C1, C2, 3, s, 1, op, K.ACT_NONE)

    class Holder:
        pass

    def mkconv(w, b):
        h = Holder()
        h.weight, h.bias = as_param(pack(w.detach(), transposed).to(dev), transposed), torch.nn.Parameter(b.detach().to(dev))
        return h

    def mkbn(gm, bt):
        h = Holder()
        C = gm.numel()
        h.weight, h.bias = torch.nn.Parameter(gm.detach().to(dev)), torch.nn.Parameter(bt.detach().to(dev))
        h.running_mean, h.running_var = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        h.num_batches_tracked = torch.zeros((), dtype=torch.long, device=dev)
        return h

    def run(fused):
        c1, c2, n1, n2 = mkconv(w1, b1), mkconv(w2, b2), mkbn(gm1, bt1), mkbn(gm2, bt2)
        xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
        a1 = blocks.conv_bn_leaky(xd, c1, n1, sp1, True)
        assert hasattr(a1, "_ctvae_bn_link")
        if not fused:
            del a1._ctvae_bn_link
        a2 = blocks.conv_bn_leaky(a1, c2, n2, sp2, True)
        native.prof_enable(True)
        a2.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
        torch.cuda.synchronize()
        native.prof_enable(False)
        rep = native.prof_report()
        return xd.grad, c1, c2, n1, n2, rep

    gx_f, c1, c2, n1, n2, rep_f = run(True)
    gx_u, *_rest, rep_u = run(False)
    ws = native.workspace(dev)
    Hin = H * 2 if transposed else H          # input size of the second block
    # the plan of the call backward really makes: the paired launch (ctvae_conv_backward) splits K later than a lone data gradient
    rows_of = native.load().ctvae_conv_backward_bn_rows if K._PAIR else native.load().ctvae_conv_dgrad_bn_rows
    rows = rows_of(kind, B, Hin, Hin, C1, C2, 3, s, 1, op, ws.numel() * 4)
    if B >= 32:
        assert rows > 0, "this shape is expected to take the fused path"
    assert rep_u["bn_bwd_partial_kernel"]["count"] == 2
    assert rep_f["bn_bwd_partial_kernel"]["count"] == (1 if rows > 0 else 2), "fused BN-backward sums not used as planned"
    ref = x.grad.permute(0, 2, 3, 1).numpy()
    sc = max(1.0, float(np.abs(ref).max()))
    # fused vs separate pass: same signs of the LeakyReLU arguments on both sides -> strict
    np.testing.assert_allclose(gx_f.cpu().numpy(), gx_u.cpu().numpy(), atol=TOL * sc, rtol=1e-3)
    # vs torch CPU: a pre-activation within rounding of 0 may take the other LeakyReLU slope (x100 on that element's
    # gradient, spread over its 3x3 neighbourhood by the dgrad) -> allow a vanishing fraction of outliers
    bad = np.abs(gx_f.cpu().numpy() - ref) > 2 * TOL * sc + 1e-3 * np.abs(ref)
    assert bad.mean() < 5e-4, f"{bad.sum()} of {bad.size} elements differ"
    # parameter gradients sum over up to 5e5 pixels, so one flipped slope moves them visibly: relative L2 error
    for got, want in ((n1.weight.grad, gm1.grad), (n1.bias.grad, bt1.grad), (n2.weight.grad, gm2.grad), (n2.bias.grad, bt2.grad),
                      (c1.weight.grad, w1.grad), (c2.weight.grad, w2.grad)):
        err = float((got.cpu() - want).norm() / want.norm())
        assert err < 2e-3, err
    # (conv biases in front of train-mode BN have an analytically zero gradient -- pure summation noise, not compared)


CHAIN_CASES = [
    # transposed, B, H, channel widths, kernels that must have run            what the round-3 paths must cover
    (False, 8, 8, (128, 256, 512), ("bn_fused_fwd_kernel", "bn_fused_bwd_kernel")),    # deep stride-2 encoder blocks: dense forward slices, class-major backward slices
    (True, 8, 2, (256, 128, 64, 32), ("bn_fused_fwd_kernel", "bn_fused_bwd_kernel")),  # transposed decoder blocks: class-major forward slices, dense backward slices, 2-channel owners
    (False, 8, 32, (32, 64, 128, 256), ("bn_fused_fwd_kernel",)),                      # 2048 -> 512 -> 128 pixels; short paired data gradients stay unsplit
    (False, 64, 32, (32, 64, 128), ("lazy-apply",)),                                   # bs = 64 shapes of encoder.1 / encoder.2: the unsplit producer hands its raw output on
    (False, 3, 16, (32, 64, 128), ("bn_fused_fwd_kernel",)),                           # odd batch: 192 / 48 / 12 pixels, lanes beyond R
]


@pytest.mark.parametrize("case", CHAIN_CASES)
def test_chain_of_bn_blocks_small_batch_paths(K, case):
    """blocks.Chain of Conv|ConvTranspose -> BatchNorm -> LeakyReLU blocks (vanilla_vae.py:25-35,47-62) at small batch against torch
    autograd on the CPU: outputs, running statistics, every gradient.  The shapes are chosen so that the round-3 paths run --
    split-K slices kept channel-major and finished by bn_fused_fwd_kernel / bn_fused_bwd_kernel, the data gradient in front of a
    BatchNorm handed over as slices (placeholder tensor), and a block whose apply would be a launch of its own handing its
    consumer the raw tensor (tile kernel / weight-gradient kernel with the lazy apply) -- and the test asserts that they did."""
    from ctvae_amd import native
    from ctvae_amd.models import blocks
    transposed, B, H, widths, expect = case
    g = torch.Generator().manual_seed(77 + B + H)
    dev = torch.device("cuda")
    x = torch.randn(B, widths[0], H, H, generator=g).requires_grad_(True)
    ws_, bs_, gms, bts = [], [], [], []
    for ci, co in zip(widths[:-1], widths[1:]):
        ws_.append((torch.randn((ci, co, 3, 3) if transposed else (co, ci, 3, 3), generator=g) / (ci * 9) ** 0.5).requires_grad_(True))
        bs_.append(torch.randn(co, generator=g).requires_grad_(True))
        gms.append((torch.rand(co, generator=g) + 0.5).requires_grad_(True))
        bts.append((torch.rand(co, generator=g) - 0.5).requires_grad_(True))
    # torch reference
    t = x
    rstats = []
    for w, b, gm, bt in zip(ws_, bs_, gms, bts):
        t = F.conv_transpose2d(t, w, b, stride=2, padding=1, output_padding=1) if transposed else F.conv2d(t, w, b, stride=2, padding=1)
        rm, rv = torch.zeros(t.shape[1]), torch.ones(t.shape[1])
        t = F.leaky_relu(F.batch_norm(t, rm, rv, gm, bt, True, 0.1, 1e-5), 0.01)
        rstats.append((rm, rv))
    go = torch.randn(t.shape, generator=g)
    t.backward(go)
    # product
    mods = []
    for (ci, co), w, b, gm, bt in zip(zip(widths[:-1], widths[1:]), ws_, bs_, gms, bts):
        m = blocks.ConvBNLeaky(ci, co, 3, 2, 1, out_pad=1 if transposed else 0, transposed=transposed)
        with torch.no_grad():
            m._modules["0"].weight.copy_(w)
            m._modules["0"].bias.copy_(b)
            m._modules["1"].weight.copy_(gm)
            m._modules["1"].bias.copy_(bt)
        mods.append(m)
    chain = blocks.Chain(*mods).to(dev).train()
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    native.prof_enable(True)
    out = chain(xd)
    out.backward(go.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    native.prof_enable(False)
    rep = native.prof_report()
    names = " ".join(rep)
    for want in expect:
        if want == "lazy-apply":
            assert "false,3,true>" in names or "false,0,true>" in names, sorted(rep)     # the tile kernel with the lazy apply
            assert rep.get("bn_apply_act_kernel", {"count": 0})["count"] < len(mods), sorted(rep)
        else:
            assert want in rep, (want, sorted(rep))
    sc = max(1.0, float(t.detach().abs().max()))
    bad = np.abs(out.detach().cpu().permute(0, 3, 1, 2).numpy() - t.detach().numpy()) > 2 * TOL * sc
    assert bad.mean() < 5e-4, f"{bad.sum()} of {bad.size} outputs differ"       # (a pre-activation within rounding of 0 may flip its slope)
    for m, (rm, rv) in zip(mods, rstats):
        np.testing.assert_allclose(m._modules["1"].running_mean.cpu().numpy(), rm.numpy(), atol=1e-5, rtol=1e-4)
        np.testing.assert_allclose(m._modules["1"].running_var.cpu().numpy(), rv.numpy(), atol=1e-5, rtol=1e-4)
        assert int(m._modules["1"].num_batches_tracked) == 1
    ref = x.grad.permute(0, 2, 3, 1).numpy()
    gsc = max(1.0, float(np.abs(ref).max()))
    badg = np.abs(xd.grad.cpu().numpy() - ref) > 2 * TOL * gsc + 2e-3 * np.abs(ref)
    assert badg.mean() < 2e-3, f"{badg.sum()} of {badg.size} input-gradient elements differ"
    for m, w, gm, bt in zip(mods, ws_, gms, bts):
        for got, want in ((m._modules["0"].weight.grad, w.grad), (m._modules["1"].weight.grad, gm.grad), (m._modules["1"].bias.grad, bt.grad)):
            err = float((got.cpu() - want).norm() / want.norm())
            assert err < 2e-3, err


WINO_CASES = [
    # B, H, Ci, Co     (fewer than 200 workgroups of 64 tiles x 64 channels -> the frequency-split kernel, 64 x 32)
    (256, 8, 256, 256),    # the MCQ-VAE residual 3x3 at its bench batch: 256 workgroups, wino_conv_kernel
    (100, 16, 64, 128),    # 200 workgroups, one image per block, wino_conv_kernel
    (128, 8, 256, 256),    # the CT-MCQ-VAE per-GPU batch: 128 -> wino_conv_fs_kernel; four images per workgroup block
    (130, 8, 256, 256),    # batch not a multiple of the images per block
    (64, 16, 128, 128),    # one image per block
    (32, 32, 64, 64),      # 2 x 2 blocks of 8 x 8 tiles per image
    (512, 4, 256, 256),    # sixteen images per block
]


@pytest.mark.parametrize("with_bias", [True, False])
@pytest.mark.parametrize("case", WINO_CASES)
def test_winograd_conv3x3(K, case, with_bias):
    """3x3 / stride 1 / pad 1 layers run Winograd (wino.hip): F(2x2,3x3) for forward and data gradient, F(3x3,2x2) for
    the weight gradient (bias gradient from the same pass).  Against torch's direct convolution within the 1e-4
    bound, and the kernels must really have been the ones that ran."""
    from ctvae_amd import native
    B, H, Ci, Co = case
    g = torch.Generator().manual_seed(500 + B + H)
    x = torch.randn(B, Ci, H, H, generator=g).requires_grad_(True)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True) if with_bias else None
    ya = F.relu(F.conv2d(x, w, b, stride=1, padding=1))
    gy = torch.randn(ya.shape, generator=g)
    ya.backward(gy)

    dev = torch.device("cuda")
    spec = K.ConvSpec(K.CONV, Ci, Co, 3, 1, 1, 0, K.ACT_RELU)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wp = as_param(pack(w.detach(), False).to(dev), False)
    bp = torch.nn.Parameter(b.detach().to(dev)) if with_bias else None
    native.prof_enable(True)
    out = K.ConvAct.apply(xd, wp, bp, None, spec)
    out.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    torch.cuda.synchronize()
    native.prof_enable(False)
    rep = native.prof_report()
    nconv = sum(rep.get(k, {"count": 0})["count"] for k in ("wino_conv_kernel", "wino_conv_fs_kernel"))
    assert nconv == 2, sorted(rep)                                     # forward + data gradient
    fwd = "wino_conv_kernel" if (B * (H // 2) ** 2 + 63) // 64 * (Co // 64) >= 200 else "wino_conv_fs_kernel"
    assert fwd in rep, sorted(rep)
    assert ("wino_wgrad_kernel" in rep) == (H % 8 == 0), sorted(rep)   # chunks of 4 x 8 output pixels
    np.testing.assert_allclose(out.detach().cpu().permute(0, 3, 1, 2).numpy(), ya.detach().numpy(), atol=TOL, rtol=1e-4)
    # a pre-activation within rounding of 0 may land on the other side of the ReLU (one flip moves the 3x3 x Ci input
    # gradients under it by O(1)): allow a vanishing fraction of outliers, as in the BatchNorm chain tests above
    sc = max(1.0, float(x.grad.abs().max()))
    got, ref = xd.grad.cpu().permute(0, 3, 1, 2).numpy(), x.grad.numpy()
    bad = np.abs(got - ref) > TOL * sc + 1e-4 * np.abs(ref)
    assert bad.mean() < 2e-3, f"{bad.sum()} of {bad.size} elements differ"
    dw = (wp.grad.cpu() - w.grad).numpy()
    scale = max(1.0, float(w.grad.abs().max()))
    assert (np.abs(dw) > TOL * scale).mean() < 1e-2, "weight gradient differs beyond what a ReLU flip or two explain"
    assert float(np.linalg.norm(dw) / w.grad.norm()) < 5e-3
    if with_bias:
        db = (bp.grad.cpu() - b.grad).numpy()
        off = np.abs(db) > 1e-3 * max(1.0, float(b.grad.abs().max()))
        assert off.sum() <= 3, "bias gradient (a ReLU flip moves one channel's sum by O(1); more than a few is a bug)"


def test_gaussian_latent_node(K):
    """kernels.GaussianLatent (mu | log_var views + reparameterisation as one node) against the torch expression of
    vanilla_vae.py:107-117 with injected noise, and the statistics / stream advance of the in-kernel Philox noise."""
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    B, L = 64, 128
    heads = torch.randn(B, 2 * L, generator=g) * 0.5
    eps = torch.randn(B, L, generator=g)
    wz, wm, wl = (torch.randn(B, L, generator=g) for _ in range(3))
    hr = heads.clone().requires_grad_(True)
    mu_r, lv_r = hr[:, :L], hr[:, L:]
    z_r = eps * torch.exp(0.5 * lv_r) + mu_r
    ((z_r * wz).sum() + (mu_r * wm).sum() + (lv_r * wl).sum()).backward()
    hd = heads.to(dev).requires_grad_(True)
    mu, lv, z = K.GaussianLatent.apply(hd, eps.to(dev), None)
    ((z * wz.to(dev)).sum() + (mu * wm.to(dev)).sum() + (lv * wl.to(dev)).sum()).backward()
    np.testing.assert_allclose(z.detach().cpu().numpy(), z_r.detach().numpy(), atol=1e-6, rtol=1e-6)
    np.testing.assert_allclose(hd.grad.cpu().numpy(), hr.grad.numpy(), atol=1e-6, rtol=1e-5)
    # in-kernel noise: N(0,1) moments over 2^20 draws, reproducible for one stream position, fresh after a backward
    B2 = 8192
    h2 = torch.zeros(B2, 2 * L, device=dev, requires_grad=True)           # mu = 0, log_var = 0 -> z == eps
    rng = torch.tensor([1234567, 0], dtype=torch.int64, device=dev)
    _, _, z1 = K.GaussianLatent.apply(h2, None, rng)
    _, _, z1b = K.GaussianLatent.apply(h2, None, rng)
    assert torch.equal(z1, z1b), "same key and stream position must give the same noise"
    z1.sum().backward()
    assert int(rng[1]) == 1
    _, _, z2 = K.GaussianLatent.apply(h2, None, rng)
    assert not torch.equal(z1, z2)
    v = z1.detach().double().flatten()
    n = v.numel()
    assert abs(float(v.mean())) < 5.0 / n ** 0.5 and abs(float(v.var()) - 1.0) < 0.01
    assert abs(float((v ** 3).mean())) < 0.02 and abs(float((v ** 4).mean()) - 3.0) < 0.05
    assert abs(float((z1.detach().flatten()[:-1] * z1.detach().flatten()[1:]).mean())) < 0.01      # neighbours uncorrelated
    assert abs(float((z1.detach() * z2.detach()).mean())) < 0.01                                      # steps uncorrelated
