"""GPU: the small kernels that replaced torch glue around the causal-transition layer (csrc/ctmisc.hip) against the torch
expressions they stand for, the shared-bank weight gradients of GroupLinear, and the companion-row encoder pass of CT-MCQ-VAE
against two separate passes."""
import os

import pytest
import torch
import torch.nn.functional as F
import yaml

from ctvae_amd import filler
from tests import helpers as H
from tests.test_ct_gpu import build_ct

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


@pytest.mark.parametrize("rows,C,G,nmat,grouped", [(128, 800, 13, 1, True), (128, 800, 13, 1, False), (256, 64, 21, 3, True),
                                                     (37, 1, 5, 1, True), (130, 70, 4, 2, True)])
def test_group_rowsum_matches_one_hot_matmul(dev, rows, C, G, nmat, grouped):
    from ctvae_amd import native
    g = torch.Generator().manual_seed(rows + C)
    parts = torch.randn(nmat, rows, C, generator=g).to(dev)
    grp = torch.randint(0, G, (rows,), generator=g, dtype=torch.int32).to(dev) if grouped else None
    out = torch.full((nmat, G, C), 7.0, device=dev)
    native.call("ctvae_group_rowsum", parts.data_ptr(), rows * C, nmat, rows, C, C, native.ptr(grp), G, out.data_ptr(), 0)
    sel = F.one_hot(grp.long(), G).float() if grouped else F.one_hot(torch.zeros(rows, dtype=torch.long, device=dev), G).float()
    want = torch.matmul(sel.t().unsqueeze(0).double(), parts.double()).float()
    torch.testing.assert_close(out, want, rtol=1e-5, atol=1e-5)
    again = out.clone()
    native.call("ctvae_group_rowsum", parts.data_ptr(), rows * C, nmat, rows, C, C, native.ptr(grp), G, out.data_ptr(), 1)   # accumulate
    torch.testing.assert_close(out, 2 * again, rtol=1e-6, atol=1e-6)
    out2 = torch.empty_like(out)
    native.call("ctvae_group_rowsum", parts.data_ptr(), rows * C, nmat, rows, C, C, native.ptr(grp), G, out2.data_ptr(), 0)
    assert torch.equal(out2, again), "not reproducible"


def test_mask_blend_posenc_one_hot(dev):
    from ctvae_amd import kernels as K
    g = torch.Generator().manual_seed(5)
    B = 6
    s = torch.rand(2, B, 64, 64, generator=g).to(dev).requires_grad_(True)
    m = torch.rand(B, 64, 1, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(B, 64, 64, generator=g).to(dev)
    out = K.MaskBlend.apply(s, m)
    (out * w).sum().backward()
    s2, m2 = s.detach().clone().requires_grad_(True), m.detach().clone().requires_grad_(True)
    ref = s2[0] * (1 - m2) + s2[1] * m2
    (ref * w).sum().backward()
    assert torch.equal(out, ref)                                       # same arithmetic, same order
    torch.testing.assert_close(s.grad, s2.grad, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(m.grad, m2.grad, rtol=1e-5, atol=1e-5)
    # positional encoding + dropout mask
    x = torch.randn(B, 64, 64, generator=g).to(dev).requires_grad_(True)
    pe = torch.randn(64, 64, generator=g).to(dev)
    keep = (torch.rand(B, 64, 64, generator=g) > 0.1).float().to(dev)
    y = K.PosEncode.apply(x, pe, keep, 1.0 / 0.9)
    (y * w).sum().backward()
    x2 = x.detach().clone().requires_grad_(True)
    yr = (x2 + pe.unsqueeze(0)) * keep * (1.0 / 0.9)
    (yr * w).sum().backward()
    torch.testing.assert_close(y, yr, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(x.grad, x2.grad, rtol=1e-6, atol=1e-6)
    assert torch.equal(K.PosEncode.apply(x.detach(), pe, None, 1.0), x.detach() + pe.unsqueeze(0))
    # one-hot
    inds = torch.randint(0, 64, (B, 4, 8, 8), generator=g).to(dev)
    assert torch.equal(K.one_hot_f32(inds, 64), F.one_hot(inds, 64).float())


def test_group_linear_segments_of_one_bank_share_one_gradient(dev):
    """The discoverers' first layers: four segments over ONE bank (columns 0..D-1 / D..2D-1, matrix 0 for everybody and matrix
    1 + action per sample).  Gradients of the bank, its bias and the input against per-sample torch matmuls."""
    from ctvae_amd import kernels as K
    g = torch.Generator().manual_seed(11)
    B, D, Hd, G = 10, 64, 96, 5
    x = torch.randn(B, 64, D, generator=g).to(dev).requires_grad_(True)
    W = (0.1 * torch.randn(G, Hd, 2 * D, generator=g)).to(dev).requires_grad_(True)
    b = (0.1 * torch.randn(G, Hd, generator=g)).to(dev).requires_grad_(True)
    grp = torch.randint(1, G, (B,), generator=g, dtype=torch.int32).to(dev)
    grp[grp == 3] = 2                                                    # a group nobody uses: its rows must come back zero
    wgt = torch.randn(B, 64, 4 * Hd, generator=g).to(dev)
    y = K.GroupLinear.apply(x, D, Hd, ((0, None), (D, None), (0, grp), (D, grp)), W, None, W, b, W, None, W, b)
    (y * wgt).sum().backward()
    x2, W2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, W, b))
    gl = grp.long()
    ref = torch.cat([x2 @ W2[0, :, :D].t(), x2 @ W2[0, :, D:].t() + b2[0],
                     torch.einsum("bmk,bnk->bmn", x2, W2[gl][:, :, :D]),
                     torch.einsum("bmk,bnk->bmn", x2, W2[gl][:, :, D:]) + b2[gl].unsqueeze(1)], dim=-1)
    (ref * wgt).sum().backward()
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)
    for got, want, name in ((x.grad, x2.grad, "x"), (W.grad, W2.grad, "W"), (b.grad, b2.grad, "b")):
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4 * float(want.abs().max()), msg=lambda m: f"{name}: {m}")
    assert float(W.grad[3].abs().max()) == 0.0 and float(b.grad[3].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["action", "causal"])
def test_companion_rows_equal_two_encoder_passes(dev, mode, monkeypatch):
    """encode_pair (x with autograd, y riding through the same launches) against the two separate passes: outputs, losses and
    every parameter gradient.  The kernels that run differ with the row count (Winograd variants, tile shapes), so the
    comparison is within the parity bound, not bitwise."""
    from ctvae_amd import kernels as K
    from ctvae_amd.models import causal, ct_mcq_vae
    B, A = 8, 12
    xs, ys, act = filler.synthetic_pairs(77, B, A)
    xs, ys, act = xs.to(dev), ys.to(dev), act.to(dev)
    res = {}
    for paired in (True, False):
        monkeypatch.setattr(ct_mcq_vae, "_PAIR_ENCODE", paired)
        m = build_ct(dev, 1250)
        prev = causal.set_noise_source(H.CTNoise(3, dev))
        try:
            out = m(xs, input_y=ys, action=act, mode=[mode] * B)
            losses = m.loss_function(*out, M_N=0.00025)
            m.zero_grad()
            K.backward(losses["loss"])
            m.gather_torch_grads()
        finally:
            causal.set_noise_source(prev)
        res[paired] = (out[0].detach().clone(), {k: float(v) for k, v in losses.items() if torch.is_tensor(v) and v.dim() == 0},
                       m.flat_grads.clone())
    torch.testing.assert_close(res[True][0], res[False][0], rtol=1e-4, atol=1e-4)
    for k, v in res[False][1].items():
        assert abs(res[True][1][k] - v) <= 1e-4 * max(1.0, abs(v)), (k, res[True][1][k], v)
    ga, gb = res[True][2], res[False][2]
    rel = float((ga - gb).norm() / gb.norm().clamp_min(1e-30))
    assert rel <= 1e-4, rel
