#!/usr/bin/env python3
"""Generate tests/golden/ct_*.npz from the REFERENCE's own ``models/ct_mcq_vae.py``.  TEST INFRASTRUCTURE ONLY.

Runs only in the build container (needs /root/reference).  ``ct_mcq_vae.py`` imports ``torch_geometric`` (lines 2, 5),
which is absent here and not installable.  This script registers an EMPTY ``torch_geometric`` package in ``sys.modules``
solely so that those two import lines execute; it holds no arithmetic:

* ``gnn.GATv2Conv`` / ``gnn.Sequential`` are placeholders that take any constructor arguments and RAISE when called, so no
  fixture can contain a value that went through a stand-in GATv2 — those two layers stay "parity unpinned";
* ``torch_geometric.utils.dense_to_sparse`` hands the dense (padded) adjacency through untouched.

With that, the reference's file is exec'd unmodified where it lies (like ``gen_golden.py`` does for the other models)
and fixtures are captured from its own methods: ``_compute_mask``, ``_compute_adj__comp_optim``, ``_sample_bernoulli``,
the regularisers, ``latent_loss``, the accuracies, ``ct_preprocess`` / ``ct_postprocess``, ``loss_function``.
``_compute_y`` and the three modes need *a* graph network between the reference's pre- and post-processing: the script
assigns ``tests.helpers.GNNDouble`` (a plain differentiable function of (nodes, dense adjacency), declared a test double,
not GATv2) to ``ct_layer.graph_transitioner``; the parity tests install the same double on the product, so everything
around the GNN — padding, head gather, mask blend, softmax, mode dispatch, losses — is pinned by the reference's own code.

Noise is injected by patching the draw primitives the reference reaches — ``Tensor.exponential_`` (inside the
real ``F.gumbel_softmax``), ``F.dropout`` (``nn.Dropout`` of PositionalEncoding), ``torch.rand`` (adjacency_KL_loss) — to
return ``ctvae_amd.filler.ct_noise`` draws, in the order listed per mode below.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_ct_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("CTVAE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from ctvae_amd import filler                    # noqa: E402
from oracle.gen_golden import load_reference_models, cks, np32, CT_CONV_CFG   # noqa: E402
from tests import helpers as H                  # noqa: E402


class _Absent(nn.Module):
    """Placeholder for a torch_geometric class: constructible (CausalTransition.__init__ builds two), never callable."""

    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise RuntimeError("torch_geometric is absent: GATv2Conv / gnn.Sequential are not available (parity unpinned)")


def load_reference_ct():
    load_reference_models()                                    # models.{types_,base,vanilla_vae,vq_vae,mcq_vae}
    tg = types.ModuleType("torch_geometric")
    tg.nn = types.ModuleType("torch_geometric.nn")
    tg.utils = types.ModuleType("torch_geometric.utils")
    tg.nn.GATv2Conv = _Absent
    tg.nn.Sequential = _Absent
    tg.utils.dense_to_sparse = lambda adj: (adj, None)         # hand-through to the test double, no edge arithmetic
    sys.modules["torch_geometric"] = tg
    sys.modules["torch_geometric.nn"] = tg.nn
    sys.modules["torch_geometric.utils"] = tg.utils
    spec = importlib.util.spec_from_file_location("models.ct_mcq_vae", os.path.join(REF, "models", "ct_mcq_vae.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["models.ct_mcq_vae"] = mod
    spec.loader.exec_module(mod)
    return mod


class RefTransitioner(nn.Module):
    """Adapter: the reference calls graph_transitioner(nodes [B*N,D], edge_index, edge_attr=...); edge_index is the dense
    padded adjacency handed through by the placeholder dense_to_sparse."""

    def __init__(self, double):
        super().__init__()
        self.fn = double

    def forward(self, nodes, edge_index, edge_attr=None):
        B, N, _ = edge_index.shape
        return self.fn(nodes.view(B, N, -1), edge_index).reshape(B * N, -1)


SCRIPT_BASE = [("dropout", "pos_dropout"), ("exp", "adj_gumbel")]
SCRIPT_ACTION = [("dropout", "mask_dropout"), ("exp", "mask_gumbel"), ("dropout", "pos_dropout"), ("exp", "adj_gumbel"),
                 ("rand", "kl_target")]


class RefNoise:
    """Context: the reference's draws come from filler.ct_noise in the scripted order (asserted)."""

    def __init__(self, seed, script):
        self.seed, self.script, self.pos, self.counts = seed, list(script), 0, {}

    def _next(self, kind, shape, p=0.0):
        assert self.pos < len(self.script), f"unexpected extra {kind} draw"
        k_kind, tag = self.script[self.pos]
        assert k_kind == kind, f"draw {self.pos}: reference asked for {kind}, script says {k_kind}:{tag}"
        self.pos += 1
        k = self.counts.get(tag, 0)
        self.counts[tag] = k + 1
        return filler.ct_noise(self.seed, tag, k, shape, p)

    def __enter__(self):
        self._exp, self._drop, self._rand = torch.Tensor.exponential_, F.dropout, torch.rand
        me = self

        def exponential_(t, lambd=1, *, generator=None):
            if generator is not None:
                return me._exp(t, lambd, generator=generator)
            return t.copy_(me._next("exp", t.shape))

        def dropout(input, p=0.5, training=True, inplace=False):
            if not training:
                return input
            return input * me._next("dropout", input.shape, p) * (1.0 / (1.0 - p))

        def rand(*size, generator=None, **kw):
            if generator is not None:
                return me._rand(*size, generator=generator, **kw)
            shape = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
            return me._next("rand", tuple(shape))

        torch.Tensor.exponential_, F.dropout, torch.rand = exponential_, dropout, rand
        return self

    def __exit__(self, *exc):
        torch.Tensor.exponential_, F.dropout, torch.rand = self._exp, self._drop, self._rand
        if exc[0] is None:
            assert self.pos == len(self.script), f"reference made {self.pos} draws, script has {len(self.script)}"
        return False


def wts(seed, i, shape):
    """Deterministic cotangent for a backward pass (tests rebuild it with helpers.ct_w)."""
    return H.ct_w(seed, i, shape)


def build_ct_layer(mod, A, seed, **kw):
    ct = mod.CausalTransition(64, A, **kw)
    specs = [s for s in filler.specs_of(ct) if not s[0].startswith("graph_transitioner")]
    assert specs == H.ct_layer_specs(A), "helpers.ct_layer_specs drifted from the reference's state_dict"
    missing = ct.load_state_dict(filler.fill_state(H.ct_layer_specs(A), seed + 3), strict=False)
    assert set(missing.missing_keys) == {"pos_encoding.pe"}, missing
    ct.graph_transitioner = RefTransitioner(H.GNNDouble(64, A + 1, seed + 5))
    return ct.train()


def grads_of(module, out, prefix, full=()):
    named = dict(module.named_parameters())
    for k, p in named.items():
        out[f"{prefix}.gradcks.{k}"] = cks(p.grad if p.grad is not None else torch.zeros_like(p))
    for k in full:
        p = named[k]
        out[f"{prefix}.grad.{k}"] = np32(p.grad if p.grad is not None else torch.zeros_like(p))
    module.zero_grad(set_to_none=True)


def gen_parts(mod, A, B, seed):
    """Method-level fixtures of CausalTransition (ct_mcq_vae.py:117-333)."""
    ct = build_ct_layer(mod, A, seed)
    S = D = 64
    _, one_hot = H.ct_codes(seed, B, S, D)
    action = H.ct_actions(B, A)
    out = {"seed": np.int64(seed), "B": np.int64(B), "A": np.int64(A)}
    used = sorted(set((3 * torch.arange(B) % A).tolist()))
    disc_full = ["graph_discovers.0.2.weight", "graph_discovers.0.0.bias", f"graph_discovers.{1 + used[1]}.2.weight",
                 f"graph_discovers.{1 + used[-1]}.0.bias"]

    # PositionalEncoding + _compute_mask (:117-127)
    with RefNoise(seed, [("dropout", "mask_dropout"), ("exp", "mask_gumbel")]):
        mask = ct._compute_mask(one_hot, action)
    (mask * wts(seed, 0, mask.shape)).sum().backward()
    out["mask"] = np32(mask)
    grads_of(ct, out, "mask", full=["mask.0.weight", "mask.0.bias"])
    with RefNoise(seed, [("dropout", "pos_dropout")]):
        pos = ct.pos_encoding(one_hot)
    out["pos"] = np32(pos)
    with RefNoise(seed, []):
        ct.eval()
        out["pos_eval"] = np32(ct.pos_encoding(one_hot))
        ct.train()

    # _compute_adj__comp_optim (:140-154)
    pos_l = pos.detach().clone().requires_grad_(True)
    mask_c = mask.detach()
    with RefNoise(seed, []):
        adj = ct._compute_adj__comp_optim(pos_l, action, mask_c)
    (adj * wts(seed, 1, adj.shape)).sum().backward()
    out["adj"] = np32(adj)
    out["adj.g_pos"] = np32(pos_l.grad)
    grads_of(ct, out, "adj", full=disc_full)

    # _sample_bernoulli (:180-183)
    adj_l = adj.detach().clone().requires_grad_(True)
    with RefNoise(seed, [("exp", "adj_gumbel")]):
        graph = ct._sample_bernoulli(adj_l)
    (graph * wts(seed, 2, graph.shape)).sum().backward()
    out["graph"] = np32(graph).astype(np.uint8)
    out["graph.g_adj"] = np32(adj_l.grad)

    # _compute_y (:188-228) around the test double
    pos_l = pos.detach().clone().requires_grad_(True)
    w_l = (adj * graph).detach().clone().requires_grad_(True)
    with RefNoise(seed, []):
        y = ct._compute_y(pos_l, action, w_l, mask_c)
    (y * wts(seed, 3, y.shape)).sum().backward()
    out["y"] = np32(y)
    out["y.g_pos"] = np32(pos_l.grad)
    out["y.g_adj"] = np32(w_l.grad)
    grads_of(ct, out, "y", full=["a_dense.weight", "a_dense.bias"])

    # regularisers (:314-323), latent_loss (:299-311), accuracies (:325-333)
    adj_l = adj.detach().clone().requires_grad_(True)
    with RefNoise(seed, [("rand", "kl_target")]):
        kl = ct.adjacency_KL_loss(adj_l)
    kl.backward()
    out["kl"], out["kl.g_adj"] = np.float64(kl.item()), np32(adj_l.grad)
    graph_l = graph.detach().clone().requires_grad_(True)
    gs = ct.graph_size_loss(graph_l)
    gs.backward()
    out["gsize"], out["gsize.g_graph"] = np.float64(gs.item()), np32(graph_l.grad)
    adj_l = adj.detach().clone().requires_grad_(True)
    pt = ct.positive_trial_loss(adj_l)
    pt.backward()
    out["ptrial"], out["ptrial.g_adj"] = np.float64(pt.item()), np32(adj_l.grad)
    # a second, sharper case for positive_trial_loss: rows with exact zeros of (1 - adj) (the product's prod backward
    # must be exact there too)
    adj_z = (torch.rand(B, S, S, generator=torch.Generator().manual_seed(seed + 9)) * 0.08)
    adj_z[:, ::7, 3] = 1.0
    adj_z.requires_grad_(True)
    ptz = ct.positive_trial_loss(adj_z)
    ptz.backward()
    out["ptrial_z"], out["ptrial_z.g_adj"] = np.float64(ptz.item()), np32(adj_z.grad)
    probs = y.detach().permute(0, 2, 1).reshape(B, D, 8, 8).clone().requires_grad_(True)
    _, tgt_oh = H.ct_codes(seed + 1, B, S, D)
    tgt = tgt_oh.permute(0, 2, 1).reshape(B, D, 8, 8)
    ll = ct.latent_loss(probs, tgt)
    ll.backward()
    out["latent_loss"], out["latent_loss.g"] = np.float64(ll.item()), np32(probs.grad)
    pa = torch.rand(16, A, generator=torch.Generator().manual_seed(seed + 11)).softmax(-1)
    act16 = H.ct_actions(16, A, shift=1)
    act16[::3] = F.one_hot(pa[::3].argmax(-1), A).float()          # some hits
    act16[1::3] = F.one_hot((pa[1::3].argmax(-1) + A // 2) % A, A).float()   # right variation, wrong direction
    out["acc"] = np.float64(ct.causal_accuracy(pa, act16).item())
    out["acc_nodir"] = np.float64(ct.causal_undirected_accuracy(pa, act16).item())

    # the three entry points of the layer (:231-295)
    lat4 = one_hot.permute(0, 2, 1).reshape(B, D, 8, 8)
    with RefNoise(seed, SCRIPT_BASE):
        ly, reg, met = ct(lat4)
    ((ly * wts(seed, 4, ly.shape)).sum() + reg).backward()
    out["fwd.latent_y"], out["fwd.ct_reg"], out["fwd.ct_adjacency"] = np32(ly), np.float64(reg.item()), np32(met["ct_adjacency"])
    grads_of(ct, out, "fwd", full=["graph_discovers.0.2.weight", "a_dense.weight"])
    with RefNoise(seed, SCRIPT_ACTION):
        ly, reg, met = ct.forward_action(lat4, action)
    ((ly * wts(seed, 5, ly.shape)).sum() + reg).backward()
    out["act.latent_y"], out["act.ct_reg"] = np32(ly), np.float64(reg.item())
    out["act.ct_adjacency"], out["act.ct_mask"] = np32(met["ct_adjacency"]), np32(met["ct_mask"])
    grads_of(ct, out, "act", full=disc_full + ["mask.0.bias", "a_dense.weight"])
    _, y_oh = H.ct_codes(seed + 2, B, S, D)
    with RefNoise(seed, SCRIPT_ACTION * A):
        probs_a, zero, met = ct.forward_transition(lat4, y_oh.permute(0, 2, 1).reshape(B, D, 8, 8))
    (probs_a * wts(seed, 6, probs_a.shape)).sum().backward()
    assert float(zero) == 0.0 and met == {}
    out["trans.probs"] = np32(probs_a)
    grads_of(ct, out, "trans", full=["mask.0.bias"])
    np.savez_compressed(os.path.join(OUT, f"ct_parts_a{A}.npz"), **out)
    print(f"ct_parts A={A} B={B}: mask mean {float(mask.mean()):.3f} adj mean {float(adj.mean()):.3f} "
          f"graph mean {float(graph.mean()):.3f} kl {kl.item():.5f} gsize {gs.item():.4f} ptrial {pt.item():.3e}")


def ct_yaml_kwargs(A):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ct_mcq_vae.yaml")))["model_params"]
    cfg["action_dim"] = A
    cfg["hidden_dims"] = list(cfg["hidden_dims"])
    return cfg


def gen_model(mod, A, B, seed):
    """CTMCQVAE (configs/ct_mcq_vae.yaml; action_dim overridden for the TCelebA-shaped case): the three modes,
    loss_function, ct_preprocess / ct_postprocess (ct_mcq_vae.py:472-620)."""
    torch.manual_seed(0)
    cfg = ct_yaml_kwargs(A)
    model = mod.CTMCQVAE(**cfg)
    conv = filler.fill_state(H.mcq_specs(CT_CONV_CFG), seed + 1)
    ctl = filler.fill_state(H.ct_layer_specs(A), seed + 3)
    r = model.load_state_dict({**conv, **{"ct_layer." + k: v for k, v in ctl.items()}}, strict=False)
    assert set(r.missing_keys) == {"ct_layer.pos_encoding.pe"} and not r.unexpected_keys, r
    model.ct_layer.graph_transitioner = RefTransitioner(H.GNNDouble(64, A + 1, seed + 5))
    model.train()
    x, y, _ = filler.synthetic_pairs(seed, B, A)
    action = H.ct_actions(B, A)
    out = {"seed": np.int64(seed), "B": np.int64(B), "A": np.int64(A), "gamma": np.float64(cfg["gamma"])}
    full = ["encoder.0.0.bias", "decoder.10.0.bias", "vq_layer.quantizers.0.embedding.weight", "ct_layer.mask.0.bias",
            "ct_layer.a_dense.weight", "ct_layer.graph_discovers.0.2.weight"]
    scripts = {"base": SCRIPT_BASE, "action": SCRIPT_ACTION, "causal": SCRIPT_ACTION * A}
    for mode in ("base", "action", "causal"):
        kw = {"mode": [mode] * B}
        if mode != "base":
            kw.update(input_y=y, action=action)
        with RefNoise(seed, scripts[mode]):
            res = model(x, **kw)
        losses = model.loss_function(*res)
        losses["loss"].backward()
        p = mode
        if mode == "causal":
            out[f"{p}.probs"] = np32(res[0])
        else:
            out[f"{p}.recons_strided"] = np32(res[0][:, :, ::4, ::4])
            out[f"{p}.recons_cks"] = cks(res[0])
        for k, v in losses.items():
            if k == "mode":
                assert v == mode
            elif v.dim() == 0:
                out[f"{p}.loss.{k}"] = np.float64(v.item())
            else:
                out[f"{p}.metric.{k}"] = np32(v)
        grads_of(model, out, p, full=full)
    # ct_preprocess / ct_postprocess at codebooks 1 and 4 (:472-496)
    for K in (1, 4):
        m = model if K == 1 else mod.CTMCQVAE(**{**ct_yaml_kwargs(A), "codebooks": 4})
        g = torch.Generator().manual_seed(seed + 20 + K)
        inds = torch.randint(0, 64, (B, K, 8, 8), generator=g)
        shape = (B, 128, 8, 8)
        pre = m.ct_preprocess(inds, shape)
        assert tuple(pre.shape) == (B, 64, K * 8, 8)
        out[f"pre{K}.where"] = np32(pre.reshape(B, 64, -1).argmax(1)).astype(np.int16)      # hot class per (row, col)
        out[f"pre{K}.cks"] = cks(pre * torch.arange(1, pre.numel() + 1, dtype=torch.float32).view(pre.shape) / pre.numel())
        pr = torch.rand(B, 64, K * 8, 8, generator=g)
        post = m.ct_postprocess(pr, shape)
        out[f"post{K}"] = np32(post).astype(np.int16)
        assert torch.equal(m.ct_postprocess(pre, shape), inds)
    # loss_function on hand-made inputs: MSE branch with non-zero vq, CE branch (:594-620)
    g = torch.Generator().manual_seed(seed + 30)
    rec, inp = torch.rand(B, 3, 16, 16, generator=g), torch.rand(B, 3, 16, 16, generator=g)
    l = model.loss_function(rec, inp, torch.tensor(0.37), torch.tensor(1.9), {"mode": "action", "extra": torch.tensor(5.0)})
    out["lossfn.mse"] = np.array([l["loss"].item(), l["Reconstruction_Loss"].item(), l["VQ_Loss"].item(), l["CT_Loss"].item(),
                                  l["extra"].item()])
    pr = torch.rand(B, A, generator=g).softmax(-1)
    pr[0, 0] = 0.0                                                     # exercises clamp(min=1e-4)
    l = model.loss_function(pr, action, torch.tensor(0.0), torch.tensor(0.25), {"mode": "causal"})
    out["lossfn.ce"] = np.array([l["loss"].item(), l["Reconstruction_Loss"].item()])
    l = model.loss_function(rec, inp, torch.tensor(0.1), torch.tensor(0.2))                  # 4 args: no metrics dict
    out["lossfn.nometrics"] = np.array([l["loss"].item()])
    assert set(l) == {"loss", "Reconstruction_Loss", "VQ_Loss", "CT_Loss"}
    np.savez_compressed(os.path.join(OUT, f"ct_model_a{A}.npz"), **out)
    print(f"ct_model A={A} B={B}:", {k: round(float(out[k]), 6) for k in out if ".loss.loss" in k or ".loss.CT_Loss" in k})


if __name__ == "__main__":
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    mod = load_reference_ct()
    gen_parts(mod, 12, 4, 1250)          # configs/ct_mcq_vae.yaml: action_dim 12 (TShapes3D), manual_seed 1250
    gen_parts(mod, 20, 4, 1251)          # TCelebA-shaped: 10 variations -> 20 actions, 21 heads / discoverers
    gen_model(mod, 12, 4, 1250)
    gen_model(mod, 20, 4, 1251)
