"""The checks of an implementation of ``CausalTransition`` / ``CTMCQVAE`` against the fixtures captured from the reference's own
``models/ct_mcq_vae.py`` (oracle/gen_ct_golden.py).  One body, two users: the CPU oracle (tests/test_ct_oracle_golden.py,
through the ``OracleCT`` adapter) and the HIP product (tests/test_ct_parity_gpu.py, the module itself).  An implementation
exposes the reference's method names (ct_mcq_vae.py:117-333).

GATv2Conv / dense_to_sparse are NOT covered (torch_geometric absent -> "parity unpinned"): ``graph_transitioner`` is
``helpers.GNNDouble`` on both sides of every comparison here.
"""
import numpy as np
import torch

from tests import helpers as H

TOL = 1e-4


def close(got, want, what, atol=TOL, rtol=1e-3, scale=None, outliers=0.0):
    """outliers > 0: that fraction of the elements may miss the tolerance by up to 5 % of the tensor's largest magnitude.
    Used only for gradients of a discoverer's first Linear: a pre-activation within rounding of 0 takes either LeakyReLU
    slope depending on the summation order (the product evaluates W1[x_i;x_j] as U_i + V_j), which moves the one hidden
    unit's gradient by 99 % of a single pair's contribution."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    s = float(np.abs(want).max()) if scale is None else scale
    a = atol * max(1.0, s) if scale is None else atol * s
    if outliers > 0.0:
        bad = np.abs(got - want) > a + rtol * np.abs(want)
        assert bad.mean() <= outliers, f"{what}: {bad.sum()} of {bad.size} elements off"
        assert float(np.abs(got - want).max()) <= 0.05 * max(float(np.abs(want).max()), 1e-30), what
        return
    np.testing.assert_allclose(got, want, atol=a, rtol=rtol, err_msg=what)


def rel(got, want, what, tol=1e-3):
    """max |got - want| <= tol * max |want| (gradients of tiny losses)."""
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    m = float(np.abs(want).max())
    assert float(np.abs(got - want).max()) <= tol * max(m, 1e-30), f"{what}: {np.abs(got - want).max()} vs scale {m}"


def check_grads(g, prefix, grads, rtol=2e-3):
    """grads: name -> tensor or None, names as in the reference's named_parameters() (graph_transitioner.fn.* = the double)."""
    seen = 0
    for key in g:
        if key.startswith(prefix + ".gradcks."):
            name = key[len(prefix) + 9:]
            got = grads(name)
            want = g[key]
            if got is None:
                assert want[1] == 0.0, f"{prefix}: {name} has no gradient here but the reference's is non-zero"
                continue
            H.assert_cks_close(H.cks(got), want, rtol=rtol, atol=1e-5, what=f"{prefix}:{name}")
            seen += 1
        elif key.startswith(prefix + ".grad."):
            name = key[len(prefix) + 6:]
            got = grads(name)
            want = g[key]
            if got is None:
                assert not np.any(want), name
                continue
            first_linear = "graph_discovers." in name and (name.endswith(".0.bias") or name.endswith(".0.weight"))
            close(got, want, f"{prefix}:{name}", rtol=rtol, outliers=0.005 if first_linear else 0.0)
    assert seen > 0, prefix


def check_parts(g, ct, ns, dev, grads, zero_grad):
    """Method-level parity of CausalTransition (fixture ct_parts_a*.npz)."""
    seed, B, A = int(g["seed"]), int(g["B"]), int(g["A"])
    S = D = 64
    _, one_hot = H.ct_codes(seed, B, S, D)
    one_hot = one_hot.to(dev)
    action = H.ct_actions(B, A).to(dev)
    w = lambda i, shape: H.ct_w(seed, i, shape).to(dev)       # noqa: E731

    # PositionalEncoding, _compute_mask (ct_mcq_vae.py:14-38,117-127)
    zero_grad()
    ns.reset()
    mask = ct._compute_mask(one_hot, action)
    assert tuple(mask.shape) == (B, S, 1)
    np.testing.assert_array_equal(mask.detach().cpu().numpy(), g["mask"], err_msg="intervention mask (exact)")
    (mask * w(0, mask.shape)).sum().backward()
    check_grads(g, "mask", grads)
    ns.reset()
    pos = ct.pos_encoding(one_hot)
    close(pos, g["pos"], "pos_encoding (train)", atol=1e-6)

    # _compute_adj (:140-154): >= 3 distinct actions in the batch
    zero_grad()
    ns.reset()
    pos_l = torch.from_numpy(g["pos"]).to(dev).requires_grad_(True)
    mask_c = torch.from_numpy(g["mask"]).to(dev)
    adj = ct._compute_adj(pos_l, action, mask_c)
    close(adj, g["adj"], "adjacency coefficients", atol=2e-6)
    (adj * w(1, adj.shape)).sum().backward()
    close(pos_l.grad, g["adj.g_pos"], "d adj / d pos")
    check_grads(g, "adj", grads)

    # _sample_bernoulli (:180-183)
    ns.reset()
    adj_l = torch.from_numpy(g["adj"]).to(dev).requires_grad_(True)
    graph = ct._sample_bernoulli(adj_l)
    np.testing.assert_array_equal(graph.detach().cpu().numpy().astype(np.uint8), g["graph"], err_msg="sampled graph (exact)")
    (graph * w(2, graph.shape)).sum().backward()
    close(adj_l.grad, g["graph.g_adj"], "straight-through gradient")

    # _compute_y (:188-228) around the double
    zero_grad()
    ns.reset()
    pos_l = torch.from_numpy(g["pos"]).to(dev).requires_grad_(True)
    w_l = (torch.from_numpy(g["adj"]) * torch.from_numpy(g["graph"]).float()).to(dev).requires_grad_(True)
    y = ct._compute_y(pos_l, action, w_l, mask_c)
    close(y, g["y"], "_compute_y", atol=2e-6)
    (y * w(3, y.shape)).sum().backward()
    close(pos_l.grad, g["y.g_pos"], "d y / d pos")
    close(w_l.grad, g["y.g_adj"], "d y / d adjacency")
    check_grads(g, "y", grads)

    # regularisers, latent_loss, accuracies (:299-333)
    ns.reset()
    adj_l = torch.from_numpy(g["adj"]).to(dev).requires_grad_(True)
    kl = ct.adjacency_KL_loss(adj_l)
    assert abs(kl.item() - float(g["kl"])) <= 1e-6 + 1e-4 * abs(float(g["kl"]))
    kl.backward()
    rel(adj_l.grad, g["kl.g_adj"], "d KL / d adj")
    graph_l = torch.from_numpy(g["graph"]).float().to(dev).requires_grad_(True)
    gs = ct.graph_size_loss(graph_l)
    assert abs(gs.item() - float(g["gsize"])) <= 1e-4 * float(g["gsize"])
    gs.backward()
    rel(graph_l.grad, g["gsize.g_graph"], "d |graph|_F / d graph")
    adj_l = torch.from_numpy(g["adj"]).to(dev).requires_grad_(True)
    pt = ct.positive_trial_loss(adj_l)
    assert abs(pt.item() - float(g["ptrial"])) <= 1e-3 * float(g["ptrial"])
    pt.backward()
    rel(adj_l.grad, g["ptrial.g_adj"], "d positive_trial / d adj", tol=2e-3)
    adj_z = (torch.rand(B, S, S, generator=torch.Generator().manual_seed(seed + 9)) * 0.08)
    adj_z[:, ::7, 3] = 1.0
    adj_z = adj_z.to(dev).requires_grad_(True)
    ptz = ct.positive_trial_loss(adj_z)
    assert abs(ptz.item() - float(g["ptrial_z"])) <= 1e-4 * float(g["ptrial_z"])
    ptz.backward()
    rel(adj_z.grad, g["ptrial_z.g_adj"], "d positive_trial / d adj with exact zeros", tol=2e-3)
    probs = torch.from_numpy(g["y"]).permute(0, 2, 1).reshape(B, D, 8, 8).contiguous().to(dev).requires_grad_(True)
    _, tgt_oh = H.ct_codes(seed + 1, B, S, D)
    tgt = tgt_oh.permute(0, 2, 1).reshape(B, D, 8, 8).to(dev)
    ll = ct.latent_loss(probs, tgt)
    assert abs(ll.item() - float(g["latent_loss"])) <= TOL
    ll.backward()
    close(probs.grad, g["latent_loss.g"], "d latent_loss / d probs")
    pa = torch.rand(16, A, generator=torch.Generator().manual_seed(seed + 11)).softmax(-1)
    act16 = H.ct_actions(16, A, shift=1)
    act16[::3] = torch.nn.functional.one_hot(pa[::3].argmax(-1), A).float()
    act16[1::3] = torch.nn.functional.one_hot((pa[1::3].argmax(-1) + A // 2) % A, A).float()
    assert abs(float(ct.causal_accuracy(pa.to(dev), act16.to(dev))) - float(g["acc"])) < 1e-6
    assert abs(float(ct.causal_undirected_accuracy(pa.to(dev), act16.to(dev))) - float(g["acc_nodir"])) < 1e-6
    assert float(g["acc"]) < float(g["acc_nodir"]) < 1.0

    # forward / forward_action / forward_transition (:231-295)
    lat4 = one_hot.permute(0, 2, 1).reshape(B, D, 8, 8)
    zero_grad()
    ns.reset()
    ly, reg, met = ct(lat4)
    close(ly, g["fwd.latent_y"], "forward: latent_y", atol=2e-6)
    assert abs(reg.item() - float(g["fwd.ct_reg"])) <= TOL
    close(met["ct_adjacency"], g["fwd.ct_adjacency"], "forward: ct_adjacency", atol=2e-6)
    ((ly * w(4, ly.shape)).sum() + reg).backward()
    check_grads(g, "fwd", grads)
    zero_grad()
    ns.reset()
    ly, reg, met = ct.forward_action(lat4, action)
    close(ly, g["act.latent_y"], "forward_action: latent_y", atol=2e-6)
    assert abs(reg.item() - float(g["act.ct_reg"])) <= TOL * max(1.0, abs(float(g["act.ct_reg"])))
    close(met["ct_adjacency"], g["act.ct_adjacency"], "forward_action: ct_adjacency", atol=2e-6)
    close(met["ct_mask"], g["act.ct_mask"], "forward_action: ct_mask", atol=1e-6)
    ((ly * w(5, ly.shape)).sum() + reg).backward()
    check_grads(g, "act", grads)
    zero_grad()
    ns.reset()
    _, y_oh = H.ct_codes(seed + 2, B, S, D)
    probs_a, zero, met = ct.forward_transition(lat4, y_oh.permute(0, 2, 1).reshape(B, D, 8, 8).to(dev))
    assert float(zero) == 0.0 and met == {}
    close(probs_a, g["trans.probs"], "forward_transition: action probabilities", atol=1e-5)
    (probs_a * w(6, probs_a.shape)).sum().backward()
    check_grads(g, "trans", grads)


def check_model_mode(g, mode, res, losses, grads):
    """One mode of CTMCQVAE.forward + loss_function + backward against ct_model_a*.npz."""
    p = mode
    assert len(res) == 5 and res[4]["mode"] == mode and losses["mode"] == mode
    if mode == "causal":
        close(res[0], g[f"{p}.probs"], "action probabilities", atol=1e-5)
    else:
        close(res[0][:, :, ::4, ::4], g[f"{p}.recons_strided"], "reconstruction")
        H.assert_cks_close(H.cks(res[0]), g[f"{p}.recons_cks"], rtol=1e-4, atol=1e-4, what="recons checksum")
    want_keys = {k.split(".", 2)[2] for k in g if k.startswith(f"{p}.loss.") or k.startswith(f"{p}.metric.")}
    assert set(losses) == want_keys | {"mode"}, (sorted(losses), sorted(want_keys))
    for k in want_keys:
        if f"{p}.loss.{k}" in g:
            want = float(g[f"{p}.loss.{k}"])
            assert abs(float(losses[k]) - want) <= TOL * max(1.0, abs(want)), (mode, k, float(losses[k]), want)
        else:
            close(losses[k], g[f"{p}.metric.{k}"], f"{mode}: {k}", atol=2e-6)
    check_grads(g, p, grads)


class OracleCT:
    """The CPU oracle (oracle/causal_cpu.py) behind the reference's method names, for check_parts."""

    def __init__(self, sd, double, ns):
        from oracle import causal_cpu as C
        self.C, self.ns, self.double = C, ns, double
        self.sd = {k: (v.detach().clone().requires_grad_(True) if k != "pos_encoding.pe" else v) for k, v in sd.items()}
        self.gnn = lambda nodes, adj: double(nodes, adj)

    def grads(self, name):
        if name.startswith("graph_transitioner.fn."):
            return getattr(self.double, name.rsplit(".", 1)[1]).grad
        return self.sd[name].grad

    def zero_grad(self):
        for v in list(self.sd.values()) + list(self.double.parameters()):
            v.grad = None

    def pos_encoding(self, x):
        return self.C.pos_encoding(x, self.ns, "pos_dropout")

    def _compute_mask(self, one_hot, action):
        return self.C.compute_mask(self.sd, one_hot, action, self.ns)

    def _compute_adj(self, latent, action, mask):
        return self.C.compute_adj(self.sd, latent, action, mask)

    def _sample_bernoulli(self, adj):
        return self.C.bernoulli_st(adj, self.ns.draw("adj_gumbel", tuple(adj.shape) + (2,)))

    def _compute_y(self, latent, action, adjacency, mask):
        return self.C.compute_y(self.sd, latent, action, adjacency, mask, self.gnn, self.ns)

    def adjacency_KL_loss(self, adj):
        return self.C.adjacency_kl_loss(adj, self.ns)

    def graph_size_loss(self, graph):
        return self.C.graph_size_loss(graph)

    def positive_trial_loss(self, adj):
        return self.C.positive_trial_loss(adj)

    def latent_loss(self, latent, latent_y):
        return self.C.latent_loss(latent, latent_y)

    def causal_accuracy(self, p, a):
        return self.C.causal_accuracy(p, a)

    def causal_undirected_accuracy(self, p, a):
        return self.C.causal_undirected_accuracy(p, a)

    def __call__(self, latent):
        return self.C.ct_forward(self.sd, latent, self.ns, self.gnn)

    def forward_action(self, latent, action):
        return self.C.ct_forward_action(self.sd, latent, action, self.ns, self.gnn)

    def forward_transition(self, latent, latent_y):
        return self.C.ct_forward_transition(self.sd, latent, latent_y, self.ns, self.gnn)
