"""CausalTransition (reference: models/ct_mcq_vae.py:14-333) — SURVEY.md §8(f) "next #1".

Status: every method is pinned against the reference's own class (tests/golden/ct_parts_a{12,20}.npz) except
the arithmetic inside the two ``GATv2Conv`` layers / ``dense_to_sparse``: they live in torch-geometric 2.2.0, which is absent
here, so they follow the published GATv2 algorithm and **that part's parity is unpinned** (no reference fixture covers it).
At the shapes of ``ct_mcq_vae.yaml`` and the published runs (64 latent nodes, head width <= 128, grouped-linear dims % 4) the
whole layer runs as HIP kernels -- grouped MFMA Linear layers, the pair scorer, the fused dense GATv2 layer, the regulariser /
blend-softmax / cross-entropy / mask / sampler kernels (csrc/{glinear,pairmlp,gatlayer,gat,ctmisc}.hip); the
torch device branches below serve any other ``latent_dims`` / node count and a replacement ``graph_transitioner`` module.

Design notes (MI355X-first rather than a PyG translation):
* graphs are B disjoint dense graphs of 64(+1 action [+1 noise]) nodes, so GATv2 runs as *dense batched*
  masked attention over ``[B, N, N]`` instead of scatter/gather over an edge list;
* the pair-MLP ``Linear(2D, 800)`` over all 64x64 node pairs is evaluated as ``U_i + V_j`` with two
  ``[B*64, D] x [D, 800]`` GEMMs (the concatenation in ct_mcq_vae.py:141-147 is linear in its halves),
  which removes the 0.85 GMAC/sample pair GEMM the reference materialises;
* ``state_dict`` keys mirror the reference (``graph_transitioner.module_{0,2}.*`` as PyG names them).
"""
import math
from typing import List

import torch
from torch import nn
from torch.nn import functional as F

from .. import kernels as K
from .types_ import Tensor


class DeviceNoise:
    """Default noise source: every stochastic op of the layer draws on the device from torch's generator.
    ``draw(tag, shape, p)`` returns, by the tag's suffix: ``*_dropout`` a keep mask (1.0 with probability 1-p),
    ``*_gumbel`` standard exponential draws E (the Gumbel noise is -log E, as F.gumbel_softmax forms it), ``*_noise``
    N(0,1), otherwise U[0,1).  Tests install a source with the same interface that replays CPU-generated draws
    (SURVEY N1): ``set_noise_source``."""

    def __init__(self, device=None):
        self.device = device

    def draw(self, tag, shape, p=0.0, device=None):
        dev = device if device is not None else self.device
        if tag.endswith("_dropout"):
            return torch.empty(shape, device=dev, dtype=torch.float32).bernoulli_(1.0 - p)      # one launch (was rand, >=, cast)
        if tag.endswith("_gumbel"):
            return torch.empty(shape, device=dev, dtype=torch.float32).exponential_()
        if tag.endswith("_noise"):
            return torch.randn(shape, device=dev)
        return torch.rand(shape, device=dev)


_noise = DeviceNoise()


def set_noise_source(src):
    """Install a noise source (None: the device default); returns the previous one."""
    global _noise
    prev, _noise = _noise, (src if src is not None else DeviceNoise())
    return prev


def _draw(tag, shape, p=0.0, device=None):
    t = _noise.draw(tag, tuple(shape), p) if not isinstance(_noise, DeviceNoise) else _noise.draw(tag, tuple(shape), p, device)
    return t if device is None or t.device == torch.device(device) else t.to(device)


class PositionalEncoding(nn.Module):
    """Sinusoidal table + Dropout(0.1) (ct_mcq_vae.py:14-38)."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 4096):
        super().__init__()
        self.p = dropout
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe)

    def forward(self, x: Tensor, tag: str = "pos_dropout") -> Tensor:       # x [B, S, D]
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and (x.size(1) * x.size(2)) % 4 == 0 and self.pe.device == x.device:
            drop = self.training and self.p > 0.0                            # one launch (csrc/ctmisc.hip)
            keep = _draw(tag, x.shape, self.p, x.device) if drop else None
            return K.PosEncode.apply(x, self.pe[:x.size(1), 0], keep, 1.0 / (1.0 - self.p) if drop else 1.0)
        y = x + self.pe[:x.size(1), 0].to(x.device).unsqueeze(0)
        if not self.training or self.p == 0.0:
            return y
        return y * _draw(tag, y.shape, self.p, x.device) * (1.0 / (1.0 - self.p))


class DenseGATv2(nn.Module):
    """GATv2Conv(in, out, heads, edge_dim=1, concat=True, add_self_loops=True(fill 'mean'), negative_slope=0.2) on a batch
    of dense weighted graphs.  adj[b, r, c] != 0 is an edge r -> c carrying attribute adj[b, r, c].  PARITY UNPINNED
    (torch_geometric is absent): written from the published GATv2 algorithm, cross-checked against the oracle's edge-list
    restatement."""

    def __init__(self, in_channels, out_channels, heads, negative_slope=0.2):
        super().__init__()
        self.heads, self.out_channels, self.negative_slope = heads, out_channels, negative_slope
        self.lin_l = nn.Linear(in_channels, heads * out_channels)
        self.lin_r = nn.Linear(in_channels, heads * out_channels)
        self.lin_edge = nn.Linear(1, heads * out_channels, bias=False)
        self.att = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        for w in (self.lin_l.weight, self.lin_r.weight, self.lin_edge.weight, self.att):
            nn.init.xavier_uniform_(w)
        nn.init.zeros_(self.lin_l.bias)
        nn.init.zeros_(self.lin_r.bias)

    def fused_ok(self, x):
        H, C = self.heads, self.out_channels
        return (x.is_cuda and x.size(1) == 64 and 16 <= C <= 128 and C % 4 == 0 and self.lin_l.in_features % 4 == 0
                and self.lin_l.weight.is_contiguous() and self.lin_r.weight.is_contiguous() and H <= 64)

    def forward_fused(self, x, adj, slots=None, act=K.ACT_NONE):
        """HIP path (csrc/gatlayer.hip): x [B,64,Cin], adj [B,64,64]; slots: None = every head, or a list with one entry per
        head slot -- None (head 0 for every sample) or an int32 [B] tensor naming each sample's head -> [B,64,Hs*C];
        act: the activation behind the layer, applied in the kernel's epilogue."""
        B = x.size(0)
        H, C = self.heads, self.out_channels
        Kin = x.size(-1)
        wl, wr = self.lin_l.weight, self.lin_r.weight
        heads_sel = None
        if slots is None:               # every head: lin_l | lin_r as two segments of one grouped GEMM (group 0 for everybody)
            Hs = H
            xlr = K.GroupLinear.apply(x, Kin, H * C, ((0, None), (0, None)), wl.view(1, H * C, Kin), self.lin_l.bias.view(1, H * C),
                                      wr.view(1, H * C, Kin), self.lin_r.bias.view(1, H * C))
        else:                           # head slots: each slot reads the C rows of its head inside lin_l / lin_r
            Hs = len(slots)
            cols = list(slots)
            zero = torch.zeros(B, dtype=torch.int32, device=x.device) if any(c is None for c in cols) else None
            heads_sel = torch.stack([zero if c is None else c for c in cols], dim=1)
            wb = []
            for w, b in ((wl, self.lin_l.bias), (wr, self.lin_r.bias)):
                for _ in range(Hs):
                    wb += [w.view(H, C, Kin), b.view(H, C)]
            xlr = K.GroupLinear.apply(x, Kin, C, tuple((0, g) for g in cols + cols), *wb)
        return K.GATLayer.apply(xlr, adj, self.lin_edge.weight.view(H, C), self.att[0], self.bias, heads_sel, Hs, C,
                                self.negative_slope, act)

    def forward(self, x, adj):                    # x [B,N,Cin], adj [B,N,N]: every head, any N (torch device ops)
        B, N, _ = x.shape
        H, C = self.heads, self.out_channels
        xl = self.lin_l(x).view(B, N, H, C)       # source side
        xr = self.lin_r(x).view(B, N, H, C)       # target side
        eye = torch.eye(N, dtype=torch.bool, device=x.device).unsqueeze(0)
        edge = (adj != 0) & ~eye                  # existing self loops are removed first
        w = adj * edge
        deg = edge.sum(1).clamp(min=1)            # incoming edges per target
        attr = w + torch.diag_embed(w.sum(1) / deg)   # self loops carry the mean incoming attribute
        keep = edge | eye
        we = self.lin_edge.weight.view(H, C)
        outs = []
        for h in range(H):                        # head by head keeps the [B,N,N,C] score tensor small
            m = xl[:, :, None, h, :] + xr[:, None, :, h, :] + attr.unsqueeze(-1) * we[h]
            s = (F.leaky_relu(m, self.negative_slope) * self.att[0, h]).sum(-1)
            alpha = torch.softmax(s.masked_fill(~keep, float('-inf')), dim=1)      # over sources r of each target c
            outs.append(torch.einsum('brc,brk->bck', alpha, xl[:, :, h, :]))
        return torch.stack(outs, 2).reshape(B, N, H * C) + self.bias


class _GraphTransitioner(nn.Module):
    """gnn.Sequential([GATv2Conv, LeakyReLU]*k + [GATv2Conv]) with PyG's child names module_{i}."""

    def __init__(self, input_dim, latent_dims, heads):
        super().__init__()
        c, idx = input_dim, 0
        self.order = []
        for dim in latent_dims:
            self.add_module(f"module_{idx}", DenseGATv2(c, dim, heads))
            self.add_module(f"module_{idx + 1}", nn.LeakyReLU())
            self.order += [f"module_{idx}", f"module_{idx + 1}"]
            idx += 2
            c = dim * heads
        self.add_module(f"module_{idx}", DenseGATv2(c, input_dim, heads))
        self.order.append(f"module_{idx}")

    def gat_layers(self):
        return [self._modules[n] for n in self.order if isinstance(self._modules[n], DenseGATv2)]

    def fused_ok(self, x):
        return all(l.fused_ok(x) for l in self.gat_layers())

    def forward_fused(self, x, adj, slots):
        """The 64 latent nodes only, the last layer restricted to the head slots ``slots`` (DenseGATv2.forward_fused)
        -> [B,64,len(slots)*D]."""
        layers = self.gat_layers()
        for l in layers[:-1]:
            x = l.forward_fused(x, adj, None, K.ACT_LRELU)      # nn.LeakyReLU() behind the layer, in its epilogue
        return layers[-1].forward_fused(x, adj, slots, K.ACT_NONE)

    def forward(self, x, adj):
        for name in self.order:
            m = self._modules[name]
            x = m(x, adj) if isinstance(m, DenseGATv2) else m(x)
        return x


class _DiscoverBank(nn.ModuleList):
    """``graph_discovers``: 1 + action_dim edge scorers Sequential(Linear(2D, H), LeakyReLU, Linear(H, 1), Sigmoid)
    (ct_mcq_vae.py:86-95) under the reference's parameter names, stored as four BANKS -- all first-layer weights back to
    back, then the first-layer biases, the scorer rows, the scorer biases -- so that the kernels index the discoverer of a
    sample's action inside the bank instead of gathering per-sample copies of the weights.  FlatParamMixin lays the banks
    out (storage_blocks); a stand-alone module that is not under a flat root falls back to torch.stack."""

    autograd_grads = True        # placed by storage_blocks(), but the gradients arrive through autograd (kernels.BankView)

    def __init__(self, n, in_dim, hidden):
        super().__init__([nn.Sequential(nn.Linear(in_dim, hidden), nn.LeakyReLU(), nn.Linear(hidden, 1), nn.Sigmoid())
                          for _ in range(n)])
        self.in_dim, self.hidden = in_dim, hidden

    def _lists(self):
        return ([d[0].weight for d in self], [d[0].bias for d in self], [d[2].weight for d in self], [d[2].bias for d in self])

    def storage_blocks(self):
        blocks = []
        for plist in self._lists():
            n = plist[0].numel()
            views = [(p, k * n, tuple(p.shape), tuple(p.detach().contiguous().stride())) for k, p in enumerate(plist)]
            blocks.append((-(-len(plist) * n // 4) * 4, views))       # every bank starts 16-byte aligned
        return blocks

    def tensors(self):
        """(W1 [G,H,2D], b1 [G,H], w2 [G,H], b2 [G])"""
        G, Hd = len(self), self.hidden
        out = []
        for plist in self._lists():
            out.append(K.BankView.apply(*plist) if (plist[0].is_cuda and K.banked(plist)) else torch.stack(plist))
        return out[0], out[1], out[2].reshape(G, Hd), out[3].reshape(G)


class _ProdLastDim(torch.autograd.Function):
    """x.prod(-1) with a backward that never leaves the device: d/dx_i = g * prod_{j != i} x_j from an exclusive prefix
    and suffix product.  torch's own prod backward counts the zeros of x with .item() (a host sync, which also makes the
    step impossible to capture into a hipGraph); this form is exact with zeros too."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.prod(-1)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        one = torch.ones_like(x[..., :1])
        prefix = torch.cat([one, torch.cumprod(x, -1)[..., :-1]], -1)                               # prod_{j<i}
        suffix = torch.cat([torch.cumprod(x.flip(-1), -1)[..., :-1].flip(-1), one], -1)             # prod_{j>i}
        return g.unsqueeze(-1) * prefix * suffix


def sample_bernoulli_st(p, tag, weighted=False):
    """Straight-through Bernoulli sample via 2-class Gumbel-softmax(tau=1, hard=True) of log(clamp([1-p, p], 1e-4))
    (ct_mcq_vae.py:124-126,180-183): HIP kernel; the two exponential draws per element come from the noise source.
    weighted: also p * sample from the same launch."""
    return K.CTSample.apply(p, _draw(tag, tuple(p.shape) + (2,), device=p.device), weighted)


class CausalTransition(nn.Module):

    def __init__(self, input_dim: int, action_dim: int, latent_dims: List = None, noise: str = "off",
                 c_alpha: float = 0.7, c_beta: float = 0.4, c_delta: float = 0.4, c_epsilon: float = 0.4,
                 comp_adj_optim: str = "comp", **kwargs) -> None:
        super().__init__()
        self.input_dim, self.action_dim, self.noise = input_dim, action_dim, noise
        self.alpha, self.beta, self.delta, self.epsilon = c_alpha, c_beta, c_delta, c_epsilon
        self.a_dense = nn.Linear(action_dim, input_dim)
        self.pos_encoding = PositionalEncoding(input_dim)
        if latent_dims is None:
            latent_dims = [800, 100]
        self.latent_dims = latent_dims
        self.graph_discovers = _DiscoverBank(action_dim + 1, 2 * input_dim, latent_dims[0])
        self.mask = nn.Sequential(nn.Linear(action_dim + input_dim, input_dim), nn.Sigmoid())
        self.nb_heads = 1 + action_dim
        self.graph_transitioner = _GraphTransitioner(input_dim, latent_dims[1:], self.nb_heads)

    # ---- pieces ----------------------------------------------------------------------------------
    def _pair_coeffs(self, disc, x):
        """sigmoid(w2 . lrelu(W1 [x_i ; x_j] + b1) + b2) for every ordered pair (i, j): [B,N,D] -> [B,N,N]."""
        D = x.size(-1)
        lin1, lin2 = disc[0], disc[2]
        u = F.linear(x, lin1.weight[:, :D])                    # x_i half
        v = F.linear(x, lin1.weight[:, D:], lin1.bias)         # x_j half (+ bias)
        if x.is_cuda and x.size(1) <= 64:
            # HIP: lrelu(u_i + v_j) . w2 + b2 -> sigmoid for every pair without the [B,N,N,hidden] intermediates
            return K.PairMLP.apply(u, v, lin2.weight.view(-1), lin2.bias)
        h = F.leaky_relu(u.unsqueeze(2) + v.unsqueeze(1))      # [B,N,N,hidden]
        return torch.sigmoid(F.linear(h, lin2.weight, lin2.bias)).squeeze(-1)

    def _compute_mask(self, one_hot_latent, action):
        B, S, D = one_hot_latent.shape
        if one_hot_latent.is_cuda and S == 64 and D == 64 and not one_hot_latent.requires_grad:
            pe = self.pos_encoding                                   # one launch each way (kernels.CTMask, csrc/ctmisc.hip)
            drop = pe.training and pe.p > 0.0
            keep = _draw("mask_dropout", (B, S, D), pe.p, one_hot_latent.device) if drop else None
            expo = _draw("mask_gumbel", (B, S, 2), device=one_hot_latent.device)
            return K.CTMask.apply(one_hot_latent, action.to(torch.float32), pe.pe[:S, 0], keep, 1.0 / (1.0 - pe.p) if drop else 1.0,
                                  self.mask[0].weight, self.mask[0].bias, expo).unsqueeze(-1)
        act = action.unsqueeze(1).expand(B, S, action.size(-1)).to(torch.float32)
        pos = self.pos_encoding(torch.zeros_like(one_hot_latent), "mask_dropout")
        inter = self.mask(torch.cat([act, pos], dim=-1))
        p = (one_hot_latent * inter).sum(dim=-1)               # [B,S]
        return sample_bernoulli_st(p, "mask_gumbel").unsqueeze(-1)

    def _action_group(self, action):
        """1 + argmax(action) as int32 [B]: the discoverer / head a sample's action selects (ct_mcq_vae.py:147, 224); formed once
        per action tensor (three launches) and reused by the adjacency and the transition of the same forward."""
        c = getattr(self, "_grp_cache", None)
        if c is not None and c[0] is action and c[1] == action._version:
            return c[2]
        grp = (torch.argmax(action, dim=-1) + 1).to(torch.int32)
        self._grp_cache = (action, action._version, grp)
        return grp

    def _sample_bernoulli(self, adjacency):
        return sample_bernoulli_st(adjacency, "adj_gumbel")

    def _compute_adj(self, latent, action, mask):
        """ct_mcq_vae.py:140-154.  W1 [x_i ; x_j] = W1a x_i + W1b x_j: the projections u = x W1a^T, v = x W1b^T + b1 of
        discoverer 0 and of every sample's own discoverer are ONE grouped GEMM (kernels.GroupLinear), the all-pairs scorer
        reads them and the scorer bank in place (kernels.PairScores).  mask None: no intervention (base mode)."""
        D, Hd = latent.size(-1), self.latent_dims[0]
        if latent.is_cuda and latent.size(1) == 64 and K.glinear_ok(D, Hd, 2 * D):
            W1, b1, w2, b2 = self.graph_discovers.tensors()
            if mask is None:
                uv = K.GroupLinear.apply(latent, D, Hd, ((0, None), (D, None)), W1, None, W1, b1)
                return K.PairScores.apply(uv, w2, b2, None, Hd)[0]
            grp = self._action_group(action)
            uv = K.GroupLinear.apply(latent, D, Hd, ((0, None), (D, None), (0, grp), (D, grp)), W1, None, W1, b1, W1, None, W1, b1)
            s = K.PairScores.apply(uv, w2, b2, grp, Hd)
            return K.MaskBlend.apply(s, mask)                       # s[0] * (1 - mask) + s[1] * mask
        no_inter = self._pair_coeffs(self.graph_discovers[0], latent)
        if mask is None:
            return no_inter
        ids = torch.argmax(action, dim=-1)
        inter = torch.zeros_like(no_inter)
        for i in set(ids.tolist()):
            sel = torch.where(ids == i)[0]
            inter[sel] = self._pair_coeffs(self.graph_discovers[1 + i], latent[sel])
        return no_inter * (1 - mask) + inter * mask

    def _compute_y(self, latent, action, adjacency, mask):
        """ct_mcq_vae.py:188-228.  With the layer's own GATv2 stack on the device the fused form runs (below); any other
        ``graph_transitioner`` module (dense calling convention (nodes [B,N,D], adjacency [B,N,N]) -> [B,N,heads*D]) gets the
        reference's data flow: nodes padded with the action (and noise) node, adjacency padded with their edges."""
        gt = self.graph_transitioner
        if isinstance(gt, _GraphTransitioner) and latent.size(1) == 64 and gt.fused_ok(latent):
            return self._compute_y_fused(latent, action, adjacency, mask)
        B, S, D = latent.shape
        action_node = self.a_dense(action)
        if self.noise == "exo":
            latent = latent + _draw("exo_noise", latent.shape, device=latent.device)
            supp = action_node.unsqueeze(1)
        elif self.noise == "endo":
            supp = torch.stack([action_node, _draw("endo_noise", action_node.shape, device=action_node.device)], dim=1)
        else:
            supp = action_node.unsqueeze(1)
        ns = supp.size(1)
        nodes = torch.cat([latent, supp], 1)
        # extra nodes: edges from every latent node to them (column of ones), none leaving (row of zeros)
        adj = F.pad(F.pad(adjacency, (0, ns, 0, 0), value=1.0), (0, 0, 0, ns), value=0.0)
        y = gt(nodes, adj)[:, :S].view(B, S, self.nb_heads, D)
        base = y[:, :, 0]
        if mask is None:
            return base.softmax(dim=-1)
        head = (action.argmax(dim=-1) + 1).view(B, 1, 1, 1).expand(B, S, 1, D)
        return (base * (1 - mask) + torch.gather(y, 2, head).squeeze(2) * mask).softmax(dim=-1)

    def _compute_y_fused(self, latent, action, adjacency, mask):
        """Same result on the HIP GATv2 kernels, computing only what reaches it:
        * the appended action / noise nodes have no outgoing edge (their rows of the padded adjacency are zero) and their own
          outputs are dropped (ct_mcq_vae.py:203-221), so the graph is the 64 latent nodes (``a_dense`` gets the exact zero
          gradient it has in the reference);
        * of the last layer only head 0 and head 1 + argmax(action) are read (:224-226): two head slots per sample instead of
          13 / 21 heads (one slot when there is no intervention mask)."""
        B, S, D = latent.shape
        if self.noise == "exo":
            latent = latent + _draw("exo_noise", latent.shape, device=latent.device)
        elif self.noise == "endo":
            _draw("endo_noise", (B, D), device=latent.device)      # the draw the reference makes for its isolated noise node
        slots = [None] if mask is None else [None, self._action_group(action)]      # head 0 (, head 1 + action)
        y = self.graph_transitioner.forward_fused(latent, adjacency, slots).view(B, S, len(slots), D)
        if D <= 64:
            return K.BlendSoftmax.apply(y, mask)          # head blend + softmax in one launch (csrc/ctmisc.hip)
        if mask is None:
            return y[:, :, 0].softmax(dim=-1)
        return (y[:, :, 0] * (1 - mask) + y[:, :, 1] * mask).softmax(dim=-1)

    # ---- modes -----------------------------------------------------------------------------------
    def forward(self, latent: Tensor, **kwargs) -> List[Tensor]:
        shape = latent.shape                                   # [B,D,H,W]
        lat = latent.permute(0, 2, 3, 1).reshape(shape[0], -1, shape[1])
        pos = self.pos_encoding(lat)
        action = torch.zeros(lat.size(0), self.action_dim, device=lat.device)
        adj = self._compute_adj(pos, action, None)             # mask == 0 in base mode
        graph, weighted = sample_bernoulli_st(adj, "adj_gumbel", True)
        latent_y = self._compute_y(pos, action, weighted, None)
        ident = torch.eye(graph.size(-1), device=lat.device, dtype=graph.dtype).expand_as(graph)
        y_id = self._compute_y(pos, action, ident, None)
        ct_reg = self.alpha * (F.cross_entropy(y_id.reshape(-1, shape[1]).clamp(min=1e-4).log(),
                                               lat.reshape(-1, shape[1]).argmax(dim=-1))
                               + F.mse_loss(graph, ident))
        return [latent_y.permute(0, 2, 1).unflatten(2, tuple(shape[2:])), ct_reg, {"ct_adjacency": adj.mean(0)}]

    def forward_action(self, latent: Tensor, action: Tensor, **kwargs) -> List[Tensor]:
        shape = latent.shape
        lat = latent.permute(0, 2, 3, 1).reshape(shape[0], -1, shape[1])
        mask = self._compute_mask(lat, action)
        pos = self.pos_encoding(lat)
        adj = self._compute_adj(pos, action, mask)
        graph, weighted = sample_bernoulli_st(adj, "adj_gumbel", True)
        latent_y = self._compute_y(pos, action, weighted, mask)
        if adj.is_cuda and adj.size(-1) == 64:            # the three regularisers in one launch each way (csrc/ctmisc.hip)
            uni = _draw("kl_target", (adj.size(0), adj.size(1) * adj.size(2)), device=adj.device)
            ct_reg = K.CTActionReg.apply(adj, graph, uni, self.beta, self.delta, self.epsilon)
        else:
            ct_reg = self.beta * self.adjacency_KL_loss(adj) + self.delta * self.graph_size_loss(graph) \
                + self.epsilon * self.positive_trial_loss(adj)
        return [latent_y.permute(0, 2, 1).unflatten(2, tuple(shape[2:])), ct_reg,
                {"ct_mask": mask.view(shape[:1] + shape[2:]).mean(0), "ct_adjacency": adj.mean(0)}]

    def forward_transition(self, latent: Tensor, latent_y: Tensor, **kwargs) -> List[Tensor]:
        B, A = latent.size(0), self.action_dim
        y_inds = latent_y.permute(0, 2, 3, 1).reshape(-1, latent_y.size(1)).argmax(dim=-1)
        dist = []
        for i in range(A):
            a = F.one_hot(torch.full((B,), i, device=latent.device), A).to(latent.dtype)
            y = self.forward_action(latent, a)[0]
            y_log = y.permute(0, 2, 3, 1).reshape(-1, latent_y.size(1)).clamp(min=1e-4).log()
            dist.append(F.cross_entropy(y_log, y_inds, reduction='none').view(B, -1).mean(dim=-1))
        return [F.softmin(torch.stack(dist, 1), dim=-1), torch.zeros((), device=latent.device), {}]

    # ---- losses / metrics (ct_mcq_vae.py:297-333) --------------------------------------------------
    def latent_loss(self, latent, latent_y, target_inds=None):
        """target_inds (int64, one index per (b, h, w) row of latent_y in that order): the arg-max of latent_y where the caller
        already holds it -- CTMCQVAE builds latent_y as the one-hot of exactly these indices, so it skips both the one-hot
        and its arg-max (latent_y may then be None)."""
        lat = latent.permute(0, 2, 3, 1).reshape(-1, latent.size(1))        # a view when latent is [B,S,D] memory (forward*)
        if target_inds is not None:
            tgt = target_inds.reshape(-1)
        else:
            tgt = latent_y.detach().permute(0, 2, 3, 1).reshape(-1, latent_y.size(1)).argmax(dim=-1)
        if lat.is_cuda and lat.size(1) <= 64:
            return K.LatentCE.apply(lat, tgt)             # clamp + log + cross-entropy in one launch (csrc/ctmisc.hip)
        return F.cross_entropy(lat.clamp(min=1e-4).log(), tgt)

    def adjacency_KL_loss(self, adj):
        logc = adj.reshape(adj.size(0), -1).log_softmax(dim=-1)
        target = _draw("kl_target", logc.shape, device=logc.device).softmax(dim=-1)
        return F.kl_div(logc, target, reduction="batchmean")

    def graph_size_loss(self, graph):
        return torch.linalg.matrix_norm(graph).mean()

    def positive_trial_loss(self, adj):
        return torch.linalg.vector_norm(_ProdLastDim.apply(1 - adj), dim=-1).mean()

    def causal_accuracy(self, action_probas, action):
        return (torch.argmax(action_probas, dim=-1) == torch.argmax(action, dim=-1)).float().mean()

    def causal_undirected_accuracy(self, action_probas, action):
        dim = action.size(-1)
        rec = F.one_hot(torch.argmax(action_probas, dim=-1), num_classes=dim)
        return self.causal_accuracy(rec[:, dim // 2:] + rec[:, :dim // 2], action[:, dim // 2:] + action[:, :dim // 2])
