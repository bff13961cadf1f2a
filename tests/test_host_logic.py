"""CPU: host-side logic of the drop-in boundary — state_dict keys/shapes, packed layouts, the flat buffers,
YAML contract, synthetic batch contract."""
import os

import pytest
import torch
import yaml

from ctvae_amd import filler
from ctvae_amd.models import vae_models
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_vanilla_state_dict_contract():
    m = vae_models["VanillaVAE"](in_channels=3, latent_dim=128, name="VanillaVAE")      # unknown kwargs are swallowed
    specs = H.vanilla_specs()
    sd = m.state_dict()
    assert [k for k, _, _ in specs] == list(sd.keys())
    assert all(tuple(sd[k].shape) == tuple(s) for k, s, _ in specs)
    assert sum(p.numel() for p in m.parameters()) == 3_937_635                           # SURVEY §6 census
    ref = filler.fill_state(specs, 3)
    m.load_state_dict(ref)                                                                # strict
    assert all(torch.equal(v, ref[k]) for k, v in m.state_dict().items())
    # packed layouts
    w = m.encoder[1]._modules["0"].weight
    assert w.stride() == (1, 64, 3 * 32 * 64, 32 * 64)
    wt = m.decoder[0]._modules["0"].weight
    assert wt.shape == (512, 256, 3, 3) and wt.stride() == (256, 1, 3 * 512 * 256, 512 * 256)
    assert m.fc_mu.weight.stride() == (1, 256) and m.fc_var.weight.data_ptr() - m.fc_mu.weight.data_ptr() == 128 * 4
    # gradients are views of the flat gradient buffer with the same layout
    assert all(p.grad is not None and p.grad.stride() == p.stride() for p in m.parameters())
    m.flat_grads.fill_(1.0)
    assert float(m.fc_var.bias.grad.sum()) == 128.0
    m.zero_grad()
    assert float(m.flat_grads.abs().sum()) == 0.0
    # the reference mutates hidden_dims in place and hard-codes 512
    hd = [32, 64, 128, 256, 512]
    vae_models["VAE"](3, 16, hd)
    assert hd == [512, 256, 128, 64, 32]
    with pytest.raises(ValueError):
        vae_models["GaussianVAE"](3, 16, [32, 64])


@pytest.mark.parametrize("cfg", [H.MCQ_CFG, H.CT_CONV_CFG])
def test_mcq_state_dict_contract(cfg):
    m = vae_models["MCQVAE"](**{**cfg, "hidden_dims": list(cfg["hidden_dims"])})
    specs = H.mcq_specs(cfg)
    sd = m.state_dict()
    assert [k for k, _, _ in specs] == list(sd.keys())
    assert all(tuple(sd[k].shape) == tuple(s) for k, s, _ in specs)
    assert sum(p.numel() for p in m.parameters()) == 10_108_163
    m.vq_layer._codebooks()                                                               # back to back after flattening


def test_ctmcqvae_contract():
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "ct_mcq_vae.yaml")))["model_params"]
    m = vae_models["CTMCQVAE"](**cfg)
    keys = list(m.state_dict().keys())
    conv_keys = [k for k, _, _ in H.mcq_specs(H.CT_CONV_CFG)]
    assert [k for k in keys if not k.startswith("ct_layer.")] == conv_keys
    ct = [k for k in keys if k.startswith("ct_layer.")]
    for want in ("ct_layer.a_dense.weight", "ct_layer.pos_encoding.pe", "ct_layer.graph_discovers.12.2.bias",
                 "ct_layer.mask.0.weight", "ct_layer.graph_transitioner.module_0.att",
                 "ct_layer.graph_transitioner.module_0.lin_l.weight", "ct_layer.graph_transitioner.module_0.lin_edge.weight",
                 "ct_layer.graph_transitioner.module_2.bias"):
        assert want in ct, want
    sd = m.state_dict()
    assert sd["ct_layer.graph_discovers.0.0.weight"].shape == (800, 128)                  # Linear(2*64, 800)
    assert sd["ct_layer.graph_transitioner.module_0.lin_l.weight"].shape == (1300, 64)   # heads 13 x 100
    assert sd["ct_layer.graph_transitioner.module_2.lin_r.weight"].shape == (832, 1300)  # heads 13 x 64
    assert sd["ct_layer.mask.0.weight"].shape == (64, 76)
    n_ct = sum(p.numel() for n, p in m.named_parameters() if n.startswith("ct_layer."))
    assert abs(n_ct - 3.70e6) < 0.05e6                                                    # SURVEY A.3 (~3.70 M)
    sl = m.flat_range("ct_layer")                                                         # update_parameters: "ct_layer"
    # (the range may hold a few padding floats: the discoverer banks start 16-byte aligned)
    assert 0 <= (sl.stop - sl.start) - n_ct < 8 and sl.stop == m.flat_params.numel()
    # one-hot formatting round trip (ct_mcq_vae.py:472-496)
    inds = torch.randint(0, 64, (2, 1, 8, 8))
    oh = m.ct_preprocess(inds, (2, 128, 8, 8))
    assert oh.shape == (2, 64, 8, 8) and torch.equal(m.ct_postprocess(oh, (2, 128, 8, 8)), inds)


def test_yaml_contract():
    """Facts of the three in-scope reference configs (configs/vae.yaml, mcq_vae.yaml, ct_mcq_vae.yaml)."""
    want = {
        "vae.yaml": ("VanillaVAE", {"in_channels": 3, "latent_dim": 128}, {"LR": 0.005, "scheduler_gamma": 0.95, "kld_weight": 0.00025, "manual_seed": 1265}, 64),
        "mcq_vae.yaml": ("MCQVAE", {"embedding_dim": 128, "hidden_dims": [64, 128, 256], "num_embeddings": 64, "codebooks": 4, "beta": 0.25},
                         {"LR": 0.0005, "scheduler_gamma": 0.98, "manual_seed": 1320}, 64),
        "ct_mcq_vae.yaml": ("CTMCQVAE", {"action_dim": 12, "codebooks": 1, "beta": 0.1, "gamma": 1.5, "c_alpha": 0.01, "c_beta": 0.4, "c_delta": 0.01, "c_epsilon": 0.1, "noise": "off"},
                            {"LR": 0.0005, "scheduler_gamma": 0.994, "manual_seed": 1250, "update_parameters": "ct_layer", "find_unused_parameters": True}, 16),
    }
    for fn, (name, mp, ep, bs) in want.items():
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", fn)))
        assert set(cfg) == {"model_params", "data_params", "exp_params", "trainer_params", "logging_params"}
        assert cfg["model_params"]["name"] == name and name in vae_models
        assert all(cfg["model_params"][k] == v for k, v in mp.items())
        assert all(cfg["exp_params"][k] == v for k, v in ep.items())
        assert cfg["data_params"]["train_batch_size"] == bs and cfg["data_params"]["patch_size"] == 64


def test_synthetic_batch_contract():
    from ctvae_amd.run import SyntheticData
    d = SyntheticData({"dataset_name": "TShapes3D", "train_batch_size": 4, "val_batch_size": 4, "patch_size": 64},
                      {"action_dim": 12}, torch.device("cpu"), rank=1, world=2, steps_per_epoch=3)
    batches = list(d.train())
    assert [b[2]["mode"][0] for b in batches] == ["base", "action", "causal"]
    x, labels, opts = batches[1]
    assert x.shape == (4, 3, 64, 64) and opts["action"].shape == (4, 12) and opts["input_y"].shape == (4, 3, 64, 64)
    assert float(opts["action"].sum()) == 4.0 and 0.0 <= float(x.min()) and float(x.max()) < 1.0
    d0 = SyntheticData({"train_batch_size": 4, "val_batch_size": 4}, {}, torch.device("cpu"), rank=0, world=2, steps_per_epoch=1)
    d1 = SyntheticData({"train_batch_size": 4, "val_batch_size": 4}, {}, torch.device("cpu"), rank=1, world=2, steps_per_epoch=1)
    assert not torch.equal(next(iter(d0.train()))[0], next(iter(d1.train()))[0])           # ranks see different rows


def test_bank_gradients_written_in_place_reach_the_flat_buffer():
    """kernels.flat_grad_alias (host logic, no kernel involved): a backward that writes a bank's gradient into the alias of the
    flat gradient buffer leaves every parameter of the bank with the right gradient after gather_torch_grads(), whether or not
    autograd kept the alias (it does when nobody else holds the tensor; a clone would only cost the copy back); the alias is
    handed out once per zero_grad, and never after an optimizer-side epoch bump without a zero_grad."""
    import torch
    from ctvae_amd import kernels as K
    from ctvae_amd.models.causal import _DiscoverBank
    from ctvae_amd.models.packing import FlatParamMixin

    class Root(FlatParamMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bank = _DiscoverBank(3, 8, 4)
            self.flatten_parameters()

    class WritesInPlace(torch.autograd.Function):
        @staticmethod
        def forward(ctx, W):
            ctx.save_for_backward(W)
            return W.sum()

        @staticmethod
        def backward(ctx, g):
            (W,) = ctx.saved_tensors
            dW = K.flat_grad_alias(W)
            WritesInPlace.aliased.append(dW is not None)
            if dW is None:
                dW = torch.empty_like(W)
            dW.copy_(torch.arange(W.numel(), dtype=torch.float32).view_as(W) * g)
            return dW

    r = Root()
    plist = r.bank._lists()[0]
    assert K.banked(plist)
    want = torch.arange(plist[0].numel() * len(plist), dtype=torch.float32).view(len(plist), *plist[0].shape)
    for lazy in (True, False):
        WritesInPlace.aliased = []
        r.zero_grad(lazy=lazy)
        WritesInPlace.apply(K.BankView.apply(*plist)).backward()
        assert WritesInPlace.aliased == [True]
        r.gather_torch_grads()
        for k, p in enumerate(plist):
            assert torch.equal(p.grad, want[k])
        base = r._flat_grads.data_ptr()
        assert all(p.grad.data_ptr() == base + 4 * (plist[0].numel() * k + (plist[0].data_ptr() - r._flat_params.data_ptr()) // 4)
                   for k, p in enumerate(plist))
    # a second writer in the same step gets no alias (autograd adds its tensor), nor does anyone after an epoch bump
    r.zero_grad(lazy=True)
    W = K.BankView.apply(*plist)
    assert K.flat_grad_alias(W) is not None and K.flat_grad_alias(W) is None
    r.zero_grad(lazy=True)
    K.bump_param_epoch()
    assert K.flat_grad_alias(W) is None


def test_lazy_zero_grad_host_protocol():
    """FlatParamMixin.zero_grad(lazy=True) / kernels.grad_target / settle_grads on the host (no kernel involved): a block is
    declared zero without a fill, its first writer is told to overwrite (once), blocks nobody wrote are zero after the settle,
    and the parameters of a PackedLinearGroup (one GEMM writes the whole block) share the flag."""
    import torch
    from ctvae_amd import kernels as K
    from ctvae_amd.models.packing import FlatParamMixin, PackedConv, PackedLinear, PackedLinearGroup

    class Root(FlatParamMixin, torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = PackedConv(4, 8, 3)
            self.fc_a, self.fc_b = PackedLinear(16, 4), PackedLinear(16, 4)
            grp = PackedLinearGroup([self.fc_a, self.fc_b])
            self.fc_a._linear_group = self.fc_b._linear_group = grp
            self.flatten_parameters()

    r = Root()
    r._flat_grads.fill_(7.0)                       # stale contents
    r.zero_grad(lazy=True)
    assert float(r._flat_grads.min()) == 7.0       # no fill happened
    g, acc = K.grad_target(r.conv.weight)
    assert acc == 0 and g is r.conv.weight.grad
    g.fill_(1.0)                                   # "the kernel" overwrites
    assert K.grad_target(r.conv.weight)[1] == 1    # a second writer accumulates
    assert K.grad_target(r.fc_a.weight)[1] == 0 and K.grad_target(r.fc_b.weight)[1] == 1      # one flag for the group's block
    r.fc_a.weight.grad.fill_(2.0)
    r.fc_b.weight.grad.fill_(2.0)
    r.settle_grads()
    assert torch.all(r.conv.weight.grad == 1.0) and torch.all(r.fc_a.weight.grad == 2.0) and torch.all(r.fc_b.weight.grad == 2.0)
    for p in (r.conv.bias, r.fc_a.bias, r.fc_b.bias):          # never written: zero after the settle
        assert torch.all(p.grad == 0.0)
    r.zero_grad()                                  # the eager form fills and clears every flag
    assert float(r._flat_grads.abs().max()) == 0.0 and K.grad_target(r.conv.weight)[1] == 1


def test_placeholder_gradient_protocols():
    """Host side of the round-3 hand-overs (no kernel runs): a consumer may leave its data gradient as split-K slices plus an
    unwritten placeholder only for a tensor model code declared single-consumer, the BatchNorm that takes the slices insists on
    exactly that placeholder (anything else would be garbage: it raises), and the element-wise consumers' registry hands a
    placeholder out once, and only while it is untouched."""
    from ctvae_amd import kernels as K
    y = torch.zeros(2, 4, 4, 8)
    link = K.BNLink(y, torch.zeros(8), torch.ones(8), torch.ones(8), torch.zeros(8), K.ACT_LRELU)
    assert link.sole is False
    a = torch.zeros(2, 4, 4, 8)
    a._ctvae_bn_link = link
    assert K.mark_sole_consumer(a) is a and link.sole is True
    assert K.mark_sole_consumer(torch.zeros(3)) is not None            # a tensor without a link: nothing to mark, no error
    g, slices = torch.empty(2, 4, 4, 8), torch.empty(3 * 256)
    geom = (K.CONV, 2, 4, 4, 8, 16, 3, 2, 1, 0)
    link.publish_lazy(g, slices, 3, geom)
    assert link.take(g) == (None, 0, None)                              # the ordinary sums are not published in that mode
    link.publish_lazy(g, slices, 3, geom)
    got = link.take_lazy(g)
    assert got[0] is slices and got[1] == 3 and got[2] == geom
    assert link.take_lazy(g) is None                                    # one use only
    link.publish_lazy(g, slices, 3, geom)
    with pytest.raises(RuntimeError):
        link.take_lazy(g + 1.0)                                         # autograd summed something onto it / another tensor arrived
    link.publish_lazy(g, slices, 3, geom)
    g.add_(0.0)                                                         # touched in place: version changed
    with pytest.raises(RuntimeError):
        link.take_lazy(g)
    # the registry of element-wise consumers
    h = torch.empty(2, 128)
    K.offer_lazy_grad(h, slices, 4)
    assert K.claim_lazy_grad(h.view(2, 1, 1, 128))[1] == 4              # a view (reshape's backward) is the same placeholder
    assert K.claim_lazy_grad(h) is None                                 # handed out once
    K.offer_lazy_grad(h, slices, 4)
    h.mul_(1.0)
    assert K.claim_lazy_grad(h) is None                                 # modified since: not the placeholder any more
    t = K.grad_slices_ok(torch.zeros(2, 3))
    assert t._ctvae_grad_slices_ok is True


def test_chain_marks_intermediates_and_keeps_reference_keys():
    """blocks.Chain is nn.Sequential for state_dict purposes (same child names) and declares the single consumer of every
    intermediate tensor; VanillaVAE's encoder / decoder are Chains, the zoo models that build their own Sequential are not."""
    from ctvae_amd.models import blocks
    m = vae_models["VanillaVAE"](in_channels=3, latent_dim=128)
    assert isinstance(m.encoder, blocks.Chain) and isinstance(m.decoder, blocks.Chain)
    assert isinstance(m.encoder, torch.nn.Sequential)
    assert list(m.encoder._modules) == ["0", "1", "2", "3", "4"] and list(m.decoder._modules) == ["0", "1", "2", "3"]
    assert hasattr(m.final_layer, "reads_lazy_input") and all(hasattr(b, "reads_lazy_input") for b in m.encoder)
    seen = []

    class Probe(torch.nn.Module):
        def forward(self, x):
            seen.append(x)
            return x + 1

    c = blocks.Chain(Probe(), Probe(), Probe())
    out = c(torch.zeros(2))
    assert float(out[0]) == 3.0 and len(seen) == 3
    lv = vae_models["LVAE"](**H.LVAE_CFG)
    assert not isinstance(lv.decoder, blocks.Chain)


def test_clamp_scaling_constants_are_exact_powers_of_two():
    """csrc/satmath.hpp: relu(t) = 2^64 sat(t 2^-64), [t > 0] = sat(t 2^60).  The scalings are exact in binary floating point only
    if the constants are the powers of two they claim to be (and float32 holds them exactly)."""
    import os
    import re
    import numpy as np
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ct-vae_amd", "csrc", "satmath.hpp")).read()
    vals = {m.group(1): float(m.group(2)) for m in re.finditer(r"constexpr float (k\w+) = ([0-9.e+-]+)f;", src)}
    assert set(vals) >= {"kSatDown", "kSatUp", "kStepUp", "kStepDown"}
    want = {"kSatDown": 2.0 ** -64, "kSatUp": 2.0 ** 64, "kStepUp": 2.0 ** 60, "kStepDown": 2.0 ** -60}
    for k, w in want.items():
        assert float(np.float32(vals[k])) == w, (k, vals[k], w)
    assert float(np.float32(vals["kSatDown"]) * np.float32(vals["kSatUp"])) == 1.0
    assert float(np.float32(vals["kStepDown"]) * np.float32(vals["kStepUp"])) == 1.0


def test_hot_kernels_keep_their_loops_free_of_scratch_traffic():
    """tools/isa_audit.py on the compiler's assembly (hipcc -S, no GPU): a spill reload inside a main loop is a vector-memory
    operation whose s_waitcnt vmcnt(0) also drains the loads issued ahead for the next iteration -- up_wgrad_kernel ran 45 instead of
    29 us that way.  The kernels below must have no scratch at all (up_wgrad: register budget of one wave per SIMD) or none inside
    their loops."""
    import os
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        import pytest
        pytest.skip("hipcc not available")
    files = [os.path.join(root, "ct-vae_amd", "csrc", f) for f in ("upconv.hip", "pairmlp.hip")]
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_audit.py")] + files, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-500:]
    blocks, cur = {}, None
    for line in out.stdout.splitlines():
        if not line.startswith("    "):
            cur = line.split(": ", 1)[1] if ": " in line else line
            blocks[cur] = []
        elif cur is not None:
            blocks[cur].append(line)
    def block(prefix):
        hits = [k for k in blocks if k.startswith(prefix)]
        assert hits, (prefix, list(blocks))
        return [l for k in hits for l in blocks[k]]
    for name in ("void up_wgrad_kernel<true>", "void up_wgrad_kernel<false>", "up_fwd_kernel", "void pair_mlp_bwd64_kernel<4>"):
        lines = block(name)
        assert any("scratch 0 B" in l for l in lines), (name, lines[:2])
        assert not any(l.strip().startswith("loop") and l.rstrip().endswith("scratch") for l in lines), (name, lines)
    # up_fwd_kernel's epilogues decide the launch-uniform activation once: as act_fwd(v, a.act) per value the kernel was ~4 900 lines
    # with a branch tree per stored value (DESIGN.md 4.8 "Epilogues"); the audit's branch count pins that
    import re
    head = [l for l in block("up_fwd_kernel") if " branches" in l][0]
    nbranch = int(re.search(r"(\d+) branches", head).group(1))
    ninstr = int(re.search(r"(\d+) instructions", head).group(1))
    assert nbranch <= 80 and ninstr <= 4200, head


def test_tile_kernels_fetch_their_arguments_in_one_batch_and_keep_their_epilogues_flat():
    """tools/isa_audit.py on tapgemm_fast.hip (hipcc -S, no GPU).  Pins two round-3 findings: (1) the tile kernels read
    their 1.9 KB argument block behind ONE batch of scalar loads (common.hpp kernarg_warm) -- eight dependent batches in front of the
    first operand load were 2.9 us per workgroup of a 9 us launch; (2) their sixteen-value epilogue decides the launch's options once
    per group of values -- as per-value `if (a.add) ... if (a.act == ...)` chains it was ~600 branches / 7 000 lines."""
    import os
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        import pytest
        pytest.skip("hipcc not available")
    src = os.path.join(root, "ct-vae_amd", "csrc", "tapgemm_fast.hip")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "isa_audit.py"), src], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-500:]
    lines = out.stdout.splitlines()
    heads = {lines[i].split(": ", 1)[1]: lines[i + 1] for i in range(len(lines) - 1) if not lines[i].startswith("    ") and ": " in lines[i]}
    tile = [v for k, v in heads.items() if k.startswith("void tapgemm_fast_kernel<2, 2, 1, 1, false, 3, false")]
    pair = [v for k, v in heads.items() if k.startswith("void conv_bwd_pair_kernel<false")]
    assert tile and pair, list(heads)
    for h in tile + pair:
        assert "argument lines fetched in one batch" in h, h
    assert int(re.search(r"(\d+) scalar-load batches", tile[0]).group(1)) <= 2, tile[0]
    assert int(re.search(r"(\d+) branches", tile[0]).group(1)) <= 220, tile[0]
    assert int(re.search(r"(\d+) instructions", tile[0]).group(1)) <= 4600, tile[0]
