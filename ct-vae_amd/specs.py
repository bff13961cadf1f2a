"""State-dict layouts (key, shape, dtype) of the three hot-path models and the synthetic-workload helpers that go with them.

Used by ``bench.py`` (synthetic weights of the benchmarked configurations), ``__graft_entry__.smoke()`` and the tests: the
layouts restate what the reference's constructors register (``vanilla_vae.py:11-75``, ``mcq_vae.py:144-239``,
``ct_mcq_vae.py:78-100``), so that ``filler.fill_state`` can produce a full ``state_dict`` without building a module.
"""
import torch

from . import filler

MCQ_CFG = dict(in_channels=3, embedding_dim=128, hidden_dims=[64, 128, 256], num_embeddings=64, img_size=64,
               codebooks=4, beta=0.25)
CT_CONV_CFG = dict(in_channels=3, embedding_dim=128, hidden_dims=[64, 128, 256], num_embeddings=64, img_size=64,
                   codebooks=1, beta=0.1)

def vanilla_specs():
    """state_dict keys/shapes of VanillaVAE(in_channels=3, latent_dim=128) (vanilla_vae.py:11-75)."""
    s = []
    f32, i64 = torch.float32, torch.int64

    def bn(p, c):
        s.extend([(p + ".weight", (c,), f32), (p + ".bias", (c,), f32), (p + ".running_mean", (c,), f32),
                  (p + ".running_var", (c,), f32), (p + ".num_batches_tracked", (), i64)])

    ci = 3
    for i, c in enumerate([32, 64, 128, 256, 512]):
        s.extend([(f"encoder.{i}.0.weight", (c, ci, 3, 3), f32), (f"encoder.{i}.0.bias", (c,), f32)])
        bn(f"encoder.{i}.1", c)
        ci = c
    s.extend([("fc_mu.weight", (128, 2048), f32), ("fc_mu.bias", (128,), f32),
              ("fc_var.weight", (128, 2048), f32), ("fc_var.bias", (128,), f32),
              ("decoder_input.weight", (2048, 128), f32), ("decoder_input.bias", (2048,), f32)])
    hd = [512, 256, 128, 64, 32]
    for i in range(4):
        s.extend([(f"decoder.{i}.0.weight", (hd[i], hd[i + 1], 3, 3), f32), (f"decoder.{i}.0.bias", (hd[i + 1],), f32)])
        bn(f"decoder.{i}.1", hd[i + 1])
    s.extend([("final_layer.0.weight", (32, 32, 3, 3), f32), ("final_layer.0.bias", (32,), f32)])
    bn("final_layer.1", 32)
    s.extend([("final_layer.3.weight", (3, 32, 3, 3), f32), ("final_layer.3.bias", (3,), f32)])
    return s


def mcq_specs(cfg):
    """state_dict keys/shapes of MCQVAE(**cfg) (mcq_vae.py:144-239)."""
    s = []
    f32 = torch.float32
    hd = list(cfg["hidden_dims"])
    n = len(hd)
    D, K, C = cfg["embedding_dim"], cfg["num_embeddings"], cfg["codebooks"]
    ci = cfg["in_channels"]
    for i, c in enumerate(hd):
        s.extend([(f"encoder.{i}.0.weight", (c, ci, 4, 4), f32), (f"encoder.{i}.0.bias", (c,), f32)])
        ci = c
    s.extend([(f"encoder.{n}.0.weight", (ci, ci, 3, 3), f32), (f"encoder.{n}.0.bias", (ci,), f32)])
    for j in range(6):
        s.extend([(f"encoder.{n + 1 + j}.resblock.0.weight", (ci, ci, 3, 3), f32),
                  (f"encoder.{n + 1 + j}.resblock.2.weight", (ci, ci, 1, 1), f32)])
    s.extend([(f"encoder.{n + 8}.0.weight", (D, ci, 1, 1), f32), (f"encoder.{n + 8}.0.bias", (D,), f32)])
    for i in range(C):
        s.append((f"vq_layer.quantizers.{i}.embedding.weight", (K, D // C), f32))
    s.extend([("decoder.0.0.weight", (ci, D, 3, 3), f32), ("decoder.0.0.bias", (ci,), f32)])
    for j in range(6):
        s.extend([(f"decoder.{1 + j}.resblock.0.weight", (ci, ci, 3, 3), f32),
                  (f"decoder.{1 + j}.resblock.2.weight", (ci, ci, 1, 1), f32)])
    rev = hd[::-1]
    for i in range(n - 1):
        s.extend([(f"decoder.{8 + i}.0.weight", (rev[i], rev[i + 1], 4, 4), f32), (f"decoder.{8 + i}.0.bias", (rev[i + 1],), f32)])
    k = 8 + n - 1
    s.extend([(f"decoder.{k}.0.weight", (rev[-1], cfg["in_channels"], 4, 4), f32), (f"decoder.{k}.0.bias", (cfg["in_channels"],), f32)])
    return s


def ct_layer_specs(action_dim, input_dim=64, hidden=800):
    """state_dict keys/shapes of CausalTransition(input_dim, action_dim) WITHOUT graph_transitioner (torch_geometric's
    GATv2Conv, absent here), in the reference's registration order (ct_mcq_vae.py:78-100)."""
    f32 = torch.float32
    s = [("a_dense.weight", (input_dim, action_dim), f32), ("a_dense.bias", (input_dim,), f32),
         ("pos_encoding.pe", (4096, 1, input_dim), f32)]
    for k in range(action_dim + 1):
        s.extend([(f"graph_discovers.{k}.0.weight", (hidden, 2 * input_dim), f32), (f"graph_discovers.{k}.0.bias", (hidden,), f32),
                  (f"graph_discovers.{k}.2.weight", (1, hidden), f32), (f"graph_discovers.{k}.2.bias", (1,), f32)])
    s.extend([("mask.0.weight", (input_dim, action_dim + input_dim), f32), ("mask.0.bias", (input_dim,), f32)])
    return s



class CTNoise:
    """Noise source for ctvae_amd.models.causal (set_noise_source): the draws of filler.ct_noise, counted per tag."""

    def __init__(self, seed, device):
        self.seed, self.device, self.counts = seed, device, {}

    def reset(self):
        self.counts = {}

    def draw(self, tag, shape, p=0.0):
        k = self.counts.get(tag, 0)
        self.counts[tag] = k + 1
        return filler.ct_noise(self.seed, tag, k, shape, p).to(self.device)
