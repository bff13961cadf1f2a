"""Input side of the hot path: transition table, mode-pure batch sampler, HBM-resident images.

What the reference does (``dataset.py:60-166``, ``datasets/transition.py``): a ``TransitionDataset`` wraps an image dataset
and a CSV ``variation_attrs_<V>.txt`` (columns: row id, input name, output name, variation id, source, target, split);
its index space is ``[ base images | action pairs | causal pairs ]``; ``TransitionBatchSampler`` emits batches that never
mix modes; every image goes ToTensor -> CenterCrop(148) -> Resize(patch_size) on DataLoader worker processes and crosses
PCIe every step.

MI355X-first restatement: 288 GB of HBM hold the whole decoded dataset (CelebA 202 599 x 218 x 178 x 3 bytes = 23.6 GB;
3DShapes 480 000 x 64 x 64 x 3 = 5.9 GB), so the images are uploaded ONCE as uint8 and a batch is a list of row numbers:
one HIP launch (``ctvae_crop_resize_u8``) gathers the rows, converts to [0,1] floats, centre-crops (zero padding when the
image is smaller than the crop, as torchvision does) and resizes bilinearly (align_corners=False, no antialias: what
``transforms.Resize`` does to a tensor) straight into the NHWC fp32 batch the encoder reads.  No worker processes, no
host copies in the step.

Contracts kept from the reference (pinned by tests/golden/data_transition.npz, generated from the reference's own
classes by oracle/gen_data_golden.py): CSV columns and split codes (train 0, valid 1, test 2), pair -> item resolution
through the dataset's name list, one-hot actions ``[2V]`` with the direction bit ``target < source``, index space layout,
one mode per batch, sequential batch order, ``drop_last``.  Deliberate difference: with ``world > 1`` the reference hands
every rank the FIRST batches of each mode (each rank restarts the per-mode iterators, transition.py:176-186); here the
batch list is sharded disjointly, rank r takes positions r, r+world, ...
"""
import csv
import io
import os
from typing import Iterator, List, Optional, Sequence

import torch

MODES = ("base", "action", "causal")
SPLIT_CODES = {"train": (0,), "valid": (1,), "test": (2,), "all": (0, 1, 2)}


def synthetic_transition_csv(names: Sequence[str], n_rows: int, num_variations: int, seed: int) -> str:
    """Deterministic ``variation_attrs_<V>.txt`` text over the given item names (tests and the synthetic bench)."""
    g = torch.Generator().manual_seed(seed)
    r = lambda hi: int(torch.randint(0, hi, (1,), generator=g))
    buf = io.StringIO()
    w = csv.writer(buf, lineterminator="\n")
    w.writerow(["", "input", "output", "variation", "source", "target", "split"])
    for i in range(n_rows):
        a, b = r(len(names)), r(len(names))
        w.writerow([i, names[a], names[b], r(num_variations), r(5), r(5), r(3)])
    return buf.getvalue()


class TransitionTable:
    """The pairs of one split and the index space [base | action | causal] (TransitionDataset, transition.py:14-110)."""

    def __init__(self, csv_source, names: Sequence[str], num_variations: int, split: str = "train"):
        text = open(csv_source).read() if os.path.exists(str(csv_source)) else str(csv_source)
        rows = list(csv.reader(io.StringIO(text)))[1:]
        keep = SPLIT_CODES[split]
        pos = {}
        for i, n in enumerate(names):       # first occurrence, as list.index() resolves it
            pos.setdefault(n, i)
        xs, ys, acts = [], [], []
        for row in rows:
            if int(row[6]) not in keep:
                continue
            xs.append(pos[row[1]])
            ys.append(pos[row[2]])
            direction = int(int(row[5]) < int(row[4]))
            acts.append(num_variations * direction + int(row[3]))
        self.num_variations = num_variations
        self.num_base = len(names)
        self.x_index = torch.tensor(xs, dtype=torch.int64)
        self.y_index = torch.tensor(ys, dtype=torch.int64)
        self.actions = torch.zeros((len(xs), 2 * num_variations), dtype=torch.float32)
        if xs:
            self.actions[torch.arange(len(xs)), torch.tensor(acts)] = 1.0

    @property
    def num_pairs(self) -> int:
        return int(self.x_index.numel())

    def __len__(self) -> int:
        return self.num_base + 2 * self.num_pairs

    def mode_range(self, mode: str) -> range:
        ld, lt = self.num_base, self.num_pairs
        return {"base": range(ld), "action": range(ld, ld + lt), "causal": range(ld + lt, ld + 2 * lt)}[mode]

    def resolve(self, idx: int):
        """(mode id, x item, y item or -1) of one index."""
        ld, lt = self.num_base, self.num_pairs
        if idx < ld:
            return 0, idx, -1
        m = 1 if idx < ld + lt else 2
        p = idx - ld - (m - 1) * lt
        return m, int(self.x_index[p]), int(self.y_index[p])

    def resolve_batch(self, batch: Sequence[int]):
        """(mode name, x rows [B], y rows [B] or None, actions [B,2V] or None) of a mode-pure batch."""
        idx = torch.as_tensor(batch, dtype=torch.int64)
        ld, lt = self.num_base, self.num_pairs
        if int(idx.max()) < ld:
            return "base", idx, None, None
        m = 1 if int(idx.max()) < ld + lt else 2
        p = idx - ld - (m - 1) * lt
        if int(p.min()) < 0:
            raise ValueError("batch mixes modes")
        return MODES[m], self.x_index[p], self.y_index[p], self.actions[p]


class TransitionBatchSampler:
    """Mode-pure batches (TransitionBatchSampler, transition.py:119-192): per mode the indices are cut into batches,
    the list of (mode, batch number) pairs is walked sequentially or in a seeded permutation; with ``world > 1`` that
    list is sharded over the ranks (padded by wrap-around, or truncated with ``drop_last``, to a multiple of ``world``)."""

    def __init__(self, table: TransitionTable, batch_size: int, shuffle: bool, drop_last: bool, limit: Optional[int] = None,
                 rank: int = 0, world: int = 1, seed: int = 0):
        self.table, self.batch_size, self.shuffle, self.drop_last = table, batch_size, shuffle, drop_last
        self.rank, self.world, self.seed, self.epoch = rank, world, seed, 0
        g = torch.Generator().manual_seed(seed)
        self.indices: List[torch.Tensor] = []
        for m in MODES:
            r = table.mode_range(m)
            ids = torch.arange(r.start, r.stop, dtype=torch.int64)
            if limit is not None:
                ids = ids[torch.randperm(len(ids), generator=g)[:limit]]
            self.indices.append(ids)
        nb = [(len(i) // batch_size) if drop_last else -(-len(i) // batch_size) for i in self.indices]
        self.batches_per_mode = nb
        self.meta = [m for m in range(3) for _ in range(nb[m])]

    def set_epoch(self, epoch: int):
        self.epoch = epoch

    def _meta_order(self) -> List[int]:
        n = len(self.meta)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + 7919 * (self.epoch + 1))
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        if self.world > 1:
            if self.drop_last:
                order = order[:n - n % self.world]
            elif n % self.world:
                order = order + order[:self.world - n % self.world]
            order = order[self.rank::self.world]
        return order

    def __len__(self) -> int:
        return len(self._meta_order())

    def __iter__(self) -> Iterator[List[int]]:
        per_mode = []
        for m in range(3):
            ids = self.indices[m]
            if self.shuffle:
                g = torch.Generator().manual_seed(self.seed + 104729 * (self.epoch + 1) + m)
                ids = ids[torch.randperm(len(ids), generator=g)]
            per_mode.append(ids)
        # batch k of mode m = k-th slice; a meta position maps to (mode, how many of that mode came before it)
        seen = [0, 0, 0]
        slot = []
        for m in self.meta:
            slot.append((m, seen[m]))
            seen[m] += 1
        for p in self._meta_order():
            m, k = slot[p]
            yield per_mode[m][k * self.batch_size:(k + 1) * self.batch_size].tolist()


def shard_rows(order: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rows of one rank for a plain (non-transition) dataset: ``torch.utils.data.DistributedSampler`` semantics — the
    order is wrap-padded with its own head to a multiple of ``world`` and dealt round-robin, so EVERY rank gets
    ceil(N / world) rows and therefore the same number of batches.  (Each training step issues a gradient all-reduce
    and each validation batch a scalar all-reduce: a rank with one batch more than its peers would wait in RCCL for ever.)"""
    n = int(order.numel())
    if world <= 1 or n == 0:
        return order
    total = -(-n // world) * world
    if total > n:
        reps = -(-total // n)
        order = order.repeat(reps)[:total]
    return order[rank::world]


class HbmImageStore:
    """A decoded uint8 dataset ``[N,H,W,3]`` resident in HBM; ``fetch`` is the reference's per-sample transform pipeline
    (ToTensor -> CenterCrop -> Resize, dataset.py:72-80) for a whole batch in one launch."""

    def __init__(self, images_u8: torch.Tensor, device, crop: int = 148, size: int = 64):
        if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[-1] != 3:
            raise ValueError("expected a uint8 tensor [N,H,W,3]")
        self.data = images_u8.to(device).contiguous()
        self.crop, self.size = crop, size

    def __len__(self):
        return self.data.shape[0]

    def fetch(self, rows: torch.Tensor) -> torch.Tensor:
        """rows: int64 [B] (any device) -> logical NCHW ``[B,3,size,size]`` fp32 view of an NHWC batch in HBM."""
        from . import native
        if not self.data.is_cuda:
            raise RuntimeError("HbmImageStore.fetch needs the store on a GPU (there is no CPU fallback on the product path)")
        rows = rows.to(self.data.device, dtype=torch.int64).contiguous()
        B, (N, H, W, _) = rows.numel(), self.data.shape
        out = torch.empty((B, self.size, self.size, 3), dtype=torch.float32, device=self.data.device)
        native.call("ctvae_crop_resize_u8", self.data.data_ptr(), rows.data_ptr(), out.data_ptr(), B, N, H, W, self.crop, self.size)
        return out.permute(0, 3, 1, 2)


class TransitionLoader:
    """Iterates ``(x, target, options)`` exactly as the reference's DataLoader does for a TransitionDataset
    (transition.py:87-107; experiment.py:44-50 consumes ``real_img, labels, options``): ``options`` holds ``mode`` (a list
    of B equal strings, as default_collate produces) and, for pair modes, ``action`` [B,2V] and ``input_y`` [B,3,S,S]."""

    def __init__(self, store: HbmImageStore, table: TransitionTable, sampler: TransitionBatchSampler, labels: Optional[torch.Tensor] = None):
        self.store, self.table, self.sampler = store, table, sampler
        dev = store.data.device
        self.labels = labels.to(dev) if labels is not None else None
        self.actions = table.actions.to(dev)
        self.x_index, self.y_index = table.x_index.to(dev), table.y_index.to(dev)

    def __len__(self):
        return len(self.sampler)

    def __iter__(self):
        for batch in self.sampler:
            mode, xr, yr, act = self.table.resolve_batch(batch)
            x = self.store.fetch(xr)
            target = self.labels[xr.to(self.labels.device)] if self.labels is not None else xr.to(x.device)
            options = {"mode": [mode] * len(batch)}
            if mode != "base":
                options["input_y"] = self.store.fetch(yr)
                options["action"] = act.to(x.device)
            yield x, target, options
