"""Entry point mirroring the reference's ``run.py`` (run.py:21-110): ``python -m ctvae_amd.run -c configs/vae.yaml``.

Consumes the reference's YAML layout unchanged (model_params / data_params / exp_params / trainer_params /
logging_params; ``model_params`` is splatted into ``vae_models[name]``), with the tolerances of SURVEY N4:
``find_unused_parameters`` optional, ``gpus`` int or list, ``dataset_name`` defaults to a synthetic source.
One process per GPU (launch with ``python -m torch.distributed.run``); Lightning / wandb are not required.
Checkpoints: top-k on ``val_Reconstruction_Loss`` + ``last.ckpt`` with ``state_dict`` keys prefixed ``model.``
so they interchange with the reference's (run.py:85-97), plus a ``trainer`` entry (optimizer moments, scheduler, epoch, global
step, random streams) for ``trainer_params.resume_from_checkpoint`` without ``load_weights_only`` (run.py:85-101: full resume).
"""
import argparse
import json
import os

import torch
import torch.distributed as dist
import yaml

from . import filler
from .ddp import GradBucketAllReduce
from .experiment import VAEXperiment
from .models import vae_models


class SyntheticData:
    """Synthetic stand-in for dataset.py's VAEDataset with the reference's batch contracts: plain datasets
    yield (X, labels); T-datasets yield (X, labels, {"action", "input_y", "mode"}) with one mode per batch."""

    def __init__(self, data_params, model_params, device, rank=0, world=1, steps_per_epoch=8, seed=0):
        self.p, self.mp, self.dev, self.rank, self.world = data_params, model_params, device, rank, world
        self.steps, self.seed = steps_per_epoch, seed
        name = data_params.get("dataset_name", "synthetic")
        self.transition = name.startswith("T")
        self.action_dim = model_params.get("action_dim", 12)

    def _batches(self, bs, base_seed):
        img = self.p.get("patch_size", 64)
        for i in range(self.steps):
            s = base_seed + 7919 * i + 104729 * self.rank          # each rank takes its own rows of the global batch
            if self.transition:
                x, y, a = filler.synthetic_pairs(s, bs, self.action_dim, img)
                mode = ["base", "action", "causal"][i % 3]
                opts = {"mode": [mode] * bs}
                if mode != "base":
                    opts.update({"action": a.to(self.dev), "input_y": y.to(self.dev)})
                yield (x.to(self.dev), torch.zeros(bs, device=self.dev), opts)
            else:
                x, _ = filler.synthetic_batch(s, bs, img=img)
                yield (x.to(self.dev), torch.zeros(bs, device=self.dev))

    def train(self):
        return self._batches(self.p["train_batch_size"], self.seed)

    def val(self):
        return self._batches(self.p["val_batch_size"], self.seed + 1_000_003)


class HbmData:
    """Real data, resident in HBM (ctvae_amd.data): ``data_params.hbm_images`` names a ``.npy`` uint8 array [N,H,W,3] (the
    decoded dataset, converted once); ``hbm_names`` optionally a text file with one item name per row (default: the row
    number, which is what the disent datasets use, disent_dataset.py:56).  Splits come from
    ``<data_path>/<folder>/list_eval_partition.txt`` (row id, item index, split code; disent_dataset.py:70-80) when it
    exists.  ``T*`` datasets read ``<data_path>/<folder>/variation_attrs_<V>.txt`` (transition.py:111-125) with
    V = action_dim / 2 and yield mode-pure (x, labels, options) batches; others yield (x, labels).  Train batches are
    shuffled and sharded over the ranks; validation uses split ``test`` like dataset.py:88-93."""

    FOLDERS = {"TCeleba": "celeba", "TShapes3D": "3dshapes", "TCars3D": "cars3d", "TDSprites": "dsprites",
               "TSmallNORB": "smallnorb", "TSprites": "sprites"}

    def __init__(self, data_params, model_params, device, rank=0, world=1, seed=0):
        import numpy as np
        from . import data as D
        self.D, self.p, self.dev, self.rank, self.world, self.seed = D, data_params, device, rank, world, seed
        name = data_params.get("dataset_name", "")
        self.transition = name.startswith("T")
        self.folder = os.path.join(data_params.get("data_path", "."), self.FOLDERS.get(name, name.lower().lstrip("t")))
        imgs = torch.from_numpy(np.load(data_params["hbm_images"], mmap_mode="c", allow_pickle=False))
        self.store = D.HbmImageStore(imgs, device, crop=data_params.get("crop_size", 148), size=data_params.get("patch_size", 64))
        if data_params.get("hbm_names"):
            all_names = [l.strip() for l in open(data_params["hbm_names"]) if l.strip()]
        else:
            all_names = [str(i) for i in range(len(self.store))]
        part = os.path.join(self.folder, "list_eval_partition.txt")
        self.split_rows = {}
        if os.path.exists(part):
            import csv
            rows = list(csv.reader(open(part)))[1:]
            for code, split in ((0, "train"), (1, "valid"), (2, "test")):
                self.split_rows[split] = [int(r[1]) for r in rows if int(r[2]) == code]
        else:
            self.split_rows = {s: list(range(len(all_names))) for s in ("train", "valid", "test")}
        self.all_names = all_names
        self.V = int(model_params.get("action_dim", 12)) // 2
        self.epoch = 0

    def _loader(self, split, batch_size, shuffle):
        D = self.D
        rows = torch.tensor(self.split_rows[split], dtype=torch.int64)
        names = [self.all_names[i] for i in self.split_rows[split]]
        store = self.store
        if self.transition:
            table = D.TransitionTable(os.path.join(self.folder, f"variation_attrs_{self.V}.txt"), names, self.V, split)
            sampler = D.TransitionBatchSampler(table, batch_size, shuffle=shuffle, drop_last=True, limit=self.p.get("limit"),
                                               rank=self.rank, world=self.world, seed=self.seed)
            sampler.set_epoch(self.epoch)
            rows_dev = rows.to(self.dev)
            for batch in sampler:
                mode, xr, yr, act = table.resolve_batch(batch)
                x = store.fetch(rows_dev[xr.to(self.dev)])
                opts = {"mode": [mode] * len(batch)}
                if mode != "base":
                    opts.update({"input_y": store.fetch(rows_dev[yr.to(self.dev)]), "action": act.to(self.dev)})
                yield x, torch.zeros(len(batch), device=self.dev), opts
        else:
            g = torch.Generator().manual_seed(self.seed + 7919 * (self.epoch + 1))
            order = rows[torch.randperm(len(rows), generator=g)] if shuffle else rows
            order = D.shard_rows(order, self.rank, self.world)      # same number of batches on every rank
            for i in range(0, len(order), batch_size):
                r = order[i:i + batch_size]
                yield store.fetch(r), torch.zeros(len(r), device=self.dev)

    def train(self):
        self.epoch += 1
        return self._loader("train", self.p["train_batch_size"], True)

    def val(self):
        return self._loader("test", self.p["val_batch_size"], False)


def main(argv=None):
    ap = argparse.ArgumentParser(description='MI355X runner for the ct-vae models')
    ap.add_argument('--config', '-c', dest="filename", metavar='FILE', default='configs/vae.yaml')
    ap.add_argument('--steps-per-epoch', type=int, default=8, help="synthetic data: batches per epoch")
    ap.add_argument('--max-epochs', type=int, default=None)
    args = ap.parse_args(argv)
    with open(args.filename) as f:
        config = yaml.safe_load(f)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gpus = config.get('trainer_params', {}).get('gpus', 1)
    use_gpu = (len(gpus) if isinstance(gpus, (list, tuple)) else int(gpus)) != 0
    if not (use_gpu and torch.cuda.is_available()):
        raise SystemExit("ctvae_amd runs on MI355X GPUs only: the hot path has no CPU fallback (use the reference for gpus: [])")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    seed = config['exp_params'].get('manual_seed', 0)
    torch.manual_seed(seed)                                           # seed_everything (run.py:48)
    mp = dict(config['model_params'])
    model = vae_models[mp['name']](**mp).to(dev)
    # trainer_params.resume_from_checkpoint (run.py:85-101): with load_weights_only the reference loads the model's weights
    # (strict=False) and drops the key; without it the path is splatted into the Lightning Trainer, which restores optimizer,
    # scheduler, epoch and global step and continues -- the "trainer" entry of our checkpoints carries exactly that state
    ckpt_path = config['trainer_params'].get('resume_from_checkpoint')
    weights_only = bool(config['trainer_params'].get('load_weights_only'))
    ckpt = None
    if ckpt_path:
        ckpt = torch.load(ckpt_path, map_location="cpu", weights_only=True)
        model.load_state_dict({k[6:]: v for k, v in ckpt['state_dict'].items() if k.startswith("model.")}, strict=not weights_only)
    ddp = GradBucketAllReduce(model) if world > 1 else None
    log_dir = os.path.join(config['logging_params'].get('save_dir', 'logs/'), config['logging_params'].get('name', mp['name']))
    os.makedirs(os.path.join(log_dir, "checkpoints"), exist_ok=True)
    log_file = open(os.path.join(log_dir, f"metrics_rank{rank}.jsonl"), "a") if rank == 0 else None
    exp = VAEXperiment(model, config['exp_params'], ddp=ddp, log_file=log_file)
    if config['data_params'].get('hbm_images'):
        data = HbmData(config['data_params'], mp, dev, rank, world, seed)
    else:
        data = SyntheticData(config['data_params'], mp, dev, rank, world, args.steps_per_epoch, seed)

    start_epoch = 0
    if ckpt is not None and not weights_only:
        if "trainer" not in ckpt:
            raise SystemExit(f"{ckpt_path} holds weights only (no 'trainer' entry: optimizer moments, scheduler, epoch); "
                             "set trainer_params.load_weights_only to start a new run from its weights")
        exp.load_state_dict(ckpt["trainer"])
        start_epoch = int(ckpt["epoch"]) + 1
        if hasattr(data, "epoch"):
            data.epoch = start_epoch                                  # the shuffling order follows the epoch
    best = []

    def on_epoch_end(epoch, rec):
        if rank != 0:
            return
        print(json.dumps(rec), flush=True)
        state = {"state_dict": {"model." + k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()},
                 "epoch": epoch, "global_step": exp.global_step, "trainer": exp.state_dict()}
        torch.save(state, os.path.join(log_dir, "checkpoints", "last.ckpt"))
        score = rec.get("val_Reconstruction_Loss")
        if score is not None:                                         # ModelCheckpoint(save_top_k=2, monitor=val_Reconstruction_Loss)
            path = os.path.join(log_dir, "checkpoints", f"epoch={epoch}.ckpt")
            torch.save(state, path)
            best.append((score, path))
            best.sort()
            for _, p in best[2:]:
                if os.path.exists(p):
                    os.remove(p)
            del best[2:]

    epochs = args.max_epochs or config['trainer_params'].get('max_epochs', 1)
    hist = exp.fit(data.train, data.val, max_epochs=epochs, on_epoch_end=on_epoch_end, start_epoch=start_epoch)
    if world > 1:
        dist.destroy_process_group()
    return hist


if __name__ == "__main__":
    main()
