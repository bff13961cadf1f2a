"""GPU: the HBM-resident input pipeline (ctvae_crop_resize_u8 through ctvae_amd.data) against the CPU restatement of the
reference's transforms (oracle/data_cpu.py: real F.interpolate, restated center_crop), and a CT-MCQ-VAE step fed by it."""
import numpy as np
import pytest
import torch

from ctvae_amd import data as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from ctvae_amd import native
    native.load()
    return torch.device("cuda")


@pytest.mark.parametrize("H,W,crop,size", [(218, 178, 148, 64),    # CelebA
                                           (64, 64, 148, 64),      # 3DShapes-sized: zero-padded to the crop first
                                           (151, 149, 148, 64),    # odd margins: round-half-even origin
                                           (148, 148, 148, 64), (100, 200, 148, 32)])
def test_crop_resize_matches_reference_transforms(dev, H, W, crop, size):
    from oracle import data_cpu
    g = torch.Generator().manual_seed(H * 1000 + W)
    imgs = torch.randint(0, 256, (7, H, W, 3), generator=g, dtype=torch.uint8)
    rows = torch.tensor([3, 0, 6, 3, 5], dtype=torch.int64)
    want = data_cpu.reference_transform(imgs, rows, crop, size)
    store = D.HbmImageStore(imgs, dev, crop=crop, size=size)
    got = store.fetch(rows)
    assert got.shape == (5, 3, size, size) and got.permute(0, 2, 3, 1).is_contiguous()      # NHWC memory, NCHW view
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), atol=2e-6, rtol=0)


def test_loader_feeds_the_ct_model(dev):
    """A TransitionLoader batch has the shape of the reference's collated batch and drives a CT-MCQ-VAE step."""
    from tests import helpers as H
    from tests.test_ct_gpu import build_ct
    g = torch.Generator().manual_seed(2)
    n, V = 40, 6
    imgs = torch.randint(0, 256, (n, 64, 64, 3), generator=g, dtype=torch.uint8)
    names = [str(i) for i in range(n)]
    table = D.TransitionTable(D.synthetic_transition_csv(names, 90, V, 4), names, V, "all")
    sampler = D.TransitionBatchSampler(table, batch_size=3, shuffle=True, drop_last=True, seed=1)
    loader = D.TransitionLoader(D.HbmImageStore(imgs, dev), table, sampler)
    m = build_ct(dev, 11)
    seen = set()
    for x, target, opt in loader:
        mode = opt["mode"][0]
        if mode in seen:
            continue
        seen.add(mode)
        assert x.shape == (3, 3, 64, 64) and x.is_cuda and opt["mode"] == [mode] * 3
        kw = {"mode": opt["mode"]}
        if mode != "base":
            assert opt["input_y"].shape == (3, 3, 64, 64) and opt["action"].shape == (3, 2 * V)
            kw.update(input_y=opt["input_y"], action=opt["action"])
        out = m(x, **kw)
        losses = m.loss_function(*out)
        assert torch.isfinite(losses["loss"])
        if len(seen) == 3:
            break
    assert seen == {"base", "action", "causal"}


def test_runner_trains_from_an_hbm_store(dev, tmp_path):
    """python -m ctvae_amd.run on a (tiny) real-data layout: .npy image store + variation CSV + partition file."""
    import csv
    import json
    import os
    import yaml
    from ctvae_amd import run as R
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", "ct_mcq_vae.yaml")))
    n, V = 48, int(cfg["model_params"].get("action_dim", 12)) // 2
    g = torch.Generator().manual_seed(8)
    np.save(tmp_path / "imgs.npy", torch.randint(0, 256, (n, 64, 64, 3), generator=g, dtype=torch.uint8).numpy())
    folder = tmp_path / "3dshapes"
    folder.mkdir()
    with open(folder / "list_eval_partition.txt", "w") as f:
        w = csv.writer(f)
        w.writerow(["", "index", "split"])
        for i in range(n):
            w.writerow([i, i, 0 if i < 32 else 2])
    names = [str(i) for i in range(n)]
    text = D.synthetic_transition_csv(names, 400, V, 6)
    # keep only pairs inside one split and label them with that split's code (train 0 / test 2)
    rows = list(csv.reader(text.splitlines()))
    with open(folder / f"variation_attrs_{V}.txt", "w") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            a, b = int(r[1]), int(r[2])
            if (a < 32) == (b < 32):
                r[6] = "0" if a < 32 else "2"
                w.writerow(r)
    cfg["data_params"].update(dataset_name="TShapes3D", data_path=str(tmp_path), hbm_images=str(tmp_path / "imgs.npy"),
                              train_batch_size=4, val_batch_size=4)
    cfg["trainer_params"].update(max_epochs=1, gpus=[0])
    cfg["logging_params"].update(save_dir=str(tmp_path / "logs"))
    cfg["exp_params"]["hipgraph"] = False
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(cfg, f)
    hist = R.main(["-c", str(tmp_path / "cfg.yaml")])
    assert hist and hist[-1]["train_images"] > 0 and np.isfinite(hist[-1]["val_Reconstruction_Loss"])
    assert os.path.exists(tmp_path / "logs" / cfg["logging_params"].get("name", "CTMCQVAE") / "checkpoints" / "last.ckpt")
