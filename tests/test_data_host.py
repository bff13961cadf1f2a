"""CPU: the input-side host logic (ctvae_amd.data) against the fixture produced by the reference's own
TransitionDataset / TransitionBatchSampler (oracle/gen_data_golden.py -> tests/golden/data_transition.npz)."""
import os

import numpy as np
import pytest
import torch

from ctvae_amd import data as D

GOLD = os.path.join(os.path.dirname(__file__), "golden", "data_transition.npz")


@pytest.fixture(scope="module")
def gold():
    g = np.load(GOLD)
    names = [f"img_{i:04d}.png" for i in range(int(g["n_items"]))]
    text = D.synthetic_transition_csv(names, int(g["n_rows"]), int(g["V"]), int(g["seed"]))
    return g, names, text


@pytest.mark.parametrize("split", ["train", "valid", "test", "all"])
def test_transition_table_matches_reference(gold, split):
    g, names, text = gold
    t = D.TransitionTable(text, names, int(g["V"]), split)
    np.testing.assert_array_equal(t.x_index.numpy(), g[f"{split}.x_index"])
    np.testing.assert_array_equal(t.y_index.numpy(), g[f"{split}.y_index"])
    np.testing.assert_array_equal(t.actions.numpy(), g[f"{split}.actions"])
    assert len(t) == int(g[f"{split}.len"])
    res = np.array([t.resolve(i) for i in range(len(t))], dtype=np.int64)
    np.testing.assert_array_equal(res, g[f"{split}.resolve"])


@pytest.mark.parametrize("split", ["train", "all"])
@pytest.mark.parametrize("drop_last", [True, False])
def test_sequential_batches_match_reference(gold, split, drop_last):
    g, names, text = gold
    t = D.TransitionTable(text, names, int(g["V"]), split)
    s = D.TransitionBatchSampler(t, batch_size=4, shuffle=False, drop_last=drop_last)
    batches = list(s)
    assert len(batches) == len(s)
    np.testing.assert_array_equal(np.array([len(b) for b in batches]), g[f"{split}.seq.drop{int(drop_last)}.sizes"])
    np.testing.assert_array_equal(np.array([i for b in batches for i in b]), g[f"{split}.seq.drop{int(drop_last)}.flat"])


def test_shuffled_batches_are_mode_pure_and_cover_everything(gold):
    g, names, text = gold
    t = D.TransitionTable(text, names, int(g["V"]), "all")
    s = D.TransitionBatchSampler(t, batch_size=5, shuffle=True, drop_last=False, seed=3)
    a, b = list(s), list(s)
    assert a == b                                   # same epoch -> same order
    s.set_epoch(1)
    assert list(s) != a
    seen = []
    for batch in a:
        mode, xr, yr, act = t.resolve_batch(batch)  # raises if the batch mixes modes
        assert (yr is None) == (mode == "base") and (act is None) == (mode == "base")
        if act is not None:
            assert act.shape == (len(batch), 2 * int(g["V"])) and bool((act.sum(1) == 1).all())
        seen += batch
    assert sorted(seen) == list(range(len(t)))
    with pytest.raises(ValueError):
        t.resolve_batch([0, t.num_base])            # base + action in one batch


@pytest.mark.parametrize("drop_last", [True, False])
def test_ranks_take_disjoint_batches(gold, drop_last):
    g, names, text = gold
    t = D.TransitionTable(text, names, int(g["V"]), "all")
    world = 4
    per_rank = [list(D.TransitionBatchSampler(t, batch_size=4, shuffle=True, drop_last=drop_last, rank=r, world=world, seed=9))
                for r in range(world)]
    assert len({len(p) for p in per_rank}) == 1     # every rank runs the same number of steps
    all_b = [tuple(b) for p in per_rank for b in p]
    single = [tuple(b) for b in D.TransitionBatchSampler(t, batch_size=4, shuffle=True, drop_last=drop_last, seed=9)]
    if drop_last:
        assert len(set(all_b)) == len(all_b) and set(all_b) <= set(single)
    else:
        assert set(all_b) == set(single)            # wrap-around padding repeats a few batches, loses none


def test_limit_restricts_each_mode(gold):
    g, names, text = gold
    t = D.TransitionTable(text, names, int(g["V"]), "all")
    s = D.TransitionBatchSampler(t, batch_size=2, shuffle=False, drop_last=True, limit=6)
    assert s.batches_per_mode == [3, 3, 3]
    for batch in s:
        t.resolve_batch(batch)


def test_center_crop_restatement_rounding():
    """oracle.data_cpu.center_crop: odd differences round half to even (Python round), small images are zero-padded."""
    from oracle import data_cpu
    img = torch.arange(3 * 7 * 9, dtype=torch.float32).view(3, 7, 9)
    c = data_cpu.center_crop(img, 4)               # (7-4)/2 = 1.5 -> 2, (9-4)/2 = 2.5 -> 2
    assert torch.equal(c, img[:, 2:6, 2:6])
    p = data_cpu.center_crop(torch.ones(3, 2, 3), 6)
    assert p.shape == (3, 6, 6) and float(p.sum()) == 3 * 2 * 3 and float(p[:, 2:4, 1:4].sum()) == 18
