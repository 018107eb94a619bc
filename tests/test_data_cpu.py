"""Data-path host logic on CPU: the oracle restatement (oracle/data.py) against an independent formulation, and the
product's samplers (medical_image_generation_amd/data.py) against the oracle under the same numpy seed (same RNG call order ==
the boxes a seeded reference loader would draw).  PARITY UNPINNED against the reference itself: see oracle/data.py."""
import numpy as np
import pytest

from oracle import data as od


def _independent_crop(image, bbox, pad_value):
    """Pad the whole volume by the largest overhang, then slice."""
    k = len(bbox)
    lead = image.ndim - k
    if any(hi <= 0 or lo >= n for (lo, hi), n in zip(bbox, image.shape[lead:])):
        return np.zeros(list(image.shape[:lead]) + [hi - lo for lo, hi in bbox], image.dtype)
    m = max([0] + [-lo for lo, _ in bbox] + [hi - n for (_, hi), n in zip(bbox, image.shape[lead:])])
    big = np.pad(image, [(0, 0)] * lead + [(m, m)] * k, constant_values=pad_value)
    return big[tuple([slice(None)] * lead + [slice(lo + m, hi + m) for lo, hi in bbox])]


def test_oracle_crop_and_pad_matches_independent_formulation():
    rng = np.random.default_rng(0)
    for _ in range(200):
        k = int(rng.integers(1, 4))
        shape = tuple(int(v) for v in rng.integers(1, 9, size=k + 1))
        img = rng.standard_normal(shape).astype(np.float32)
        bbox = []
        for n in shape[1:]:
            lo = int(rng.integers(-6, n + 4))
            bbox.append([lo, lo + int(rng.integers(1, 8))])
        pad = float(rng.choice([0.0, -1.5]))
        got, want = od.crop_and_pad_nd(img, bbox, pad), _independent_crop(img, bbox, pad)
        assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("patch,shape", [((16, 32, 32), (40, 64, 70)), ((16, 32, 32), (10, 20, 90)), ((32, 32), (5, 50, 40))])
def test_patch_sampler_draws_the_oracles_boxes(patch, shape):
    from medical_image_generation_amd.data import PatchSampler
    is_2d = len(patch) == 2
    ps = PatchSampler(patch, batch_size=4, oversample_foreground_percent=0.33)
    ob = od.BBoxSampler((1,) + patch if is_2d else patch)
    locs = {1: np.array([[0, 3, 10, 12], [0, 4, 30, 33]])[:, 1:] if not is_2d else np.array([[2, 10, 12]]), 2: np.zeros((0, 3), int)}
    for fg in (False, True):
        np.random.seed(7)
        a = [ps.get_bbox(shape, fg, locs) for _ in range(50)]
        np.random.seed(7)
        b = [ob.get_bbox(shape, fg, locs, is_2d=is_2d) for _ in range(50)]
        assert a == b
        for lbs, ubs in a:
            assert [u - l for l, u in zip(lbs, ubs)] == list(ps.initial_patch_size)
    # the last round(B * (1 - p)).. samples of a batch are forced foreground (DATA:426-428)
    assert [ps.force_foreground(i) for i in range(4)] == [od.oversample_last_percent(i, 4, 0.33) for i in range(4)] == [False, False, False, True]


def test_batch_order_uses_every_sample_before_repeating():
    from medical_image_generation_amd.data import BatchOrder
    np.random.seed(3)
    got = list(BatchOrder(10, 4, number_of_steps=7))
    np.random.seed(3)
    assert got == od.batch_sample_order(10, 4, 7)
    assert len(got) == 7 and all(len(b) == 4 and [i for i, _ in b] == [0, 1, 2, 3] for b in got)
    first_round = [s for b in got[:2] for _, s in b]
    assert len(set(first_round)) == 8  # no repeats while unused samples remain
    assert list(BatchOrder(5, 2, 3, shuffle=False)) == [[(0, 0), (1, 1)], [(0, 2), (1, 3)], [(0, 0), (1, 1)]]
