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


class _ScriptedRandint:
    """np.random.randint replaced by a script: records (low, high) of every call, returns low / high - 1 / the middle as told."""

    def __init__(self, picks):
        self.picks, self.calls = list(picks), []

    def __call__(self, low, high=None):
        self.calls.append((int(low), int(high)))
        how = self.picks.pop(0)
        return {"lo": low, "hi": high - 1, "mid": (low + high - 1) // 2}[how]


def test_patch_sampler_boxes_derived_by_hand(monkeypatch):
    """Boxes worked out on paper from the rules of MedicalDataset.get_bbox (DATA:473-528), not from another implementation: the
    generator is scripted, so every draw's RANGE (checked) and the resulting corners (checked) are hand-derived.
      A. volume 40x100x120, patch 32x64x64: slice axis free in [0, 8]; H: centre 50, margins 18/18 -> jitter +-10; W: centre 60,
         margins 28/28 -> jitter +-10.  All draws at their lower end: corners (0, 50-10-32, 60-10-32) = (0, 8, 18).
      B. volume 16x48x200 (shorter than the patch in D and H): D needs 16 voxels of padding -> the only corner is -8; H is narrower
         than the box -> centred, 24 - 32 = -8, and NO jitter draw; W: centre 100, jitter +-10, upper end -> 100 + 10 - 32 = 78.
      C. volume 33x64x70, patch 32x64x64: D free in [0, 1]; H has no margin (32 - 32 = 0) -> no draw, corner 0; W: centre 35, margins
         min(10, 35-32, 70-35-32) = 3 -> jitter +-3, middle of [-3, 3] = 0 -> corner 3.
      D. as A with a forced foreground voxel (z, y, x) = (39, 5, 7): D corner = clamp(39 - 16, 0, 8) = 8; H / W are in-plane (jitter)."""
    from medical_image_generation_amd.data import PatchSampler
    ps = PatchSampler((32, 64, 64), batch_size=2)
    r = _ScriptedRandint(["lo"] * 5)
    monkeypatch.setattr(np.random, "randint", r)
    assert ps.get_bbox((40, 100, 120), False, None) == ([0, 8, 18], [32, 72, 82])
    assert r.calls == [(0, 9), (0, 37), (0, 57), (-10, 11), (-10, 11)]
    r = _ScriptedRandint(["lo", "lo", "lo", "hi"])
    monkeypatch.setattr(np.random, "randint", r)
    assert ps.get_bbox((16, 48, 200), False, None) == ([-8, -8, 78], [24, 56, 142])
    assert r.calls == [(-8, -7), (-8, -7), (0, 137), (-10, 11)]
    r = _ScriptedRandint(["hi", "lo", "lo", "mid"])
    monkeypatch.setattr(np.random, "randint", r)
    assert ps.get_bbox((33, 64, 70), False, None) == ([1, 0, 3], [33, 64, 67])
    assert r.calls == [(0, 2), (0, 1), (0, 7), (-3, 4)]
    r = _ScriptedRandint(["lo"] * 3 + ["hi", "hi"])
    monkeypatch.setattr(np.random, "randint", r)
    monkeypatch.setattr(np.random, "choice", lambda a: a[0] if not isinstance(a, (int, np.integer)) else 0)
    locs = {1: np.array([[39, 5, 7]]), 2: np.zeros((0, 3), int)}
    assert ps.get_bbox((40, 100, 120), True, locs) == ([8, 28, 38], [40, 92, 102])
    # 2-D (pseudo 3-D, one slice thick): the slice of the foreground voxel, in-plane centred
    ps2 = PatchSampler((32, 32), batch_size=2)
    r = _ScriptedRandint(["lo"] * 5)
    monkeypatch.setattr(np.random, "randint", r)
    assert ps2.get_bbox((5, 50, 40), True, {1: np.array([[2, 10, 12]])}) == ([2, 25 - 9 - 16, 20 - 4 - 16], [3, 32, 32])
    assert r.calls == [(0, 5), (0, 19), (0, 9), (-9, 10), (-4, 5)]


def test_batch_order_keeps_its_deck_across_epochs():
    """CustomBatchSampler shuffles `self.indices` IN PLACE at the start of every epoch (DATA:615-616), so the second epoch's deck is
    a permutation of the first epoch's, not of range(n): same seed, two epochs, against the oracle run twice on one list."""
    from medical_image_generation_amd.data import BatchOrder
    np.random.seed(11)
    bo = BatchOrder(9, 2, number_of_steps=6)
    e1, e2 = list(bo), list(bo)
    np.random.seed(11)
    idx = list(range(9))
    want = []
    for _ in range(2):
        np.random.shuffle(idx)
        deck, run = list(idx), []
        while len(run) < 12:
            if len(deck) < 2:
                deck = list(idx)
                np.random.shuffle(deck)
            run += deck[:2]
            deck = deck[2:]
        want.append([[(0, run[2 * k]), (1, run[2 * k + 1])] for k in range(6)])
    assert [e1, e2] == want and e1 != e2
    # fewer samples than a batch: the reference keeps dealing short hands until the run is long enough
    np.random.seed(1)
    short = list(BatchOrder(3, 4, number_of_steps=2, shuffle=False))
    assert short == [[(0, 0), (1, 1), (2, 2), (3, 0)], [(0, 1), (1, 2), (2, 0), (3, 1)]]


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


def test_augmentation_host_logic():
    """Host side of the GPU transform chain (medical_image_generation_amd/augment.py): samplers, filter taps, the affine, the list
    define_nnunet_transformations builds from the soft setting (DATA:748-859, 399-424).  No kernel runs here."""
    import torch
    from medical_image_generation_amd import augment as A
    np.random.seed(0)
    draws = np.array([A.BGContrast((0.9, 1.1))() for _ in range(4000)])
    assert draws.min() >= 0.9 and draws.max() <= 1.1 and abs((draws < 1).mean() - 0.5) < 0.05
    only_up = np.array([A.BGContrast((1.0, 1.5))() for _ in range(200)])
    assert only_up.min() >= 1.0
    for sigma in (0.5, 0.7, 1.0):
        t = A.gaussian_taps(sigma)
        assert len(t) % 2 == 1 and len(t) == len(od.aug_gaussian_taps(sigma)) and abs(t.sum() - 1) < 1e-6
        assert np.allclose(t, od.aug_gaussian_taps(sigma).numpy(), atol=1e-7)
    assert np.array_equal(A.affine_matrix([0, 0, 0], [1, 1, 1]), np.eye(3, dtype=np.float32))
    m = A.affine_matrix([0.1, 0, 0], [1.1, 1.1, 1.1])  # about the depth axis: d is only scaled, (h, w) rotate
    assert np.allclose(m[0], [1.1, 0, 0]) and np.allclose(m @ m.T, 1.21 * np.eye(3), atol=1e-6)
    args = dict(patch_size=(8, 16, 16), rotation=True, scaling=True, mirror=True, gaussian_noise=False, gaussian_blur=True, brightness=True,
                contrast=False, low_resolution=False, gamma=True, dummy_2d=False)
    p = A.soft_setting(args, dim=3)
    assert p["mirror_axes"] == (2,) and p["scaling_range"] == p["brightness_range"] == p["gamma_range"] == (0.9, 1.1) and p["contrast_range"] is None
    angles = [[p["rot_for_da"](image=None, dim=d) for d in range(3)] for _ in range(50)]
    assert all(a[1] == 0 and a[2] == 0 and abs(a[0]) <= 0.174533 for a in angles) and any(a[0] != 0 for a in angles)
    assert A.soft_setting(args, dim=2)["mirror_axes"] == (1,)
    chain = A.define_nnunet_transformations(p)
    kinds = [type(getattr(t, "transform", t)).__name__ for t in chain.transforms]
    assert kinds == ["SpatialTransform", "GaussianBlurTransform", "MultiplicativeBrightnessTransform", "GammaTransform", "GammaTransform",
                     "MirrorTransform"]
    with pytest.raises(RuntimeError):  # no CPU fallback
        chain(image=torch.zeros(1, 8, 16, 16))
    with pytest.raises(NotImplementedError):
        A.define_nnunet_transformations(dict(p, dummy_2d=True))


# ---- reading the reference's preprocessed cases (medical_image_generation_amd/volume_io.py).  zarr / numcodecs / blosc are absent:
# the chunk files below are written by an encoder stated from the published Blosc-1 chunk format and zarr v2 layout (PARITY UNPINNED).
def _blosc_frame(data: bytes, typesize: int, shuffle: int, blocksize: int, cname="zstd", split=False) -> bytes:
    import struct
    import zlib as _z

    import pyarrow as pa
    code = {"zlib": 3, "zstd": 4}[cname]
    comp = (lambda b: _z.compress(b)) if cname == "zlib" else (lambda b: pa.Codec("zstd").compress(b, asbytes=True))
    nbytes = len(data)
    nblocks = (nbytes + blocksize - 1) // blocksize
    flags = {0: 0, 1: 0x01, 2: 0x04}[shuffle] | (0 if split else 0x10) | (code << 5)
    body, starts = b"", []
    for b in range(nblocks):
        blk = data[b * blocksize:(b + 1) * blocksize]
        n = len(blk) // typesize
        if shuffle == 1 and typesize > 1:
            blk = np.frombuffer(blk, np.uint8, n * typesize).reshape(n, typesize).T.tobytes() + blk[n * typesize:]
        elif shuffle == 2:
            n8 = n // 8 * 8
            if n8:
                e = np.frombuffer(blk, np.uint8, n8 * typesize).reshape(n8, typesize, 1)
                bits = np.unpackbits(e, axis=-1, bitorder="little")           # [element][byte][bit]
                rows = np.packbits(bits.transpose(1, 2, 0), axis=-1, bitorder="little")  # [byte][bit][n/8]
                blk = rows.tobytes() + blk[n8 * typesize:]
        starts.append(16 + 4 * nblocks + len(body))
        full = len(blk) == blocksize
        nsplits = typesize if (split and full and typesize <= 16 and blocksize // typesize >= 128) else 1
        ne = len(blk) // nsplits
        for j in range(nsplits):
            part = blk[j * ne:(j + 1) * ne]
            c = comp(part)
            if len(c) >= len(part):
                c = part  # incompressible stream: stored, flagged by its size
            body += struct.pack("<i", len(c)) + c
    head = bytes([2, 1, flags, typesize]) + struct.pack("<III", nbytes, blocksize, 16 + 4 * nblocks + len(body))
    return head + struct.pack(f"<{nblocks}i", *starts) + body


@pytest.mark.parametrize("shuffle,split,cname", [(2, False, "zstd"), (1, True, "zstd"), (0, False, "zlib"), (2, True, "zlib")])
def test_blosc_chunks_decode(shuffle, split, cname):
    from medical_image_generation_amd import volume_io as vio
    rng = np.random.default_rng(shuffle)
    for n, typesize, blocksize in [(1000, 4, 1024), (4099, 4, 2048), (13, 4, 4096), (700, 2, 512), (257, 1, 256), (5000, 8, 8192)]:
        smooth = (np.cumsum(rng.integers(-3, 4, n)) % 251).astype({1: np.uint8, 2: np.uint16, 4: np.float32, 8: np.float64}[typesize])
        data = smooth.tobytes() + bytes(rng.integers(0, 255, rng.integers(0, typesize)).astype(np.uint8))  # ragged tail
        assert vio.blosc_decompress(_blosc_frame(data, typesize, shuffle, blocksize, cname, split)) == data
    stored = bytes([2, 1, 0x02, 4]) + np.array([8, 8, 24], "<u4").tobytes() + b"abcdefgh"
    assert vio.blosc_decompress(stored) == b"abcdefgh"
    with pytest.raises(ValueError):
        vio.blosc_decompress(b"\x02\x01")


def test_preprocessed_case_is_read_like_load_image(tmp_path):
    """<id>.zarr/image with chunks (1, 1, H, W), Blosc(zstd, bitshuffle) (configuration.py:1403-1412) + <id>.pkl; .npy / .npz fallbacks
    (DATA:535-561)."""
    import json
    import pickle

    from medical_image_generation_amd import volume_io as vio
    rng = np.random.default_rng(0)
    vol = rng.uniform(0, 1, (2, 5, 12, 9)).astype(np.float32)
    arr = tmp_path / "caseA.zarr" / "image"
    arr.mkdir(parents=True)
    (tmp_path / "caseA.zarr" / ".zgroup").write_text(json.dumps({"zarr_format": 2}))
    chunks = (1, 1, 8, 9)  # H not a multiple of the chunk: edge chunks are stored whole, padded with the fill value
    (arr / ".zarray").write_text(json.dumps({"zarr_format": 2, "shape": list(vol.shape), "chunks": list(chunks), "dtype": "<f4", "order": "C",
                                            "fill_value": 0.0, "filters": None,
                                            "compressor": {"id": "blosc", "cname": "zstd", "clevel": 5, "shuffle": 2, "blocksize": 0}}))
    for c in range(2):
        for z in range(5):
            for hb in range(2):
                if (c, z, hb) == (1, 3, 1):
                    continue  # a chunk that was never written reads as fill_value
                tile = np.zeros(chunks, np.float32)
                part = vol[c, z, hb * 8:(hb + 1) * 8]
                tile[0, 0, :part.shape[0]] = part
                (arr / f"{c}.{z}.{hb}.0").write_bytes(_blosc_frame(tile.tobytes(), 4, 2, 128))
    locs = {1: np.array([[0, 1, 2, 3], [1, 4, 11, 8]]), 2: np.zeros((0, 4), np.int64)}
    with open(tmp_path / "caseA.pkl", "wb") as f:
        pickle.dump({"class_locations": locs, "spacing": [1.0, 0.5, 0.5]}, f)
    image, props = vio.load_image(str(tmp_path), "caseA")
    want = vol.copy()
    want[1, 3, 8:] = 0
    assert image.dtype == np.float32 and np.array_equal(image, want)
    assert np.array_equal(props["class_locations"][1], locs[1]) and props["spacing"] == [1.0, 0.5, 0.5]
    np.save(tmp_path / "caseB.npy", vol)
    np.savez_compressed(tmp_path / "caseC.npz", data=vol)
    assert np.array_equal(vio.load_image(str(tmp_path), "caseB")[0], vol) and np.array_equal(vio.load_image(str(tmp_path), "caseC")[0], vol)
    (tmp_path / "caseD.b2nd").write_bytes(b"")
    with pytest.raises(NotImplementedError):
        vio.load_image(str(tmp_path), "caseD")
    with pytest.raises(FileNotFoundError):
        vio.load_image(str(tmp_path), "caseE")

    class Evil:
        def __reduce__(self):
            return (print, ("constructed from a properties file",))
    with open(tmp_path / "caseB.pkl", "wb") as f:
        pickle.dump({"x": Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        vio.load_image(str(tmp_path), "caseB")
