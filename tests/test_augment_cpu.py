"""Input pipeline, CPU side: the oracle's restatement of Pillow's nearest-neighbour affine transforms is pinned
against Pillow itself, the epoch plan (sample order + per-sample random draws) against the real torch DataLoader,
and the product's host logic (plan_epoch / build_params / normalize_lut) against both.  The kernel's use of the
parameters is emulated here in numpy so that the whole chain is checked without a GPU; tests/test_augment_gpu.py
checks the kernel itself."""
import math
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import signature_gan_amd  # noqa: E402,F401
from oracle import augment_oracle as A  # noqa: E402
from signature_gan_amd import data_loader_signatures as DL  # noqa: E402


def pil_chain(img, angle, scale, flip):
    """The reference's chain on one resized image, by Pillow: RandomRotation -> RandomAffine(scale) -> hflip."""
    im = Image.fromarray(img, "L").rotate(angle, resample=Image.NEAREST, fillcolor=255)
    if scale is not None:
        s = img.shape[0]
        m = A.tv_inverse_affine_matrix([s * 0.5, s * 0.5], 0.0, [0, 0], scale, [0.0, 0.0])
        im = im.transform((s, s), Image.AFFINE, m, Image.NEAREST, fillcolor=255)
    out = np.asarray(im)
    return np.ascontiguousarray(out[:, ::-1]) if flip else out


def apply_params(img, prm, tab, fill=255):
    """numpy emulation of k_augment for one sample (include/siggan.h: siggan_augment_batch)."""
    s = img.shape[0]
    ys, xs = np.meshgrid(np.arange(s), np.arange(s), indexing="ij")
    if prm[7] & 1:
        xs = s - 1 - xs
    ok = np.ones((s, s), bool)
    if prm[7] & 2:
        xs, ys = tab[2][xs].astype(np.int64), tab[3][ys].astype(np.int64)
        ok = (xs >= 0) & (ys >= 0)
    xs, ys = np.where(ok, xs, 0), np.where(ok, ys, 0)
    if prm[0] == 0:
        xin, yin = xs, ys
    elif prm[0] == 1:
        a = prm.astype(np.int64)
        xin, yin = (a[3] + ys * a[2] + xs * a[1]) >> 16, (a[6] + ys * a[5] + xs * a[4]) >> 16
        ok &= (xin >= 0) & (xin < s) & (yin >= 0) & (yin < s)
    else:
        xin, yin = tab[0][xs].astype(np.int64), tab[1][ys].astype(np.int64)
        ok &= (xin >= 0) & (yin >= 0)
    out = np.full((s, s), fill, np.uint8)
    out[ok] = img[np.where(ok, yin, 0), np.where(ok, xin, 0)][ok]
    return out


@pytest.mark.parametrize("size", [64, 128])
def test_oracle_matches_pillow(size):
    rng = np.random.default_rng(size)
    for t in range(24):
        img = rng.integers(0, 256, (size, size), dtype=np.uint8)
        angle = [0.0, 5.0, -5.0, 1e-9][t] if t < 4 else float(np.float32(rng.uniform(-5, 5)))
        scale = [None, 1.0, 0.9, 1.1][t] if t < 4 else float(np.float32(rng.uniform(0.9, 1.1)))
        flip = bool(t & 1)
        assert np.array_equal(A.augment_image(img, angle, scale, flip), pil_chain(img, angle, scale, flip)), (angle, scale)


@pytest.mark.parametrize("size", [64, 128])
def test_host_parameters_reproduce_pillow(size):
    rng = np.random.default_rng(7 + size)
    n = 200
    angle = rng.uniform(-5, 5, n).astype(np.float32).astype(np.float64)
    scale = rng.uniform(0.9, 1.1, n).astype(np.float32).astype(np.float64)
    flip = rng.random(n) < 0.5
    angle[:3] = [0.0, 360.0, 1e-15]                     # copy / copy / rotation that rounds to the identity matrix
    scale[3:5] = [1.0, np.nan]                          # identity scale / chain without RandomAffine
    prm, tab = DL.build_params(angle, scale, flip, size)
    assert prm.shape == (n, 8) and tab.shape == (n, 4, size)
    assert prm[0, 0] == 0 and prm[1, 0] == 0 and prm[2, 0] == 2 and not prm[4, 7] & 2
    for i in range(n):
        img = rng.integers(0, 256, (size, size), dtype=np.uint8)
        want = pil_chain(img, float(angle[i]), None if math.isnan(scale[i]) else float(scale[i]), bool(flip[i]))
        assert np.array_equal(apply_params(img, prm[i], tab[i]), want), (i, angle[i], scale[i], flip[i])


def test_normalize_table_is_totensor_normalize():
    lut = DL.normalize_lut((-1.0, 1.0))
    img = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert torch.equal(lut[torch.from_numpy(img).long()], A.to_normalized(img))
    assert torch.equal(DL.normalize_lut((0.0, 1.0)), torch.arange(256).float().div(255))


class _Draws(torch.utils.data.Dataset):
    """What the reference's transform chain draws per sample (torchvision's get_params), without the images."""

    def __init__(self, n, flip):
        self.n, self.flip = n, flip

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        a = float(torch.empty(1).uniform_(-5.0, 5.0).item())
        torch.empty(1).uniform_(-0.0, 0.0)
        s = float(torch.empty(1).uniform_(0.9, 1.1).item())
        f = float(torch.rand(1) < 0.5) if self.flip else 0.0
        return torch.tensor([float(i), a, s, f], dtype=torch.float64)


@pytest.mark.parametrize("workers,flip,drop_last", [(2, False, True), (3, True, False), (0, False, True)])
def test_epoch_plan_is_the_dataloaders(workers, flip, drop_last):
    n, bs = 53, 8
    spec = DL.get_train_transforms(64, 5.0, (0.9, 1.1), flip)
    for epoch in range(2):
        torch.manual_seed(1000 + epoch)
        ref = [b.clone() for b in torch.utils.data.DataLoader(_Draws(n, flip), batch_size=bs, shuffle=True, num_workers=workers,
                                                               drop_last=drop_last)]
        torch.manual_seed(1000 + epoch)
        batches, angle, scale, fl = DL.plan_epoch(n, bs, workers, True, drop_last, spec)
        torch.manual_seed(1000 + epoch)
        plan = A.epoch_plan(n, bs, workers, True, drop_last, 5.0, (0.9, 1.1), flip)
        assert len(batches) == len(ref) == len(plan)
        pos = 0
        for b, r, o in zip(batches, ref, plan):
            m = len(b)
            assert b == [int(v) for v in r[:, 0]] == o[0]
            assert np.array_equal(angle[pos:pos + m], r[:, 1].numpy()) and np.array_equal(angle[pos:pos + m], np.asarray(o[1]))
            assert np.array_equal(scale[pos:pos + m], r[:, 2].numpy()) and np.array_equal(scale[pos:pos + m], np.asarray(o[2]))
            assert np.array_equal(fl[pos:pos + m].astype(float), r[:, 3].numpy())
            pos += m


def test_plan_without_augmentation_or_shuffle():
    torch.manual_seed(3)
    batches, angle, scale, flip = DL.plan_epoch(10, 4, 2, False, False, DL.get_val_transforms(64))
    assert batches == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9]] and not angle.any() and np.isnan(scale).all() and not flip.any()


def test_loader_refuses_cpu(tmp_path):
    Image.fromarray(np.zeros((8, 8), np.uint8), "L").save(tmp_path / "a.png")
    with pytest.raises(RuntimeError):
        DL.create_data_loader(tmp_path, batch_size=1, device="cpu")
    with pytest.raises(ValueError):
        DL.SignatureDataset(tmp_path / "missing")
    ds = DL.SignatureDataset(tmp_path)
    assert len(ds) == 1 and ds.get_image_path(0).name == "a.png" and tuple(ds[0].shape) == (1, 8, 8)
    with pytest.raises(NotImplementedError):
        DL.get_train_transforms()(None)
