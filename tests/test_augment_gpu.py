"""Input pipeline on the device (through the C ABI): k_augment against Pillow / the oracle, bit for bit, and the
loader end to end (decode cache, epoch plan, sharding)."""
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
from signature_gan_amd import _lib  # noqa: E402
from signature_gan_amd import data_loader_signatures as DL  # noqa: E402
from test_augment_cpu import pil_chain  # noqa: E402

pytestmark = pytest.mark.gpu


def _run_kernel(cache, index, prm, tab, lut, size, augment):
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    c, i, l = torch.from_numpy(cache).to(dev), torch.from_numpy(index.astype(np.int32)).to(dev), lut.to(dev)
    p = torch.from_numpy(prm).to(dev) if prm is not None else None
    t = torch.from_numpy(tab).to(dev) if tab is not None else None
    out = torch.empty(len(index), 1, size, size, device=dev)
    _lib.check(lib.siggan_augment_batch(0, c.data_ptr(), c.shape[0], i.data_ptr(), p.data_ptr() if p is not None else None,
                                        t.data_ptr() if t is not None else None, l.data_ptr(), out.data_ptr(), len(index), size,
                                        1 if augment else 0, 255, torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("size", [64, 128])
def test_kernel_matches_pillow_bit_for_bit(size):
    rng = np.random.default_rng(size)
    n_img, b = 37, 96
    cache = rng.integers(0, 256, (n_img, size, size), dtype=np.uint8)
    index = rng.integers(0, n_img, b)
    angle = rng.uniform(-5, 5, b).astype(np.float32).astype(np.float64)
    scale = rng.uniform(0.9, 1.1, b).astype(np.float32).astype(np.float64)
    flip = rng.random(b) < 0.5
    angle[:3] = [0.0, 360.0, 1e-15]
    scale[3:5] = [1.0, np.nan]
    prm, tab = DL.build_params(angle, scale, flip, size)
    lut = DL.normalize_lut((-1.0, 1.0))
    got = _run_kernel(cache, index, prm, tab, lut, size, True)
    for k in range(b):
        want = pil_chain(cache[index[k]], float(angle[k]), None if math.isnan(scale[k]) else float(scale[k]), bool(flip[k]))
        assert torch.equal(got[k, 0], A.to_normalized(want)), (k, angle[k], scale[k], flip[k])
        if k < 8:                                           # and the oracle's own restatement
            assert np.array_equal(A.augment_image(cache[index[k]], float(angle[k]),
                                                  None if math.isnan(scale[k]) else float(scale[k]), bool(flip[k])), want)
    plain = _run_kernel(cache, index, None, None, lut, size, False)
    for k in range(b):
        assert torch.equal(plain[k, 0], A.to_normalized(cache[index[k]]))


def test_bad_arguments_are_rejected():
    lib = _lib.load()
    with pytest.raises(ValueError):
        _lib.check(lib.siggan_augment_batch(0, None, 1, None, None, None, None, None, 1, 64, 0, 255, None))


def _make_folder(tmp_path, n, rng):
    for i in range(n):
        h, w = int(rng.integers(40, 200)), int(rng.integers(40, 260))
        arr = rng.integers(0, 256, (h, w), dtype=np.uint8)
        Image.fromarray(arr, "L").save(tmp_path / f"sig_{i:03d}.{'png' if i % 3 else 'bmp'}")
    (tmp_path / "broken.png").write_bytes(b"not an image")
    (tmp_path / "notes.txt").write_text("ignored")


@pytest.mark.parametrize("workers,size", [(4, 64), (0, 128)])
def test_loader_epochs_match_the_reference_chain(tmp_path, workers, size):
    rng = np.random.default_rng(5)
    _make_folder(tmp_path, 23, rng)
    bs = 4
    loader = DL.create_data_loader(tmp_path, batch_size=bs, num_workers=workers, image_size=size, horizontal_flip=True)
    ds = loader.dataset
    assert len(ds) == 24 and len(loader) == 24 // bs and loader.batch_size == bs
    decoded = [ds.decode(i, size) for i in range(len(ds))]
    assert sum(d is None for d in decoded) == 1
    for epoch in range(2):
        torch.manual_seed(77 + epoch)
        got = [b.cpu() for b in loader]
        torch.manual_seed(77 + epoch)
        plan = A.epoch_plan(len(ds), bs, workers, True, True, 5.0, (0.9, 1.1), True)
        assert len(got) == len(plan)
        for g, (idx, ang, sc, fl) in zip(got, plan):
            assert tuple(g.shape) == (bs, 1, size, size) and g.dtype == torch.float32
            for k, i in enumerate(idx):
                if decoded[i] is None:                       # unreadable file: all-zero tensor (:135-138)
                    assert not g[k].any()
                else:
                    assert torch.equal(g[k, 0], A.to_normalized(pil_chain(decoded[i], ang[k], sc[k], fl[k])))


def test_validation_loader_and_sharding(tmp_path):
    rng = np.random.default_rng(9)
    _make_folder(tmp_path, 17, rng)
    train, val = DL.create_train_val_loaders(tmp_path, batch_size=4, num_workers=2, image_size=64, val_split=0.25, seed=42)
    perm = torch.randperm(18, generator=torch.Generator().manual_seed(42)).tolist()
    assert train.indices == perm[:14] and val.indices == perm[14:]
    vb = [b.cpu() for b in val]
    assert [b.shape[0] for b in vb] == [4]
    for k, i in enumerate(val.indices):
        d = val.dataset.decode(i, 64)
        assert torch.equal(vb[0][k, 0], A.to_normalized(d)) if d is not None else not vb[0][k].any()
    torch.manual_seed(5)
    full = [b.cpu() for b in DL.create_data_loader(tmp_path, batch_size=4, num_workers=2)]
    for rank in range(2):
        torch.manual_seed(5)
        part = [b.cpu() for b in DL.create_data_loader(tmp_path, batch_size=4, num_workers=2, rank=rank, world_size=2)]
        assert len(part) == len(full)
        for f, p in zip(full, part):
            assert torch.equal(p, f[2 * rank:2 * rank + 2])
    assert tuple(DL.get_sample_batch(val, 3).shape) == (3, 1, 64, 64)
