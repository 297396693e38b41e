"""Input pipeline of the signature GAN on the device (drop-in for the reference's data_loader_signatures.py).

The reference decodes, resizes and augments every image on CPU workers each epoch
(data_loader_signatures.py:107-138, 153-218, 244-321: PIL open -> 'L' -> Resize -> RandomRotation(+-5 deg, fill 255)
-> RandomAffine(scale 0.9-1.1, fill 255) -> ToTensor -> Normalize(0.5, 0.5), 4 workers) -- ~10^2 images/s per core
against ~3*10^4 images/s of the train step.  Here the images are decoded and resized ONCE into an (N, S, S) uint8
cache resident in HBM; every batch is then one kernel launch (`siggan_augment_batch`, csrc/ops.hip k_augment) that
gathers, rotates, scales, flips and normalises straight into the (B, 1, S, S) fp32 tensor the step consumes.

What is kept from the reference, bit for bit:
  * the order of samples and the per-sample random numbers: `iter(loader)` consumes torch's global RNG exactly like
    `iter(DataLoader(..., shuffle, num_workers))` (base seed, RandomSampler seed) and draws each sample's angle / scale /
    flip from the generator of the worker that would have served its batch (seed base + worker id, batches dealt
    round-robin).  Pinned against the real torch DataLoader in tests/test_augment_cpu.py.
  * the pixels: both resampling stages are Pillow's nearest-neighbour affine transforms (torchvision's default
    interpolation for RandomRotation / RandomAffine); the host tabulates Pillow's own arithmetic (16.16 fixed point for
    the rotation, running double sums for the axis-aligned scale) and the kernel applies it.  Pinned against Pillow.
  * ToTensor + Normalize: a 256-entry table computed with torch's own fp32 ops.
torchvision itself is not installed in the build image: its glue (draw order, matrix formulas) follows its published
source and is the one unpinned piece (oracle/augment_oracle.py header).

There is no CPU fallback: the loader needs a ROCm device and the HIP library.
"""
import logging
import math
import os
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

logger = logging.getLogger(__name__)

DEFAULT_BATCH_SIZE: int = 64
DEFAULT_NUM_WORKERS: int = 4
DEFAULT_IMAGE_SIZE: int = 64
DEFAULT_VAL_SPLIT: float = 0.1
SUPPORTED_EXTENSIONS: Tuple[str, ...] = ('.png', '.jpg', '.jpeg', '.bmp', '.tiff')
FILL = 255                                              # white paper (data_loader_signatures.py:181,191)


@dataclass
class TransformSpec:
    """What get_train_transforms / get_val_transforms describe; applied on the device by the loader."""
    image_size: int = DEFAULT_IMAGE_SIZE
    rotation_degrees: float = 0.0
    scale_range: Tuple[float, float] = (1.0, 1.0)
    horizontal_flip: bool = False
    normalize_range: Tuple[float, float] = (-1.0, 1.0)
    augment: bool = False

    def __call__(self, image):
        raise NotImplementedError("transforms run on the device, per batch: iterate the loader from create_data_loader()")


def get_train_transforms(image_size: int = DEFAULT_IMAGE_SIZE, rotation_degrees: float = 5.0,
                         scale_range: Tuple[float, float] = (0.9, 1.1), horizontal_flip: bool = False,
                         normalize_range: Tuple[float, float] = (-1.0, 1.0)) -> TransformSpec:
    """data_loader_signatures.py:153-218."""
    return TransformSpec(image_size, float(rotation_degrees), (float(scale_range[0]), float(scale_range[1])),
                         bool(horizontal_flip), tuple(normalize_range), True)


def get_val_transforms(image_size: int = DEFAULT_IMAGE_SIZE,
                       normalize_range: Tuple[float, float] = (-1.0, 1.0)) -> TransformSpec:
    """data_loader_signatures.py:221-243."""
    return TransformSpec(image_size, 0.0, (1.0, 1.0), False, tuple(normalize_range), False)


class SignatureDataset:
    """File list of the reference's SignatureDataset (:42-150): same extensions, same sorted order."""

    def __init__(self, root_dir: Union[str, Path], transform: Optional[TransformSpec] = None,
                 extensions: Tuple[str, ...] = SUPPORTED_EXTENSIONS):
        self.root_dir = Path(root_dir)
        self.transform = transform
        self.extensions = extensions
        if not self.root_dir.exists():
            raise ValueError(f"Directory does not exist: {root_dir}")
        paths = []
        for ext in extensions:
            paths.extend(self.root_dir.glob(f'*{ext}'))
            paths.extend(self.root_dir.glob(f'*{ext.upper()}'))
        self.image_paths: List[Path] = sorted(set(paths))
        if not self.image_paths:
            logger.warning(f"No images found in {root_dir}")
        else:
            logger.info(f"Found {len(self.image_paths)} images in {root_dir}")

    def __len__(self) -> int:
        return len(self.image_paths)

    def get_image_path(self, idx: int) -> Path:
        return self.image_paths[idx]

    def decode(self, idx: int, image_size: int) -> Optional[np.ndarray]:
        """PIL open -> 'L' -> Resize((S, S)) (bilinear, torchvision's default) as uint8; None if the file is unreadable
        (the reference then yields an all-zero tensor, :135-138)."""
        from PIL import Image
        try:
            im = Image.open(self.image_paths[idx]).convert('L').resize((image_size, image_size), Image.BILINEAR)
            return np.asarray(im, dtype=np.uint8)
        except Exception as e:                                  # noqa: BLE001 -- mirrors the reference's catch-all
            logger.error(f"Error loading image {self.image_paths[idx]}: {e}")
            return None

    def __getitem__(self, idx: int) -> torch.Tensor:
        """Un-augmented sample (ToTensor of the decoded image); batches come from the loader."""
        if idx >= len(self.image_paths):
            raise IndexError(f"Index {idx} out of range for dataset of size {len(self)}")
        from PIL import Image
        try:
            arr = np.asarray(Image.open(self.image_paths[idx]).convert('L'), dtype=np.uint8)
            return torch.from_numpy(arr.copy()).to(torch.float32).div(255).unsqueeze(0)
        except Exception as e:                                  # noqa: BLE001
            logger.error(f"Error loading image {self.image_paths[idx]}: {e}")
            return torch.zeros(1, DEFAULT_IMAGE_SIZE, DEFAULT_IMAGE_SIZE)


# ------------------------------------------------------------------------------------------------------------------
# host logic (pure CPU, no device): the epoch plan and the per-sample resampling parameters
# ------------------------------------------------------------------------------------------------------------------
def _uniform_many(g: Optional[torch.Generator], lo: Sequence[float], hi: Sequence[float], reps: int) -> np.ndarray:
    """`reps` rounds of `float(torch.empty(1).uniform_(lo[j], hi[j]).item())` for j = 0..len(lo)-1, drawn in that
    order from generator g (None: the global one), as one call.  torch's CPU uniform_ takes one 32-bit output x per
    element, u = (x & (2^24 - 1)) * 2^-24, and returns float32(u * (hi - lo) + lo) with the product and sum in double
    and (hi - lo) in float32 (ATen uniform_real_distribution<float>); a (0, 1) draw returns u itself."""
    k = len(lo)
    u = torch.empty(reps * k, dtype=torch.float32).uniform_(0.0, 1.0, generator=g).numpy().astype(np.float64).reshape(reps, k)
    lo32, hi32 = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    span = (hi32 - lo32).astype(np.float64)
    return (u * span + lo32.astype(np.float64)).astype(np.float32).astype(np.float64)


def plan_epoch(n: int, batch_size: int, num_workers: int, shuffle: bool, drop_last: bool, spec: TransformSpec):
    """Sample order and random transform parameters of one pass over n samples -- the values
    `for batch in DataLoader(dataset, batch_size, shuffle, num_workers, drop_last)` would produce with the reference's
    transform chain, consuming torch's global RNG the same way (see module docstring).
    Returns (index[nb][bs'] list of int lists, angle, scale, flip) with the last three as flat float64 / bool arrays in
    batch order; scale is NaN where the chain has no RandomAffine."""
    base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        perm = torch.randperm(n, generator=g).tolist()
    else:
        perm = list(range(n))
    nb = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
    batches = [perm[b * batch_size:(b + 1) * batch_size] for b in range(nb)]
    total = sum(len(b) for b in batches)
    angle, scale, flip = np.zeros(total), np.full(total, np.nan), np.zeros(total, bool)
    if not spec.augment:
        return batches, angle, scale, flip
    lo, hi, kinds = [], [], []
    if spec.rotation_degrees > 0:                       # RandomRotation.get_params: one uniform(-d, d)
        lo.append(-spec.rotation_degrees); hi.append(spec.rotation_degrees); kinds.append("angle")
    if tuple(spec.scale_range) != (1.0, 1.0):           # RandomAffine.get_params(degrees=(0,0), scale): angle draw, scale draw
        lo.append(-0.0); hi.append(0.0); kinds.append("skip")
        lo.append(spec.scale_range[0]); hi.append(spec.scale_range[1]); kinds.append("scale")
    if spec.horizontal_flip:                            # RandomHorizontalFlip: torch.rand(1) < p
        lo.append(0.0); hi.append(1.0); kinds.append("flip")
    if not kinds:
        return batches, angle, scale, flip
    starts = np.cumsum([0] + [len(b) for b in batches])
    workers = max(num_workers, 1)
    gens = [torch.Generator().manual_seed(base_seed + w) for w in range(num_workers)] if num_workers > 0 else [None]
    for w in range(workers):
        mine = list(range(w, nb, workers))
        cnt = sum(len(batches[b]) for b in mine)
        if cnt == 0:
            continue
        draws = _uniform_many(gens[w], lo, hi, cnt)
        pos = 0
        for b in mine:
            m = len(batches[b])
            d = draws[pos:pos + m]
            pos += m
            sl = slice(starts[b], starts[b] + m)
            for j, kd in enumerate(kinds):
                if kd == "angle":
                    angle[sl] = d[:, j]
                elif kd == "scale":
                    scale[sl] = d[:, j]
                elif kd == "flip":
                    flip[sl] = d[:, j] < 0.5
    return batches, angle, scale, flip


def _scale_tables(m0, m2, m4, m5, size):
    """Pillow's ImagingScaleAffine source positions for output columns / rows 0..size-1 (vectorised over samples):
    xo = a2 + a0/2, then xo += a0 per column, COORD(v) = v < 0 ? -1 : (int)v, valid if < size."""
    n = len(m0)
    xt, yt = np.full((n, size), -1, np.int16), np.full((n, size), -1, np.int16)
    xo, yo = m2 + m0 * 0.5, m5 + m4 * 0.5
    for k in range(size):
        xi = np.where(xo < 0.0, -1.0, np.trunc(np.minimum(xo, 1e9))).astype(np.int64)
        yi = np.where(yo < 0.0, -1.0, np.trunc(np.minimum(yo, 1e9))).astype(np.int64)
        xt[:, k] = np.where((xi >= 0) & (xi < size), xi, -1)
        yt[:, k] = np.where((yi >= 0) & (yi < size), yi, -1)
        xo = xo + m0
        yo = yo + m4
    return xt, yt


def build_params(angle: np.ndarray, scale: np.ndarray, flip: np.ndarray, size: int):
    """Per-sample kernel parameters: prm (n, 8) int32 and tab (n, 4, size) int16 (layout in include/siggan.h)."""
    n = len(angle)
    prm = np.zeros((n, 8), np.int32)
    tab = np.full((n, 4, size), -1, np.int16)
    cx = cy = size / 2.0
    # ---- stage 1: PIL Image.rotate(angle, NEAREST, fillcolor) (torchvision F.rotate on a PIL image) -----------------
    rot = np.zeros((n, 6))
    for i in range(n):                                   # math.* and round(): exactly what Image.rotate evaluates
        ang = float(angle[i]) % 360.0
        if ang == 0.0:
            prm[i, 0] = 0
            continue
        a = -math.radians(ang)
        m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
        m[2] = m[0] * -cx + m[1] * -cy + m[2]
        m[5] = m[3] * -cx + m[4] * -cy + m[5]
        m[2] += cx
        m[5] += cy
        rot[i] = m
        prm[i, 0] = 2 if (m[1] == 0 and m[3] == 0) else 1
    fixed = prm[:, 0] == 1
    if fixed.any():                                      # affine_fixed: FIX(v) = floor(v * 65536 + 0.5)
        r = rot[fixed]
        fix = lambda v: np.floor(v * 65536.0 + 0.5).astype(np.int64)
        cols = [fix(r[:, 0]), fix(r[:, 1]), fix(r[:, 2] + r[:, 0] * 0.5 + r[:, 1] * 0.5),
                fix(r[:, 3]), fix(r[:, 4]), fix(r[:, 5] + r[:, 3] * 0.5 + r[:, 4] * 0.5)]
        prm[fixed, 1:7] = np.stack(cols, 1).astype(np.int32)
    axis = prm[:, 0] == 2
    if axis.any():
        r = rot[axis]
        tab[axis, 0], tab[axis, 1] = _scale_tables(r[:, 0], r[:, 2], r[:, 4], r[:, 5], size)
    # ---- stage 2: torchvision F.affine(angle=0, translate=(0,0), scale=s, shear=(0,0)) -> PIL transform(AFFINE, NEAREST)
    has = ~np.isnan(scale)
    if has.any():
        s = scale[has]
        # _get_inverse_affine_matrix with rot = shear = 0: [d, -b, 0, -c, a, 0] / s = [1/s, 0, 0, -0, 1/s, 0]
        m0 = 1.0 / s
        m4 = 1.0 / s
        m1 = 0.0 / s
        m3 = -0.0 / s
        m2 = 0.0 / s + (m0 * (-cx - 0) + m1 * (-cy - 0))
        m5 = 0.0 / s + (m3 * (-cx - 0) + m4 * (-cy - 0))
        m2 = m2 + cx
        m5 = m5 + cy
        tab[has, 2], tab[has, 3] = _scale_tables(m0, m2, m4, m5, size)
        prm[has, 7] |= 2
    prm[flip, 7] |= 1
    return prm, tab


def normalize_lut(normalize_range: Tuple[float, float]) -> torch.Tensor:
    """Byte -> ToTensor (/255) -> Normalize(0.5, 0.5) unless the range is (0, 1) (:199-216), in torch's fp32."""
    lut = torch.arange(256, dtype=torch.uint8).to(torch.float32).div(255)
    if tuple(normalize_range) != (0.0, 1.0):
        lut = lut.sub_(0.5).div_(0.5)
    return lut


# ------------------------------------------------------------------------------------------------------------------
# the device loader
# ------------------------------------------------------------------------------------------------------------------
class DeviceSignatureLoader:
    """Iterable over (B, 1, S, S) fp32 batches in HBM; `len`, `.dataset`, `.batch_size` as torch's DataLoader.
    rank / world_size: data-parallel sharding -- every rank plans the same global batches (same torch seed) and yields
    its contiguous shard of each (dp.shard_bounds)."""

    def __init__(self, dataset: SignatureDataset, indices: Optional[Sequence[int]], spec: TransformSpec, batch_size: int,
                 num_workers: int, shuffle: bool, drop_last: bool, device: Union[str, torch.device, None] = None,
                 rank: int = 0, world_size: int = 1):
        from . import _lib
        self.lib = _lib.load()                           # raises if the HIP library is missing
        self._check = _lib.check
        dev = torch.device(device if device is not None else "cuda")
        if dev.type != "cuda":
            raise RuntimeError("the input pipeline runs on a ROCm device (no CPU fallback); got device=%r" % (device,))
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if batch_size % world_size:
            raise ValueError(f"global batch {batch_size} is not divisible by world size {world_size}")
        self.device, self.dataset, self.spec = dev, dataset, spec
        self.indices = list(range(len(dataset))) if indices is None else list(indices)
        self.batch_size, self.num_workers, self.shuffle, self.drop_last = batch_size, num_workers, shuffle, drop_last
        self.rank, self.world_size = rank, world_size
        s = spec.image_size
        cache = np.zeros((max(len(self.indices), 1), s, s), np.uint8)
        failed = []
        for k, idx in enumerate(self.indices):           # decode + resize once
            arr = dataset.decode(idx, s)
            if arr is None:
                failed.append(k)
            else:
                cache[k] = arr
        self.cache = torch.from_numpy(cache).to(dev)
        self.failed = torch.zeros(len(self.indices) + 1, dtype=torch.bool)
        self.failed[failed] = True
        self.any_failed = bool(failed)
        self.lut = normalize_lut(spec.normalize_range).to(dev)

    def __len__(self) -> int:
        n = len(self.indices)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n, s, dev = len(self.indices), self.spec.image_size, self.device
        batches, angle, scale, flip = plan_epoch(n, self.batch_size, self.num_workers, self.shuffle, self.drop_last, self.spec)
        if not batches:
            return
        flat = np.concatenate([np.asarray(b, np.int32) for b in batches])
        index_dev = torch.from_numpy(flat).to(dev)
        if self.spec.augment:
            prm, tab = build_params(angle, scale, flip, s)
            prm_dev, tab_dev = torch.from_numpy(prm).to(dev), torch.from_numpy(tab).to(dev)
        else:
            prm_dev = tab_dev = None
        stream = torch.cuda.current_stream(dev).cuda_stream
        pos = 0
        for b in batches:
            m = len(b)
            lo, hi = (self.rank * m // self.world_size, (self.rank + 1) * m // self.world_size) if self.world_size > 1 else (0, m)
            cnt = hi - lo
            if cnt > 0:
                out = torch.empty(cnt, 1, s, s, dtype=torch.float32, device=dev)
                o = pos + lo
                self._check(self.lib.siggan_augment_batch(
                    dev.index, self.cache.data_ptr(), self.cache.shape[0], index_dev.data_ptr() + 4 * o,
                    prm_dev.data_ptr() + 32 * o if prm_dev is not None else None,
                    tab_dev.data_ptr() + 8 * s * o if tab_dev is not None else None,
                    self.lut.data_ptr(), out.data_ptr(), cnt, s, 1 if self.spec.augment else 0, FILL, stream))
                if self.any_failed:                       # unreadable files are all-zero tensors in the reference
                    bad = self.failed[torch.as_tensor(b[lo:hi])]
                    if bool(bad.any()):
                        out[bad.to(dev)] = 0.0
                yield out
            pos += m


def _device_of(kw):
    return kw.pop("device", None), kw.pop("rank", 0), kw.pop("world_size", 1)


def create_data_loader(data_dir: Union[str, Path], batch_size: int = DEFAULT_BATCH_SIZE, num_workers: int = DEFAULT_NUM_WORKERS,
                       image_size: int = DEFAULT_IMAGE_SIZE, shuffle: bool = True, augment: bool = True,
                       rotation_degrees: float = 5.0, scale_range: Tuple[float, float] = (0.9, 1.1),
                       horizontal_flip: bool = False, pin_memory: bool = True, drop_last: bool = True, **kw) -> DeviceSignatureLoader:
    """data_loader_signatures.py:246-321 (same arguments; pin_memory is moot, the batches are born in HBM).
    Extra keywords: device, rank, world_size."""
    device, rank, world = _device_of(kw)
    if kw:
        raise TypeError(f"unexpected arguments: {sorted(kw)}")
    spec = (get_train_transforms(image_size, rotation_degrees, scale_range, horizontal_flip) if augment
            else get_val_transforms(image_size))
    dataset = SignatureDataset(root_dir=data_dir, transform=spec)
    if os.name == 'nt' and num_workers > 0:
        num_workers = min(num_workers, 4)
    loader = DeviceSignatureLoader(dataset, None, spec, batch_size, num_workers, shuffle, drop_last, device, rank, world)
    logger.info(f"Created DataLoader: {len(dataset)} images, batch_size={batch_size}, workers={num_workers}")
    return loader


def create_train_val_loaders(data_dir: Union[str, Path], batch_size: int = DEFAULT_BATCH_SIZE,
                             num_workers: int = DEFAULT_NUM_WORKERS, image_size: int = DEFAULT_IMAGE_SIZE,
                             val_split: float = DEFAULT_VAL_SPLIT, rotation_degrees: float = 5.0,
                             scale_range: Tuple[float, float] = (0.9, 1.1), horizontal_flip: bool = False, seed: int = 42,
                             pin_memory: bool = True, **kw):
    """data_loader_signatures.py:324-409: randperm(seed) split, augmented + shuffled training loader (drop_last),
    plain validation loader."""
    device, rank, world = _device_of(kw)
    if kw:
        raise TypeError(f"unexpected arguments: {sorted(kw)}")
    full = SignatureDataset(root_dir=data_dir, transform=None)
    total = len(full)
    val_size = int(total * val_split)
    train_size = total - val_size
    perm = torch.randperm(total, generator=torch.Generator().manual_seed(seed)).tolist()
    train_spec = get_train_transforms(image_size, rotation_degrees, scale_range, horizontal_flip)
    val_spec = get_val_transforms(image_size)
    if os.name == 'nt' and num_workers > 0:
        num_workers = min(num_workers, 4)
    train = DeviceSignatureLoader(full, perm[:train_size], train_spec, batch_size, num_workers, True, True, device, rank, world)
    val = DeviceSignatureLoader(full, perm[train_size:], val_spec, batch_size, num_workers, False, False, device, rank, world)
    logger.info(f"Created train/val split: {train_size} training, {val_size} validation")
    return train, val


def get_sample_batch(data_loader, num_samples: Optional[int] = None) -> torch.Tensor:
    """data_loader_signatures.py:444-466: the first batch (optionally its first num_samples images)."""
    batch = next(iter(data_loader))
    return batch if num_samples is None else batch[:num_samples]
