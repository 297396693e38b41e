"""CPU restatement of the reference's input pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (signature-gan_amd/data_loader_signatures.py + the HIP kernel k_augment) never does.

What it restates (data_loader_signatures.py in the reference):
  * SignatureDataset.__getitem__ (:107-138): PIL open -> convert('L') -> transform
  * get_train_transforms (:153-218): Resize((S,S)) -> RandomRotation(+-deg, fill=255) -> RandomAffine(degrees=0,
    scale=(lo,hi), fill=255) -> [RandomHorizontalFlip] -> ToTensor -> Normalize(0.5, 0.5)
  * create_data_loader (:244-321): DataLoader(shuffle, num_workers, drop_last) -- index order and the per-worker RNG
    streams the random transforms draw from

The arithmetic lives in third-party libraries that are NOT under /root/reference:
  * torchvision (requirements.txt:6 ">=0.15.0", absent from this image): RandomRotation.get_params / RandomAffine.get_params
    draw order, functional.rotate -> PIL Image.rotate, functional.affine -> _get_inverse_affine_matrix + PIL
    Image.transform(AFFINE), both with NEAREST (the default interpolation) -- restated from its published source;
    PARITY UNPINNED for this glue (no torchvision here, no fixture in the reference).
  * Pillow (present, 12.2.0): Image.rotate / Image.transform nearest-neighbour affine (Geometry.c: ImagingScaleAffine for
    axis-aligned matrices, 16.16 fixed point affine_fixed otherwise) -- PINNED: tests/test_augment_cpu.py checks this
    restatement bit-for-bit against Pillow itself.
  * torch.utils.data (present): RandomSampler / worker seeding -- PINNED against the real DataLoader in the same test.
"""
import math

import numpy as np
import torch


# ---- Pillow: Image.rotate's matrix (Image.py, rotate()) --------------------------------------------------------------
def pil_rotate_matrix(w, h, angle):
    angle = angle % 360.0
    if angle == 0:
        return None                                   # rotate() returns self.copy()
    cx, cy = w / 2.0, h / 2.0
    a = -math.radians(angle)
    m = [round(math.cos(a), 15), round(math.sin(a), 15), 0.0, round(-math.sin(a), 15), round(math.cos(a), 15), 0.0]
    m[2] = m[0] * -cx + m[1] * -cy + m[2]
    m[5] = m[3] * -cx + m[4] * -cy + m[5]
    m[2] += cx
    m[5] += cy
    return m


# ---- torchvision: functional._get_inverse_affine_matrix (inverted=True) ----------------------------------------------
def tv_inverse_affine_matrix(center, angle, translate, scale, shear):
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [x / scale for x in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


# ---- Pillow: nearest-neighbour affine on an 8-bit image (Geometry.c) -------------------------------------------------
def affine_nearest(img, a, fill=255):
    """img: (H, W) uint8; a: output pixel (x, y) samples input (a0 x + a1 y + a2, a3 x + a4 y + a5)."""
    h, w = img.shape
    out = np.full_like(img, fill)
    if a[1] == 0 and a[3] == 0:                       # ImagingScaleAffine: running double sums, COORD() truncation
        def coord(v):
            return -1 if v < 0.0 else int(v)
        xo = a[2] + a[0] * 0.5
        yo = a[5] + a[4] * 0.5
        xt, xmin, xmax = [0] * w, w, 0
        for x in range(w):
            xin = coord(xo)
            if 0 <= xin < w:
                xmax = x + 1
                xmin = min(xmin, x)
                xt[x] = xin
            xo += a[0]
        for y in range(h):
            yi = coord(yo)
            if 0 <= yi < h:
                for x in range(xmin, xmax):
                    out[y, x] = img[yi, xt[x]]
            yo += a[4]
        return out
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))          # affine_fixed: 16.16 fixed point
    a0, a1, a3, a4 = fix(a[0]), fix(a[1]), fix(a[3]), fix(a[4])
    a2 = fix(a[2] + a[0] * 0.5 + a[1] * 0.5)
    a5 = fix(a[5] + a[3] * 0.5 + a[4] * 0.5)
    for y in range(h):
        xx, yy = a2, a5
        for x in range(w):
            xin = xx >> 16
            if 0 <= xin < w:
                yin = yy >> 16
                if 0 <= yin < h:
                    out[y, x] = img[yin, xin]
            xx += a0
            yy += a3
        a2 += a1
        a5 += a4
    return out


def augment_image(img, angle, scale, flip=False, fill=255):
    """RandomRotation(angle) -> RandomAffine(scale) -> [hflip] on one (S, S) uint8 image."""
    h, w = img.shape
    m = pil_rotate_matrix(w, h, angle)
    if m is not None:
        img = affine_nearest(img, m, fill)
    if scale is not None:                             # RandomAffine is only in the pipeline when scale_range != (1, 1)
        m = tv_inverse_affine_matrix([w * 0.5, h * 0.5], 0.0, [0, 0], scale, [0.0, 0.0])
        img = affine_nearest(img, m, fill)
    if flip:
        img = img[:, ::-1]
    return np.ascontiguousarray(img)


def to_normalized(img_u8):
    """ToTensor + Normalize(0.5, 0.5), with torch's own fp32 ops."""
    return torch.from_numpy(np.array(img_u8, dtype=np.uint8, copy=True)).to(torch.float32).div(255).sub_(0.5).div_(0.5)


# ---- torch.utils.data: index order and per-sample draws of one epoch -------------------------------------------------
def epoch_plan(n, batch_size, num_workers, shuffle, drop_last, rotation_degrees, scale_range, horizontal_flip):
    """[(indices, angles, scales, flips)] per batch, consuming the global torch RNG exactly as
    `iter(DataLoader(...))` does: base seed, sampler seed; worker w (seeded base + w) serves batches w, w + W, ..."""
    base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
        perm = torch.randperm(n, generator=g).tolist()
    else:
        perm = list(range(n))
    nb = n // batch_size if drop_last else (n + batch_size - 1) // batch_size
    gens = [torch.Generator().manual_seed(base_seed + w) for w in range(num_workers)]
    plan = []
    for b in range(nb):
        g = gens[b % num_workers] if num_workers > 0 else None
        idx = perm[b * batch_size:(b + 1) * batch_size]
        ang, sc, fl = [], [], []
        for _ in idx:
            ang.append(float(torch.empty(1).uniform_(-float(rotation_degrees), float(rotation_degrees), generator=g).item())
                       if rotation_degrees > 0 else 0.0)
            if tuple(scale_range) != (1.0, 1.0):
                torch.empty(1).uniform_(-0.0, 0.0, generator=g)                       # RandomAffine's angle draw, degrees=(0,0)
                sc.append(float(torch.empty(1).uniform_(scale_range[0], scale_range[1], generator=g).item()))
            else:
                sc.append(None)
            fl.append(bool(torch.rand(1, generator=g) < 0.5) if horizontal_flip else False)
        plan.append((idx, ang, sc, fl))
    return plan
