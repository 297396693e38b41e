"""Shared helpers for the parity tests: build oracle-side states from the synthetic inputs and
read the golden fixtures (outputs of the reference, see tests/golden/make_golden.py)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(HERE, "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import inputs as I                                   # noqa: E402  (tests/golden/inputs.py)
from oracle import siggan_oracle as O                # noqa: E402  (the checker)

GOLDEN = os.path.join(HERE, "golden")
CASES = [(64, 100, 4), (64, 100, 64), (128, 128, 4), (128, 128, 32), (64, 100, 128),   # (64, 100, 128): BASELINE configs[3]
         (64, 100, 5),      # an odd batch: ragged row tiles in every GEMM, five samples per BatchNorm statistic
         (128, 128, 5),     # the same at 128x128 (five blocks per network, the 64-channel patch kernel)
         (64, 50, 8)]       # latent_dim % 4 != 0 (the ablation grid's z = 50): the Generator fc's generic (non-MFMA) kernels
SEED = dict(state_g=101, state_d=202, adam_g=303, adam_d=404, z=11, real=22)


def load_golden(size, batch, latent=None):
    z = "" if latent in (None, 100 if size == 64 else 128) else f"_z{latent}"
    f = np.load(os.path.join(GOLDEN, f"golden_s{size}{z}_b{batch}.npz"))
    meta = json.loads(str(f["meta"]))
    return f, meta


def d_chans(size):
    return list(O.D_CHAIN[size])


def oracle_states(size, latent, warm):
    gs, ds = O.g_state_specs(latent, size), O.d_state_specs(size)
    g_sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in I.gen_state(gs, SEED["state_g"]).items()}
    d_sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in I.gen_state(ds, SEED["state_d"]).items()}
    g_opt, d_opt = O.AdamState(O.param_names(gs), g_sd), O.AdamState(O.param_names(ds), d_sd)
    if warm:
        for opt, specs, seed in ((g_opt, gs, SEED["adam_g"]), (d_opt, ds, SEED["adam_d"])):
            m, v, step = I.gen_adam(specs, seed)
            opt.m = {k: torch.from_numpy(a).clone() for k, a in m.items()}
            opt.v = {k: torch.from_numpy(a).clone() for k, a in v.items()}
            opt.step = step
    return g_sd, d_sd, g_opt, d_opt


def masks_from(f, key, batch, size, passes):
    chans = d_chans(size) * passes
    ms = I.unpack_masks(f[key], batch, chans)
    return [torch.from_numpy(m) for m in ms]


def census_signs(f, step):
    """The reference run's decisions on its near-zero activation inputs (fixture record '<step>/census/*', step = 'dstep' |
    'gstep'; written by make_golden.ActTap) as the oracle's ``signs`` argument: one (flat NCHW index, positive) pair per
    activation layer in call order -- D step: D(real) blocks then D(fake) blocks; G step: fc, G blocks, then D blocks."""
    n = f[f"{step}/census/n"]
    idx, val = f[f"{step}/census/idx"], f[f"{step}/census/val"]
    out, o = [], 0
    for k in n:
        out.append((torch.from_numpy(idx[o:o + k].astype(np.int64)), torch.from_numpy(val[o:o + k] > 0)))
        o += int(k)
    return out


def ablation_groups(size, items):
    """Split a per-layer list in the ablation harness' CALL order -- D(real) blocks, Generator (fc + blocks), D(fake.detach())
    blocks, D(fake) blocks of the G update -- into the named groups oracle.ablation_step takes."""
    nb, ng = len(O.D_CHAIN[size]), len(O.G_CHAIN[size])
    return {"d_real": items[:nb], "g": items[nb:nb + ng], "d_fake": items[nb + ng:2 * nb + ng], "d_g": items[2 * nb + ng:]}


def flips_vs_census(f, step, signs, keep=None):
    """Decisions of another implementation (``signs``: full bool tensors per activation layer, NCHW order) that differ from the
    reference run's on the census elements: list of (layer, flat index, the reference's value / layer max).  ``keep``: per
    layer a (B, C) Dropout2d keep mask or None -- a decision inside a dropped plane carries no gradient and is not counted."""
    n, amax = f[f"{step}/census/n"], f[f"{step}/census/absmax"]
    idx, val = f[f"{step}/census/idx"], f[f"{step}/census/val"]
    out, o = [], 0
    for l, k in enumerate(n):
        i, v = idx[o:o + k].astype(np.int64), val[o:o + k]
        assert signs[l].numel() == int(f[f"{step}/census/numel"][l]), "activation layer order / shape differs from the fixture"
        mine = signs[l].reshape(-1).numpy()[i]
        for j in np.nonzero(mine != (v > 0))[0]:
            if keep is not None and keep[l] is not None and signs[l].dim() == 4:
                _, c, h, w = signs[l].shape
                if float(keep[l][int(i[j]) // (c * h * w), (int(i[j]) // (h * w)) % c]) == 0.0:
                    continue
            out.append((l, int(i[j]), float(v[j] / amax[l])))
        o += int(k)
    return out


def probe(t, name):
    a = t.detach().reshape(-1).cpu().numpy()
    return a[I.probe_idx(a.size, name)]


def assert_close(got, want, rtol, atol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    if not np.all(err <= tol):
        i = int(np.argmax(err - tol))
        raise AssertionError(f"{what}: max violation at {i}: got {got.flat[i]!r} want {want.flat[i]!r} "
                             f"(err {err.flat[i]:.3e}, tol {tol.flat[i]:.3e}); max err {err.max():.3e}")
