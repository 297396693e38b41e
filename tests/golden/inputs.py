"""Deterministic synthetic inputs shared by the fixture generator and the parity tests.

Everything here is DATA GENERATION only (numpy Philox counter-based PRNG, bit-identical on any
box with the same numpy): network states, Adam moments, latent vectors, real batches and probe
indices.  The golden fixtures store only the reference's OUTPUTS for these inputs, so the 15 MB
of weights never has to be committed; both sides regenerate them from (seed, stream).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

W_STD = 0.05          # "trained-like" spread (reference init is 0.02; larger exercises sigmoid/tanh)


def rng(seed: int, stream) -> np.random.Generator:
    if isinstance(stream, str):
        stream = zlib.crc32(stream.encode())
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def gen_state(specs, seed: int) -> "OrderedDict[str, np.ndarray]":
    """specs: OrderedDict name -> (shape, kind) (state_dict order).  Non-trivial values for
    every entry so that biases, BN affine terms and running statistics all matter."""
    out = OrderedDict()
    for name, (shape, kind) in specs.items():
        g = rng(seed, name)
        if kind == "counter":
            out[name] = np.array(7, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = (0.1 * g.standard_normal(shape)).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = g.uniform(0.5, 1.5, shape).astype(np.float32)
        elif ".1.weight" in name:                      # BatchNorm gamma
            out[name] = (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
        elif ".1.bias" in name:                        # BatchNorm beta
            out[name] = (0.1 * g.standard_normal(shape)).astype(np.float32)
        else:                                          # conv / linear weight and bias
            out[name] = (W_STD * g.standard_normal(shape)).astype(np.float32)
    return out


def gen_sn_state(d_specs, seed: int) -> "OrderedDict[str, np.ndarray]":
    """State of a spectral-norm Discriminator (torch.nn.utils.spectral_norm keys) from the plain specs: every
    '<m>.weight' becomes '<m>.weight_orig' (same values as gen_state would give '<m>.weight'), plus unit vectors
    '<m>.weight_u' (rows) and '<m>.weight_v' (columns of the weight viewed as a matrix)."""
    plain = gen_state(d_specs, seed)
    out = OrderedDict()
    for name, a in plain.items():
        if name.endswith(".weight"):
            base = name[:-len("weight")]
            out[base + "weight_orig"] = a
            for tag, n in (("weight_u", a.shape[0]), ("weight_v", int(np.prod(a.shape[1:])))):
                v = rng(seed, base + tag).standard_normal(n)
                out[base + tag] = (v / np.linalg.norm(v)).astype(np.float32)
        else:
            out[name] = a
    return out


def gen_adam(specs, seed: int):
    """Warm Adam moments for every 'param' entry: (exp_avg, exp_avg_sq, step)."""
    m, v = OrderedDict(), OrderedDict()
    for name, (shape, kind) in specs.items():
        if kind != "param":
            continue
        g = rng(seed, "adam:" + name)
        m[name] = (1e-3 * g.standard_normal(shape)).astype(np.float32)
        v[name] = (1e-6 * (0.5 + g.uniform(0.0, 1.0, shape))).astype(np.float32)
    return m, v, 5


def gen_z(batch: int, latent: int, seed: int) -> np.ndarray:
    return rng(seed, "z").standard_normal((batch, latent)).astype(np.float32)


def gen_real(batch: int, size: int, seed: int, channels: int = 1) -> np.ndarray:
    return rng(seed, "real").uniform(-1.0, 1.0, (batch, channels, size, size)).astype(np.float32)


def gen_masks(batch: int, chans, seed: int, keep: float = 0.75):
    """Dropout2d keep masks (B, C_i) per block -- used where the test, not torch, draws them."""
    return [(rng(seed, f"mask{i}").uniform(0.0, 1.0, (batch, c)) < keep).astype(np.float32)
            for i, c in enumerate(chans)]


def probe_idx(numel: int, name: str, n: int = 64) -> np.ndarray:
    """Fixed probe positions into a flattened tensor."""
    if numel <= n:
        return np.arange(numel, dtype=np.int64)
    return np.sort(rng(12345, "probe:" + name).choice(numel, size=n, replace=False)).astype(np.int64)


def pack_masks(masks) -> np.ndarray:
    return np.concatenate([np.packbits(m.astype(np.uint8).ravel()) for m in masks])


def unpack_masks(packed: np.ndarray, batch: int, chans):
    out, off = [], 0
    for c in chans:
        nbytes = (batch * c + 7) // 8
        bits = np.unpackbits(packed[off:off + nbytes])[: batch * c]
        out.append(bits.reshape(batch, c).astype(np.float32))
        off += nbytes
    return out
