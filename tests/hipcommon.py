"""Helpers for the GPU parity tests: put the synthetic states of tests/golden/inputs.py into an
Engine (the HIP path) exactly as common.oracle_states() puts them into the oracle."""
import numpy as np
import torch

from common import I, O, SEED

import signature_gan_amd                      # noqa: F401  (import shim for signature-gan_amd/)
from signature_gan_amd.engine import Engine


def load_engine_state(eng, size, latent, warm):
    gs, ds = O.g_state_specs(latent, size), O.d_state_specs(size)
    g_np, d_np = I.gen_state(gs, SEED["state_g"]), I.gen_state(ds, SEED["state_d"])
    gv, dv, bnv = eng.views("g"), eng.views("d"), eng.bn_views()
    for k, a in g_np.items():
        t = torch.from_numpy(np.asarray(a))
        (gv[k] if k in gv else bnv[k]).copy_(t)
    for k, a in d_np.items():
        dv[k].copy_(torch.from_numpy(a))
    for which, specs, seed in (("g", gs, SEED["adam_g"]), ("d", ds, SEED["adam_d"])):
        mv, vv = eng.views(which, "exp_avg"), eng.views(which, "exp_avg_sq")
        steps = getattr(eng, f"{which}_adam_steps")
        if warm:
            m, v, step = I.gen_adam(specs, seed)
            for k in m:
                mv[k].copy_(torch.from_numpy(m[k])); vv[k].copy_(torch.from_numpy(v[k]))
            steps.fill_(float(step))
        else:
            getattr(eng, f"{which}_exp_avg").zero_(); getattr(eng, f"{which}_exp_avg_sq").zero_(); steps.zero_()
    eng.g_grads.zero_(); eng.d_grads.zero_()
    eng.params_changed()
    return eng


def make_engine(size, latent, max_batch, warm=False, seed=0, dtype="f32"):
    eng = Engine(latent_dim=latent, image_size=size, max_batch=max_batch, device="cuda:0", seed=seed, dtype=dtype)
    return load_engine_state(eng, size, latent, warm)


def cuda(a):
    return torch.as_tensor(a).to("cuda:0")


# ---- sign decisions of the HIP path (borderline activations) -------------------------------------
def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def hip_signs_d(eng, size, batch, passes):
    """Sign decisions (activation > 0) the HIP path took in the Discriminator, per pass and
    block, NCHW bool -- handed to the oracle's backward (oracle._ActWithGivenSign).  passes == 1: the Discriminator pass
    of a G step (debug tensor 'd_a_g': wherever the library put those rows); 2: D(real), D(fake) of a D step."""
    out = []
    name = "d_a_g" if passes == 1 else "d_a"
    for p in range(passes):
        for l, c in enumerate(list(O.D_CHAIN[size]), start=1):
            h = size >> l
            a = eng.debug_tensor(name, l, (passes * batch, h, h, c))[p * batch:(p + 1) * batch]
            out.append(_nchw(a) > 0)
    return out


def hip_signs_g(eng, size, batch):
    chain = O.G_CHAIN[size]
    out = []
    for l, c in enumerate(chain):
        h = 4 << l
        a = _nchw(eng.debug_tensor("g_a", l, (batch, h, h, c)))
        out.append((a > 0).reshape(batch, -1) if l == 0 else a > 0)
    return out


def count_sign_flips(signs, recorded, keep=None):
    """Disagreements between the HIP sign decisions and the oracle's own; every one must be a
    pre-activation within rounding of zero (|x| <= 1e-5 of the layer's scale).  Their NUMBER is bounded too, in proportion to
    the activations compared: two fp32 implementations differ by ~1e-7 of a layer's scale, so of N inputs spread over that
    scale about 1e-7 N (times the density at zero) land on the other side -- 16, or one per million where that is more (the
    128x128 batch-32 steps compare 41 M activations; 12-17 were seen there across boxes, whose oracle runs with different
    thread counts)."""
    n, total = 0, 0
    for i, (s, x) in enumerate(zip(signs, recorded)):
        total += x.numel()
        bad = s.reshape(x.shape) != (x > 0)
        if keep is not None and keep[i] is not None:
            bad &= keep[i][:, :, None, None] > 0          # dropped planes carry no gradient
        if bad.any():
            assert float(x[bad].abs().max()) <= 1e-5 * float(x.abs().max()), "sign disagreement away from zero"
            n += int(bad.sum())
    assert n <= max(16, total // 1000000), f"{n} borderline sign decisions differ (of {total} activations)"
    return n
