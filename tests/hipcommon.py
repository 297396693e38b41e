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


def make_engine(size, latent, max_batch, warm=False, seed=0):
    eng = Engine(latent_dim=latent, image_size=size, max_batch=max_batch, device="cuda:0", seed=seed)
    return load_engine_state(eng, size, latent, warm)


def cuda(a):
    return torch.as_tensor(a).to("cuda:0")
