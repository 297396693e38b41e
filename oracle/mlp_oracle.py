"""CPU restatement of the fully-connected GAN extension.  TEST INFRASTRUCTURE ONLY (see oracle/siggan_oracle.py's header).

PARITY UNPINNED: the reference contains no fully-connected Generator / Discriminator (generator_vanilla_gan.py:106-107 and
discriminator_vanilla_gan.py:121-122 reject everything but the 64 / 128 conv models; no test, fixture or golden vector of
the reference covers an MLP), so this file is the build's own definition of BASELINE.json's configs[0] model, restated with
torch-CPU functional ops -- it checks that the HIP path computes what include/siggan_mlp.h says, not that it matches the
reference.  The step structure, loss, label smoothing and Adam follow the reference's conv path
(vanilla_gan_model.py:180-306, restated in siggan_oracle.py), from which bce / adam_update are reused."""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .siggan_oracle import AdamState, BN_EPS, BN_MOMENTUM, bce

Tensor = torch.Tensor


def g_forward(sd: Dict[str, Tensor], z: Tensor, hidden, size: int, training: bool) -> Tensor:
    x = z
    for i in range(len(hidden)):
        x = F.linear(x, sd[f"net.{i}.linear.weight"], sd[f"net.{i}.linear.bias"])
        if training:
            sd[f"net.{i}.bn.num_batches_tracked"] += 1
        x = F.batch_norm(x, sd[f"net.{i}.bn.running_mean"], sd[f"net.{i}.bn.running_var"], sd[f"net.{i}.bn.weight"],
                         sd[f"net.{i}.bn.bias"], training, BN_MOMENTUM, BN_EPS)
        x = F.relu(x)
    x = torch.tanh(F.linear(x, sd["out.weight"], sd["out.bias"]))
    return x.view(-1, 1, size, size)


def d_forward(sd: Dict[str, Tensor], x: Tensor, n_hidden: int, slope: float = 0.2) -> Tensor:
    x = x.flatten(1)
    for j in range(n_hidden):
        x = F.leaky_relu(F.linear(x, sd[f"net.{j}.weight"], sd[f"net.{j}.bias"]), slope)
    return torch.sigmoid(F.linear(x, sd["out.weight"], sd["out.bias"]))


def _leafs(sd, names):
    return {k: sd[k].detach().clone().requires_grad_(True) for k in names}


def d_step(g_sd, d_sd, d_opt: AdamState, real, z, hidden, size, lr=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9):
    names = d_opt.names
    leaf = _leafs(d_sd, names)
    with torch.no_grad():
        fake = g_forward(g_sd, z, hidden, size, training=False)
    rp, fp = d_forward(leaf, real, len(hidden)), d_forward(leaf, fake, len(hidden))
    lr_, lf_ = bce(rp, label_smoothing), bce(fp, 0.0)
    loss = lr_ + lf_
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, gl)}
    d_opt.apply(d_sd, grads, lr, beta1, beta2)
    return {"d_loss": float(loss.detach()), "d_loss_real": float(lr_.detach()), "d_loss_fake": float(lf_.detach()),
            "d_real_mean": float(rp.detach().mean()), "d_fake_mean": float(fp.detach().mean())}, grads


def g_step(g_sd, d_sd, g_opt: AdamState, z, hidden, size, lr=2e-4, beta1=0.5, beta2=0.999):
    names = g_opt.names
    leaf = dict(g_sd)
    leaf.update(_leafs(g_sd, names))
    fake = g_forward(leaf, z, hidden, size, training=True)
    for k in g_sd:
        if k not in names:
            g_sd[k] = leaf[k]
    fp = d_forward(d_sd, fake, len(hidden))
    loss = bce(fp, 1.0)
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, gl)}
    g_opt.apply(g_sd, grads, lr, beta1, beta2)
    return {"g_loss": float(loss.detach()), "g_fake_mean": float(fp.detach().mean())}, grads
