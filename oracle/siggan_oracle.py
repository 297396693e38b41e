"""CPU oracle for the signature-GAN hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, not the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under
``signature-gan_amd/`` imports, links or executes anything in ``oracle/``.

It is an independent CPU restatement (fp32, torch-CPU functional ops) of the algorithm the
reference runs on its G+D train-step / generation path.  The arithmetic of the reference
lives in PyTorch itself (``requirements.txt:5`` pins only ``torch>=2.0.0``), so the oracle is
"this restatement x torch CPU"; it is PINNED against outputs of the reference itself, produced
in the build container by importing ``/root/reference/src`` (``tests/golden/make_golden.py``,
fixtures committed under ``tests/golden/``; torch version recorded in every fixture).

Reference lines followed (all under /root/reference/src):
  * Generator  : generator_vanilla_gan.py:124-163 (layers), :189-209 (forward)
  * Discriminator: discriminator_vanilla_gan.py:51-75 (block), :131-207 (layers), :241-274
  * BCE / labels / D step / G step: vanilla_gan_model.py:107, :152-178, :180-252, :254-306
  * clipping + trainer variants of the steps: train_vanilla_gan_signatures.py:262-376
  * Adam hyper-parameters: vanilla_gan_model.py:110-120 (torch.optim.Adam defaults otherwise)

State is held in plain dicts keyed exactly like the reference's ``state_dict()``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

BN_MOMENTUM = 0.1   # torch.nn.BatchNorm default (generator_vanilla_gan.py:58,126 use defaults)
BN_EPS = 1e-5
ADAM_EPS = 1e-8     # torch.optim.Adam default (vanilla_gan_model.py:110-120 pass lr/betas only)

# generator_vanilla_gan.py:131-149 -- channel chain of the ConvTranspose2d blocks
G_CHAIN = {64: (256, 128, 64, 32, 32), 128: (512, 256, 128, 64, 32, 32)}
# discriminator_vanilla_gan.py:131-194 -- channel chain of the Conv2d blocks
D_CHAIN = {64: (64, 128, 256, 512), 128: (64, 128, 256, 512, 512)}


# --------------------------------------------------------------------------------------
# layouts (names/shapes in parameters() / state_dict() order)
# --------------------------------------------------------------------------------------
def g_state_specs(latent_dim: int, size: int, channels: int = 1) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """state_dict entries of Generator: name -> (shape, kind) with kind in
    {'param','buffer','counter'} (generator_vanilla_gan.py:124-163)."""
    if size not in G_CHAIN:
        raise ValueError(f"output_size must be 64 or 128, got {size}")   # :106-107
    chain = G_CHAIN[size]
    feat = chain[0] * 16
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    s["fc.0.weight"] = ((feat, latent_dim), "param")
    s["fc.0.bias"] = ((feat,), "param")
    s["fc.1.weight"] = ((feat,), "param")
    s["fc.1.bias"] = ((feat,), "param")
    s["fc.1.running_mean"] = ((feat,), "buffer")
    s["fc.1.running_var"] = ((feat,), "buffer")
    s["fc.1.num_batches_tracked"] = ((), "counter")
    for i in range(len(chain) - 1):
        cin, cout = chain[i], chain[i + 1]
        p = f"upsample_blocks.{i}.block."
        s[p + "0.weight"] = ((cin, cout, 4, 4), "param")       # ConvTranspose2d layout (in,out,kh,kw)
        s[p + "1.weight"] = ((cout,), "param")
        s[p + "1.bias"] = ((cout,), "param")
        s[p + "1.running_mean"] = ((cout,), "buffer")
        s[p + "1.running_var"] = ((cout,), "buffer")
        s[p + "1.num_batches_tracked"] = ((), "counter")
    s["final_conv.0.weight"] = ((channels, chain[-1], 3, 3), "param")
    s["final_conv.0.bias"] = ((channels,), "param")
    return s


def d_state_specs(size: int, channels: int = 1) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """state_dict entries of Discriminator (discriminator_vanilla_gan.py:131-207)."""
    if size not in D_CHAIN:
        raise ValueError(f"input_size must be 64 or 128, got {size}")    # :121-122
    chain = (channels,) + D_CHAIN[size]
    s: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()
    for i in range(len(chain) - 1):
        p = f"conv_blocks.{i}.block.0."
        s[p + "weight"] = ((chain[i + 1], chain[i], 4, 4), "param")
        s[p + "bias"] = ((chain[i + 1],), "param")
    s["classifier.0.weight"] = ((1, chain[-1] * 16), "param")
    s["classifier.0.bias"] = ((1,), "param")
    return s


def param_names(specs) -> List[str]:
    return [k for k, (_, kind) in specs.items() if kind == "param"]


# --------------------------------------------------------------------------------------
# forward passes
# --------------------------------------------------------------------------------------
class _ActWithGivenSign(torch.autograd.Function):
    """(Leaky)ReLU whose BACKWARD uses a supplied ``positive`` mask instead of its own x > 0.

    Two correct fp32 implementations disagree on the sign of a pre-activation that is within
    rounding (~1e-7 of the layer's scale) of zero; at batch 64 a step has ~1e7 activations, so
    about one such coin-flip per step is expected, and it moves several gradient sums by far more
    than 1e-3.  Parity tests therefore hand the OTHER implementation's sign decisions to the
    oracle (and separately bound how many disagree and how close to zero those are)."""

    @staticmethod
    def forward(ctx, x, positive, slope):
        ctx.save_for_backward(positive)
        ctx.slope = slope
        return torch.where(x > 0, x, x * slope)

    @staticmethod
    def backward(ctx, g):
        (positive,) = ctx.saved_tensors
        return g * torch.where(positive, torch.ones_like(g), torch.full_like(g, ctx.slope)), None, None


def _act(x: Tensor, slope: float, given, record: Optional[list]):
    """``given``: None (own decisions), a full bool tensor (another implementation's decisions), or a census
    ``(flat_index, positive)`` of another run's decisions on its near-zero elements (tests/golden/make_golden.py::ActTap):
    own decisions everywhere else -- away from zero every correct implementation decides alike."""
    if record is not None:
        record.append(x.detach())
    if given is None:
        return F.leaky_relu(x, slope) if slope != 0.0 else F.relu(x)
    if isinstance(given, tuple):
        idx, pos = given
        positive = (x.detach() > 0).reshape(-1).clone()
        positive[idx] = pos
        given = positive
    return _ActWithGivenSign.apply(x, given.reshape(x.shape), slope)


class _RoundFwd(torch.autograd.Function):
    """value rounded to a narrow float type, gradient passed through: a tensor the HIP path STORES in that type."""

    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _RoundBwd(torch.autograd.Function):
    """identity whose GRADIENT is rounded to a narrow float type (times a power-of-two scale, as the fp16 chains carry
    it): a gradient tensor the HIP path stores in that type."""

    @staticmethod
    def forward(ctx, x, dtype, scale):
        ctx.dtype, ctx.scale = dtype, scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (g * ctx.scale).to(ctx.dtype).to(torch.float32) / ctx.scale, None, None


class Quant:
    """Storage-rounding model of the build-defined bf16 / f16 variants (BASELINE.json configs[2] / configs[4]; the
    reference itself is fp32-only, vanilla_gan_model.py:107-120).  The arithmetic stays fp32; what is rounded to
    ``dtype`` is exactly what the HIP path keeps in 16 bits: ``a`` -- a stored activation, ``g`` -- a stored activation
    gradient (marks the tensor whose incoming gradient is stored), ``w`` -- the weight copy an MFMA convolution reads
    (the gradient flows to the fp32 master weight unchanged).  With a Quant the oracle predicts the narrow path up to
    fp32 summation order; without one it is the fp32 reference arithmetic the narrow path is compared against at a
    stated looser tolerance."""

    def __init__(self, dtype: torch.dtype, grad_scale: float = 1.0):
        self.dtype, self.grad_scale = dtype, float(grad_scale)

    def a(self, x):
        return _RoundFwd.apply(x, self.dtype)

    def g(self, x):
        return _RoundBwd.apply(x, self.dtype, self.grad_scale)

    def w(self, w):
        return _RoundFwd.apply(w, self.dtype)


class _NoQuant:
    a = g = w = staticmethod(lambda x: x)


_NOQ = _NoQuant()


def g_forward(sd: Dict[str, Tensor], z: Tensor, training: bool, size: int,
              signs: Optional[Sequence[Tensor]] = None, record: Optional[list] = None,
              q: Optional[Quant] = None) -> Tensor:
    """Generator.forward (generator_vanilla_gan.py:189-209).  ``training`` selects the
    BatchNorm mode; in training mode running stats / num_batches_tracked in ``sd`` are
    updated in place exactly as nn.BatchNorm does (momentum 0.1, unbiased running var).
    ``signs`` / ``record``: see _ActWithGivenSign (one entry per ReLU, NCHW bool)."""
    chain = G_CHAIN[size]
    sg = (lambda i: None) if signs is None else (lambda i: signs[i])
    q = q or _NOQ
    # the pre-BatchNorm tensor is kept (for the backward pass) only in training mode; in eval mode BatchNorm is folded
    # into the producing kernel's epilogue and only the activation is stored
    qy = (lambda t: q.g(q.a(t))) if training else (lambda t: t)
    n_blocks = len(chain) - 1

    def bn(x, prefix):
        if training:
            sd[prefix + "num_batches_tracked"] += 1
        return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"],
                            sd[prefix + "weight"], sd[prefix + "bias"],
                            training, BN_MOMENTUM, BN_EPS)

    x = qy(F.linear(z, sd["fc.0.weight"], sd["fc.0.bias"]))
    x = q.g(q.a(_act(bn(x, "fc.1."), 0.0, sg(0), record)))
    x = x.view(-1, chain[0], 4, 4)
    for i in range(n_blocks):
        p = f"upsample_blocks.{i}.block."
        x = qy(F.conv_transpose2d(x, q.w(sd[p + "0.weight"]), None, stride=2, padding=1))
        x = _act(bn(x, p + "1."), 0.0, sg(i + 1), record)
        if i + 1 < n_blocks or not training:
            x = q.a(x)                  # (training: the last block's activation is re-derived from y by both of its readers, never stored)
        if i + 1 < n_blocks:            # the last block's activation gradient is consumed where it is formed, never stored
            x = q.g(x)
    x = F.conv2d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"], stride=1, padding=1)
    return torch.tanh(x)


def d_features(sd: Dict[str, Tensor], x: Tensor, size: int,
               masks: Optional[Sequence[Tensor]] = None,
               dropout: float = 0.25, slope: float = 0.2,
               signs: Optional[Sequence[Tensor]] = None, record: Optional[list] = None,
               q: Optional[Quant] = None) -> Tensor:
    """Discriminator.forward_features (discriminator_vanilla_gan.py:262-274).  ``masks`` is a
    list of (B, C_i) keep masks (1 keep / 0 drop), one per block, standing in for the hidden
    RNG of nn.Dropout2d (:74-75); None means eval mode (dropout off)."""
    n_blocks = len(D_CHAIN[size])
    q = q or _NOQ
    for i in range(n_blocks):
        p = f"conv_blocks.{i}.block.0."
        w = sd[p + "weight"] if i == 0 else q.w(sd[p + "weight"])      # block 1 (Cin = 1) is not an MFMA kernel: fp32 weights
        x = q.g(F.conv2d(x, w, sd[p + "bias"], stride=2, padding=1))   # stored gradient: d(pre-activation)
        x = _act(x, slope, None if signs is None else signs[i], record)
        if masks is not None and dropout > 0:
            noise = masks[i].to(x.dtype) / (1.0 - dropout)     # bernoulli(1-p).div_(1-p)
            x = x * noise[:, :, None, None]
        x = q.a(x)                                                       # stored activation: after LeakyReLU + dropout
    return x.flatten(1)


def d_forward(sd, x, size, masks=None, dropout=0.25, slope=0.2, signs=None, record=None, q=None) -> Tensor:
    """Discriminator.forward (:241-260): probabilities (B,1)."""
    f = d_features(sd, x, size, masks, dropout, slope, signs, record, q)
    return torch.sigmoid(F.linear(f, sd["classifier.0.weight"], sd["classifier.0.bias"]))


def bce(p: Tensor, y: float) -> Tensor:
    """nn.BCELoss(reduction='mean') on probabilities (vanilla_gan_model.py:107) -- the very torch operator the reference's
    criterion calls: log terms clamped at -100 in the forward pass, and a backward pass of its own,
    (p - y) / max((1 - p) * p, 1e-12) / count, which stays finite when a prediction saturates to exactly 0 or 1 (differentiating
    the clamped logarithms by autograd gives 0 * inf = NaN there: the third step of the 128x128 batch-4 sequence fixture,
    d_real_mean 0.9999993, is such a case)."""
    return F.binary_cross_entropy(p, torch.full_like(p, y))


# --------------------------------------------------------------------------------------
# optimiser pieces
# --------------------------------------------------------------------------------------
def clip_grad_norm(grads: Sequence[Tensor], max_norm: float) -> float:
    """nn.utils.clip_grad_norm_ (train_vanilla_gan_signatures.py:275-278): global L2 norm,
    scale by min(1, max_norm/(norm+1e-6)); returns the pre-clip norm."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2.0) for g in grads]), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return float(total)


def adam_update(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int,
                lr: float, beta1: float, beta2: float, eps: float = ADAM_EPS) -> None:
    """One torch.optim.Adam update (single-tensor form, weight_decay 0, amsgrad False);
    ``step`` is the step count AFTER the increment.  In place on p, m, v."""
    m.lerp_(g, 1.0 - beta1)                                  # m = b1*m + (1-b1)*g
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)


class AdamState:
    """exp_avg / exp_avg_sq / step per parameter, in parameters() order."""

    def __init__(self, names: Sequence[str], sd: Dict[str, Tensor]):
        self.names = list(names)
        self.m = {k: torch.zeros_like(sd[k]) for k in self.names}
        self.v = {k: torch.zeros_like(sd[k]) for k in self.names}
        self.step = 0

    def apply(self, sd, grads: Dict[str, Tensor], lr, beta1, beta2):
        self.step += 1
        for k in self.names:
            adam_update(sd[k], grads[k], self.m[k], self.v[k], self.step, lr, beta1, beta2)


# --------------------------------------------------------------------------------------
# the two training steps
# --------------------------------------------------------------------------------------
def _leafs(sd, names):
    return {k: sd[k].detach().clone().requires_grad_(True) for k in names}


def d_grads(g_sd, d_sd, real: Tensor, z: Tensor, masks_real, masks_fake, size: int,
            label_smoothing: float = 0.9, dropout: float = 0.25, signs=None, record=None, q=None):
    """Forward/backward half of the D step (vanilla_gan_model.py:204-233 ==
    train_vanilla_gan_signatures.py:294-323): returns (metrics, grads, real_preds, fake_preds,
    fake_images).  G runs in eval mode under no_grad; only D parameters receive gradients."""
    names = param_names(d_state_specs(size, real.shape[1]))
    leaf = _leafs(d_sd, names)
    with torch.no_grad():
        fake = g_forward(g_sd, z, training=False, size=size, q=q)
    nb = len(D_CHAIN[size])
    s_real, s_fake = (None, None) if signs is None else (signs[:nb], signs[nb:])
    real_preds = d_forward(leaf, real, size, masks_real, dropout, signs=s_real, record=record, q=q)
    loss_real = bce(real_preds, label_smoothing)
    fake_preds = d_forward(leaf, fake, size, masks_fake, dropout, signs=s_fake, record=record, q=q)
    loss_fake = bce(fake_preds, 0.0)
    loss = loss_real + loss_fake
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, gl)}
    loss, loss_real, loss_fake = loss.detach(), loss_real.detach(), loss_fake.detach()
    real_preds, fake_preds = real_preds.detach(), fake_preds.detach()
    metrics = {
        "d_loss": float(loss), "d_loss_real": float(loss_real), "d_loss_fake": float(loss_fake),
        "d_real_mean": float(real_preds.mean()), "d_fake_mean": float(fake_preds.mean()),
        "d_real_acc": float((real_preds > 0.5).float().mean()),
        "d_fake_acc": float((fake_preds < 0.5).float().mean()),
    }
    return metrics, grads, real_preds.detach(), fake_preds.detach(), fake


def g_grads(g_sd, d_sd, z: Tensor, size: int, signs=None, record=None, q=None):
    """Forward/backward half of the G step (vanilla_gan_model.py:274-297 ==
    train_vanilla_gan_signatures.py:349-365): G in train mode (BN batch statistics, running
    stats updated in ``g_sd``), D in eval mode (dropout off), BCE against 1.0 (no smoothing)."""
    names = param_names(g_state_specs(z.shape[1], size))
    leaf = dict(g_sd)
    leaf.update(_leafs(g_sd, names))
    ng = len(G_CHAIN[size])
    s_g, s_d = (None, None) if signs is None else (signs[:ng], signs[ng:])
    fake = g_forward(leaf, z, training=True, size=size, signs=s_g, record=record, q=q)
    for k in g_sd:                      # running stats / counters were updated on the copies
        if k not in names:
            g_sd[k] = leaf[k]
    fake_preds = d_forward(d_sd, fake, size, None, signs=s_d, record=record, q=q)
    loss = bce(fake_preds, 1.0)
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, gl)}
    loss, fake_preds = loss.detach(), fake_preds.detach()
    metrics = {"g_loss": float(loss), "g_fake_mean": float(fake_preds.mean())}
    return metrics, grads, fake_preds.detach(), fake.detach()


def d_step(g_sd, d_sd, d_opt: AdamState, real, z, masks_real, masks_fake, size,
           lr=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9, clip: Optional[float] = None,
           dropout: float = 0.25, signs=None, record=None, q=None):
    metrics, grads, rp, fp, fake = d_grads(g_sd, d_sd, real, z, masks_real, masks_fake, size,
                                           label_smoothing, dropout, signs, record, q)
    metrics["d_grad_norm"] = clip_grad_norm(list(grads.values()), clip) if clip is not None else None
    d_opt.apply(d_sd, grads, lr, beta1, beta2)
    return metrics, grads


def g_step(g_sd, d_sd, g_opt: AdamState, z, size, lr=2e-4, beta1=0.5, beta2=0.999,
           clip: Optional[float] = None, signs=None, record=None, q=None):
    metrics, grads, fp, fake = g_grads(g_sd, d_sd, z, size, signs, record, q)
    metrics["g_grad_norm"] = clip_grad_norm(list(grads.values()), clip) if clip is not None else None
    g_opt.apply(g_sd, grads, lr, beta1, beta2)
    return metrics, grads


# --------------------------------------------------------------------------------------
# spectral normalisation (torch.nn.utils.spectral_norm on every D conv + the classifier,
# discriminator_vanilla_gan.py:60-62,200-202)
# --------------------------------------------------------------------------------------
SN_EPS = 1e-12


def sn_weights(d_sd, sn: Dict[str, Tensor], size: int, training: bool) -> Dict[str, Tensor]:
    """What torch's spectral-norm hook does before one Discriminator forward, for every conv and the classifier: in
    training mode one power iteration on the stored (u, v) -- v = normalize(W^T u), u = normalize(W v), in place, no
    gradient -- then in every mode sigma = u . (W v) and weight = weight_orig / sigma, differentiable w.r.t. weight_orig
    (u, v constants).  ``d_sd`` holds weight_orig under the plain '<layer>.weight' keys (leaf tensors when gradients are
    wanted); ``sn`` holds '<layer>.weight_u' / '<layer>.weight_v' and is updated in place.  Returns the dict to run the
    Discriminator with."""
    out = dict(d_sd)
    for k in [k for k in d_sd if k.endswith(".weight")]:
        base = k[:-len("weight")]
        w = d_sd[k]
        wm = w.reshape(w.shape[0], -1)
        u, v = sn[base + "weight_u"], sn[base + "weight_v"]
        if training:
            with torch.no_grad():
                v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=SN_EPS)
                u = F.normalize(torch.mv(wm, v), dim=0, eps=SN_EPS)
            sn[base + "weight_u"], sn[base + "weight_v"] = u, v
        sigma = torch.dot(u, torch.mv(wm, v))
        out[k] = w / sigma
    return out


def d_step_sn(g_sd, d_sd, sn, d_opt: AdamState, real, z, masks_real, masks_fake, size,
              lr=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9, dropout: float = 0.25, signs=None, record=None):
    """train_discriminator_step with Discriminator(use_spectral_norm=True) (vanilla_gan_model.py:180-252): D.train(), so the real
    and the fake forward each run a power iteration and see different effective weights; the gradient w.r.t. weight_orig goes
    through both sigmas."""
    names = param_names(d_state_specs(size, real.shape[1]))
    leaf = _leafs(d_sd, names)
    with torch.no_grad():
        fake = g_forward(g_sd, z, training=False, size=size)
    nb = len(D_CHAIN[size])
    s_real, s_fake = (None, None) if signs is None else (signs[:nb], signs[nb:])       # D(real) blocks, then D(fake) blocks
    real_preds = d_forward(sn_weights(leaf, sn, size, True), real, size, masks_real, dropout, signs=s_real, record=record)
    fake_preds = d_forward(sn_weights(leaf, sn, size, True), fake, size, masks_fake, dropout, signs=s_fake, record=record)
    loss_real, loss_fake = bce(real_preds, label_smoothing), bce(fake_preds, 0.0)
    loss = loss_real + loss_fake
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, gl)}
    d_opt.apply(d_sd, grads, lr, beta1, beta2)
    metrics = {"d_loss": float(loss.detach()), "d_loss_real": float(loss_real.detach()), "d_loss_fake": float(loss_fake.detach()),
               "d_real_mean": float(real_preds.detach().mean()), "d_fake_mean": float(fake_preds.detach().mean())}
    return metrics, grads


def g_step_sn(g_sd, d_sd, sn, g_opt: AdamState, z, size, lr=2e-4, beta1=0.5, beta2=0.999, signs=None, record=None):
    """train_generator_step against a spectral-norm Discriminator (vanilla_gan_model.py:254-306): D.eval() -- no power
    iteration, sigma from the stored (u, v)."""
    names = param_names(g_state_specs(z.shape[1], size))
    leaf = dict(g_sd)
    leaf.update(_leafs(g_sd, names))
    ng = len(G_CHAIN[size])
    s_g, s_d = (None, None) if signs is None else (signs[:ng], signs[ng:])             # fc, G blocks, then D blocks
    fake = g_forward(leaf, z, training=True, size=size, signs=s_g, record=record)
    for k in g_sd:
        if k not in names:
            g_sd[k] = leaf[k]
    fake_preds = d_forward(sn_weights(d_sd, sn, size, False), fake, size, None, signs=s_d, record=record)
    loss = bce(fake_preds, 1.0)
    gl = torch.autograd.grad(loss, [leaf[k] for k in names])
    grads = {k: g.detach() for k, g in zip(names, gl)}
    g_opt.apply(g_sd, grads, lr, beta1, beta2)
    return {"g_loss": float(loss.detach()), "g_fake_mean": float(fake_preds.detach().mean())}, grads


def ablation_step(g_sd, d_sd, g_opt: AdamState, d_opt: AdamState, real, z, masks_real, masks_fake, masks_g, size,
                  lr_g=2e-4, lr_d=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9, dropout: float = 0.25, q=None,
                  signs=None, record=None):
    """One iteration of AblationGANTrainer.train_epoch (ablation_vanilla_gan_signatures.py:397-467), for the standard
    (ReLU) Generator: both networks in train mode for the whole iteration; ONE Generator forward (BatchNorm batch
    statistics, running stats updated) whose detached image feeds the D update (:414-430) and through which the G update
    back-propagates (:432-441); the G update runs the UPDATED Discriminator, still in train mode (a third set of dropout
    masks), against the smoothed real label.  Returns (metrics, d_grads, g_grads).
    ``signs`` / ``record`` (see _act): dicts keyed 'g' (fc + Generator blocks), 'd_real', 'd_fake', 'd_g' (the Discriminator
    pass of the G update), each a per-layer list -- the harness calls D(real) BEFORE the Generator, this restatement after it,
    so the groups are named rather than positional."""
    g_names = param_names(g_state_specs(z.shape[1], size))
    d_names = param_names(d_state_specs(size, real.shape[1]))
    g_leaf = dict(g_sd)
    g_leaf.update(_leafs(g_sd, g_names))
    sg = (lambda key: None) if signs is None else (lambda key: signs[key])
    rc = (lambda key: None) if record is None else (lambda key: record.setdefault(key, []))
    fake = g_forward(g_leaf, z, training=True, size=size, q=q, signs=sg("g"), record=rc("g"))
    for k in g_sd:
        if k not in g_names:
            g_sd[k] = g_leaf[k]
    d_leaf = _leafs(d_sd, d_names)
    real_preds = d_forward(d_leaf, real, size, masks_real, dropout, q=q, signs=sg("d_real"), record=rc("d_real"))
    fake_preds = d_forward(d_leaf, fake.detach(), size, masks_fake, dropout, q=q, signs=sg("d_fake"), record=rc("d_fake"))
    loss_real, loss_fake = bce(real_preds, label_smoothing), bce(fake_preds, 0.0)
    d_loss = loss_real + loss_fake
    gl = torch.autograd.grad(d_loss, [d_leaf[k] for k in d_names])
    d_grads = {k: g.detach() for k, g in zip(d_names, gl)}
    d_opt.apply(d_sd, d_grads, lr_d, beta1, beta2)
    preds_g = d_forward(d_sd, fake, size, masks_g, dropout, q=q, signs=sg("d_g"), record=rc("d_g"))   # the updated D, dropout still active
    g_loss = bce(preds_g, label_smoothing)
    gl = torch.autograd.grad(g_loss, [g_leaf[k] for k in g_names])
    g_grads = {k: g.detach() for k, g in zip(g_names, gl)}
    g_opt.apply(g_sd, g_grads, lr_g, beta1, beta2)
    metrics = {"d_loss": float(d_loss.detach()), "d_loss_real": float(loss_real.detach()), "d_loss_fake": float(loss_fake.detach()),
               "d_real_mean": float(real_preds.detach().mean()), "d_fake_mean": float(fake_preds.detach().mean()),
               "g_loss": float(g_loss.detach()), "g_fake_mean": float(preds_g.detach().mean())}
    return metrics, d_grads, g_grads


def average_grads(per_rank: Sequence[Dict[str, Tensor]]) -> Dict[str, Tensor]:
    """Data-parallel emulation: mean of the replicas' gradients."""
    out = {}
    for k in per_rank[0]:
        out[k] = torch.stack([g[k] for g in per_rank]).mean(0)
    return out


# --------------------------------------------------------------------------------------
# image post-processing used by the generation callers (utils/inference.py:106-134)
# --------------------------------------------------------------------------------------
def to_uint8(img: Tensor) -> Tensor:
    """((x + 1) * 127.5).clip(0, 255).astype(uint8) -- truncation, not rounding."""
    return ((img + 1.0) * 127.5).clamp(0, 255).to(torch.uint8)
