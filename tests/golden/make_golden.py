#!/usr/bin/env python3
"""Generate the golden fixtures by running the REFERENCE itself (CPU, fp32) in the build
container:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's modules are imported from /root/reference/src (read-only, never copied); only
their OUTPUTS on the synthetic inputs of ``inputs.py`` are stored (probes, metrics, masks drawn by
torch's Dropout2d, structure manifests of the two checkpoint layouts).  The reference cannot
travel to the GPU box, the fixtures can.

Reference entry points exercised (all under /root/reference/src):
  generator_vanilla_gan.Generator.forward            :189-209
  discriminator_vanilla_gan.Discriminator.forward    :241-260, forward_features :262-274
  vanilla_gan_model.VanillaGAN.train_discriminator_step :180-252, train_generator_step :254-306,
      generate :338-371, save :433-474
  the trainer's clipped variants (train_vanilla_gan_signatures.py:281-376) are replayed call by
  call on the reference's own modules/optimisers with torch.nn.utils.clip_grad_norm_ (:275),
  because that file cannot be imported here (torchvision absent).
"""
import json
import os
import sys
import tempfile

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")

import numpy as np
import torch
import torch.nn as nn

import inputs as I
from vanilla_gan_model import VanillaGAN          # the reference


SEED_STATE_G, SEED_STATE_D, SEED_ADAM_G, SEED_ADAM_D = 101, 202, 303, 404
SEED_Z, SEED_REAL, SEED_TORCH = 11, 22, 33
CLIP = 0.05           # small enough that clipping is ACTIVE on these states (norms recorded)


def specs_from_module(mod):
    from collections import OrderedDict
    pnames = {k for k, _ in mod.named_parameters()}
    out = OrderedDict()
    for k, v in mod.state_dict().items():
        kind = "param" if k in pnames else ("counter" if v.dtype == torch.int64 else "buffer")
        out[k] = (tuple(v.shape), kind)
    return out


def load_state(mod, seed):
    specs = specs_from_module(mod)
    st = I.gen_state(specs, seed)
    mod.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()})
    return specs


def load_adam(opt, mod, specs, seed):
    m, v, step = I.gen_adam(specs, seed)
    names = [k for k, _ in mod.named_parameters()]
    sd = opt.state_dict()
    sd["state"] = {i: {"step": torch.tensor(float(step)),
                       "exp_avg": torch.from_numpy(m[k]).clone(),
                       "exp_avg_sq": torch.from_numpy(v[k]).clone()} for i, k in enumerate(names)}
    opt.load_state_dict(sd)


class MaskTap:
    """Observe (never alter) what nn.Dropout2d drew: keep[n,c] = plane not zeroed."""

    def __init__(self, disc):
        self.masks = []
        self.h = [m.register_forward_hook(self._hook) for m in disc.modules() if isinstance(m, nn.Dropout2d)]

    def _hook(self, mod, inp, out):
        if mod.training:
            self.masks.append((out.detach().abs().amax(dim=(2, 3)) > 0).float().numpy())

    def close(self):
        for h in self.h:
            h.remove()


# An activation input within this fraction of its layer's largest magnitude is "near zero".  Two fp32 implementations of
# these layers differ by ~1e-6 of the layer scale, and the parity tests fail any sign disagreement above 1e-5 of it, so 2e-5
# covers every decision that may legitimately differ (1e-4 would store five times as many entries for nothing).
CENSUS_REL = 2e-5


class ActTap:
    """Observe (never alter) the INPUT of the reference's activation modules (nn.ReLU generator_vanilla_gan.py:58-60,127;
    nn.LeakyReLU discriminator_vanilla_gan.py:66-69) with forward PRE-hooks -- the modules are inplace, the pre-hook sees the
    pre-activation -- and keep the census of its near-zero elements: flat (NCHW) index, value and the side of zero the
    reference put it on.  Two correct fp32 implementations can only disagree on the sign of such an element, so the census is
    what lets any other implementation be compared with THIS run decision for decision (tests: oracle(signs = census) must
    reproduce the fixture on any host; the HIP path's decisions are counted against it directly)."""

    def __init__(self, *nets):
        self.layers = []
        mods = [m for net in nets for m in net.modules() if isinstance(m, (nn.ReLU, nn.LeakyReLU))]
        self.h = [m.register_forward_pre_hook(self._hook) for m in mods]

    def _hook(self, mod, inp):
        flat = inp[0].detach().reshape(-1)
        amax = float(flat.abs().max())
        idx = torch.nonzero(flat.abs() <= CENSUS_REL * amax).reshape(-1)
        self.layers.append((idx.numpy().astype(np.int32), flat[idx].numpy().copy(), flat.numel(), amax))

    def close(self):
        for h in self.h:
            h.remove()

    def same_as(self, other):
        return len(self.layers) == len(other.layers) and all(
            np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:] for a, b in zip(self.layers, other.layers))

    def store(self, tag, out):
        out[f"{tag}/census/n"] = np.array([len(i) for i, _, _, _ in self.layers], np.int32)
        out[f"{tag}/census/numel"] = np.array([n for _, _, n, _ in self.layers], np.int64)
        out[f"{tag}/census/absmax"] = np.array([a for _, _, _, a in self.layers], np.float32)
        out[f"{tag}/census/idx"] = np.concatenate([i for i, _, _, _ in self.layers]) if self.layers else np.zeros(0, np.int32)
        out[f"{tag}/census/val"] = np.concatenate([v for _, v, _, _ in self.layers]) if self.layers else np.zeros(0, np.float32)


def probes(prefix, named, out):
    for k, t in named.items():
        a = t.detach().reshape(-1).numpy()
        out[f"{prefix}/{k}"] = a[I.probe_idx(a.size, k)].astype(a.dtype)


def fresh_model(size, latent, warm):
    torch.manual_seed(0)
    m = VanillaGAN(latent_dim=latent, image_size=size, image_channels=1, device="cpu")
    gs = load_state(m.generator, SEED_STATE_G)
    ds = load_state(m.discriminator, SEED_STATE_D)
    if warm:
        load_adam(m.g_optimizer, m.generator, gs, SEED_ADAM_G)
        load_adam(m.d_optimizer, m.discriminator, ds, SEED_ADAM_D)
    return m


def record_step(tag, model, net, opt, metrics, out, extra_buffers=False):
    for k, v in metrics.items():
        if v is not None:
            out[f"{tag}/metric/{k}"] = np.float32(v)
    params = dict(net.named_parameters())
    out[f"{tag}/grad_norm"] = np.array([float(p.grad.norm()) for p in params.values()], np.float32)
    probes(f"{tag}/grad", {k: p.grad for k, p in params.items()}, out)
    probes(f"{tag}/w", params, out)
    names = list(params)
    states = [opt.state[p] for p in params.values()]
    probes(f"{tag}/m", {names[i]: s["exp_avg"] for i, s in enumerate(states)}, out)
    probes(f"{tag}/v", {names[i]: s["exp_avg_sq"] for i, s in enumerate(states)}, out)
    out[f"{tag}/adam_step"] = np.float32(float(opt.state[next(iter(params.values()))]["step"]))
    if extra_buffers:
        bufs = {k: v for k, v in net.state_dict().items() if k not in params}
        probes(f"{tag}/buf", {k: v.float() for k, v in bufs.items()}, out)


def d_step_clipped(model, real, z, clip):
    """train_vanilla_gan_signatures.py:294-337 replayed on the reference's objects."""
    D, G = model.discriminator, model.generator
    D.train(); G.eval()
    B = real.size(0)
    model.d_optimizer.zero_grad()
    real_preds = D(real)
    d_loss_real = model.criterion(real_preds, torch.full((B, 1), model.label_smoothing))
    with torch.no_grad():
        fake = G(z)
    fake_preds = D(fake)
    d_loss_fake = model.criterion(fake_preds, torch.zeros(B, 1))
    d_loss = d_loss_real + d_loss_fake
    d_loss.backward()
    norm = float(nn.utils.clip_grad_norm_(D.parameters(), clip))
    model.d_optimizer.step()
    return {"d_loss": d_loss.item(), "d_loss_real": d_loss_real.item(), "d_loss_fake": d_loss_fake.item(),
            "d_real_mean": real_preds.mean().item(), "d_fake_mean": fake_preds.mean().item(),
            "d_grad_norm": norm}


def g_step_clipped(model, z, clip):
    """train_vanilla_gan_signatures.py:349-376 replayed on the reference's objects."""
    D, G = model.discriminator, model.generator
    G.train(); D.eval()
    model.g_optimizer.zero_grad()
    fake = G(z)
    fake_preds = D(fake)
    g_loss = model.criterion(fake_preds, torch.ones(z.size(0), 1))
    g_loss.backward()
    norm = float(nn.utils.clip_grad_norm_(G.parameters(), clip))
    model.g_optimizer.step()
    return {"g_loss": g_loss.item(), "g_fake_mean": fake_preds.mean().item(), "g_grad_norm": norm}


def manifest(obj):
    if isinstance(obj, torch.Tensor):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype).replace("torch.", "")}
    if isinstance(obj, dict):
        return {str(k): manifest(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        if len(obj) > 8 and all(isinstance(x, (int, float)) for x in obj):
            return {"list_of": type(obj[0]).__name__, "len": len(obj)}
        return [manifest(v) for v in obj]
    return type(obj).__name__


def make(size, latent, B, full_image):
    out = {}
    z = torch.from_numpy(I.gen_z(B, latent, SEED_Z))
    z2 = torch.from_numpy(I.gen_z(B, latent, SEED_Z + 1))
    real = torch.from_numpy(I.gen_real(B, size, SEED_REAL))
    chans = [c for c in ([64, 128, 256, 512] if size == 64 else [64, 128, 256, 512, 512])]

    # (i) G eval forward -------------------------------------------------------------
    m = fresh_model(size, latent, warm=False)
    img = m.generate(B, noise=z)
    flat = img.reshape(-1).numpy()
    if full_image:
        out["g_eval/img"] = img.numpy()
    out["g_eval/probe"] = flat[I.probe_idx(flat.size, "img", 256)]
    out["g_eval/stats"] = np.stack([img.mean((1, 2, 3)).numpy(), img.amin((1, 2, 3)).numpy(),
                                    img.amax((1, 2, 3)).numpy()])
    # (ii) G train forward (BN batch stats + running-stat update) ------------------------
    m.generator.train()
    with torch.no_grad():
        img = m.generator(z)
    flat = img.reshape(-1).numpy()
    out["g_train/probe"] = flat[I.probe_idx(flat.size, "img", 256)]
    bufs = {k: v.float() for k, v in m.generator.state_dict().items()
            if "running" in k or "num_batches" in k}
    probes("g_train/buf", bufs, out)
    # (iii) D eval forward ---------------------------------------------------------------
    m = fresh_model(size, latent, warm=False)
    m.discriminator.eval()
    with torch.no_grad():
        out["d_eval/probs"] = m.discriminator(real).reshape(-1).numpy()
        f = m.discriminator.forward_features(real).reshape(-1).numpy()
    out["d_eval/feat_probe"] = f[I.probe_idx(f.size, "feat", 256)]
    # (iv) D train forward with torch-drawn masks ----------------------------------------
    m.discriminator.train()
    tap = MaskTap(m.discriminator)
    torch.manual_seed(SEED_TORCH)
    with torch.no_grad():
        out["d_train/probs"] = m.discriminator(real).reshape(-1).numpy()
    tap.close()
    out["d_train/masks"] = I.pack_masks(tap.masks)
    assert len(tap.masks) == len(chans)

    # (v)-(vii) single steps from known states ------------------------------------------
    # (the three variants share weights and inputs, hence the forward pass and its census: stored once, equality asserted)
    census = {}
    for tag, warm, clip in (("warm", True, None), ("fresh", False, None), ("clip", True, CLIP)):
        m = fresh_model(size, latent, warm=warm)
        tap = MaskTap(m.discriminator)
        acts = ActTap(m.discriminator)                      # D(real) blocks, then D(fake) blocks
        torch.manual_seed(SEED_TORCH + 1)
        if clip is None:
            met = m.train_discriminator_step(real, noise=z)
        else:
            met = d_step_clipped(m, real, z, clip)
        tap.close(); acts.close()
        assert len(tap.masks) == 2 * len(chans) and len(acts.layers) == 2 * len(chans)
        out[f"dstep_{tag}/masks"] = I.pack_masks(tap.masks)
        if "dstep" not in census:
            census["dstep"] = acts
            acts.store("dstep", out)
        assert acts.same_as(census["dstep"]), "the D-step forward differs between the step variants"
        record_step(f"dstep_{tag}", m, m.discriminator, m.d_optimizer, met, out)

        m = fresh_model(size, latent, warm=warm)
        acts = ActTap(m.generator, m.discriminator)         # G: fc, blocks; then D blocks
        if clip is None:
            met = m.train_generator_step(B, noise=z2)
        else:
            met = g_step_clipped(m, z2, clip)
        acts.close()
        assert len(acts.layers) == 2 * len(chans) + 1
        if "gstep" not in census:
            census["gstep"] = acts
            acts.store("gstep", out)
        assert acts.same_as(census["gstep"]), "the G-step forward differs between the step variants"
        record_step(f"gstep_{tag}", m, m.generator, m.g_optimizer, met, out, extra_buffers=True)

    # (viii) 3-step sequence (D then G each step): metrics, and every half-step's census (seq3/d<s>, seq3/g<s>) so that the
    # chained steps can be compared decision for decision like the single ones
    m = fresh_model(size, latent, warm=True)
    tap = MaskTap(m.discriminator)
    torch.manual_seed(SEED_TORCH + 2)
    seq = []
    for s in range(3):
        zs = torch.from_numpy(I.gen_z(B, latent, 1000 + 2 * s))
        zg = torch.from_numpy(I.gen_z(B, latent, 1001 + 2 * s))
        acts = ActTap(m.discriminator)
        dm = m.train_discriminator_step(real, noise=zs)
        acts.close(); acts.store(f"seq3/d{s}", out)
        acts = ActTap(m.generator, m.discriminator)
        gm = m.train_generator_step(B, noise=zg)
        acts.close(); acts.store(f"seq3/g{s}", out)
        seq.append([dm["d_loss"], dm["d_loss_real"], dm["d_loss_fake"], dm["d_real_mean"],
                    dm["d_fake_mean"], gm["g_loss"], gm["g_fake_mean"]])
    tap.close()
    out["seq3/metrics"] = np.array(seq, np.float32)
    out["seq3/masks"] = I.pack_masks(tap.masks)

    out["meta"] = np.array(json.dumps({
        "torch": torch.__version__, "threads": torch.get_num_threads(), "size": size, "latent": latent,
        "batch": B, "clip": CLIP, "census_rel": CENSUS_REL,
        "seeds": dict(state_g=SEED_STATE_G, state_d=SEED_STATE_D, adam_g=SEED_ADAM_G, adam_d=SEED_ADAM_D,
                      z=SEED_Z, real=SEED_REAL, torch=SEED_TORCH)}))
    path = os.path.join(HERE, golden_name(size, latent, B))
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def golden_name(size, latent, B):
    """The latent size is part of the name only when it is not the model's default (100 at 64x64, 128 at 128x128)."""
    z = "" if latent == (100 if size == 64 else 128) else f"_z{latent}"
    return f"golden_s{size}{z}_b{B}.npz"


def make_spectral_norm():
    """(x) spectral-norm Discriminator, eval mode (no power iteration): probabilities, feature probe, state_dict keys."""
    from collections import OrderedDict
    from discriminator_vanilla_gan import Discriminator
    out = {}
    for size in (64, 128):
        d = Discriminator(input_size=size, use_spectral_norm=True).eval()
        specs = OrderedDict((k, (tuple(v.shape), "param")) for k, v in Discriminator(input_size=size).state_dict().items())
        state = I.gen_sn_state(specs, SEED_STATE_D)
        sd = d.state_dict()
        assert set(sd) == set(state), (sorted(set(sd) ^ set(state)))
        d.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        x = torch.from_numpy(I.gen_real(4, size, SEED_REAL))
        with torch.no_grad():
            p = d(x)
            f = d.forward_features(x)
        out[f"s{size}/probs"] = p.reshape(-1).numpy()
        ff = f.reshape(-1).numpy()
        out[f"s{size}/feat_probe"] = ff[I.probe_idx(ff.size, "feat", 256)]
        out[f"s{size}/keys"] = np.array(json.dumps([[k, list(v.shape)] for k, v in sd.items()]))
    out["meta"] = np.array(json.dumps({"torch": torch.__version__, "seed_state_d": SEED_STATE_D, "seed_real": SEED_REAL}))
    path = os.path.join(HERE, "golden_spectral_norm.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_spectral_norm_steps():
    """(xii) training with a spectral-norm Discriminator: VanillaGAN(use_spectral_norm=True).train_discriminator_step /
    train_generator_step (vanilla_gan_model.py:180-306) on the reference itself -- metrics, gradient / moment / weight probes
    (parameter names with weight_orig), and the weight_u / weight_v buffers after each step."""
    from collections import OrderedDict
    from discriminator_vanilla_gan import Discriminator
    out = {}
    for size, latent, B in ((64, 100, 8), (128, 128, 4)):
        z = torch.from_numpy(I.gen_z(B, latent, SEED_Z))
        z2 = torch.from_numpy(I.gen_z(B, latent, SEED_Z + 1))
        real = torch.from_numpy(I.gen_real(B, size, SEED_REAL))
        plain = OrderedDict((k, (tuple(v.shape), "param")) for k, v in Discriminator(input_size=size).state_dict().items())

        def model():
            torch.manual_seed(0)
            m = VanillaGAN(latent_dim=latent, image_size=size, image_channels=1, use_spectral_norm=True, device="cpu")
            gs = load_state(m.generator, SEED_STATE_G)
            m.discriminator.load_state_dict({k: torch.from_numpy(v) for k, v in I.gen_sn_state(plain, SEED_STATE_D).items()})
            load_adam(m.g_optimizer, m.generator, gs, SEED_ADAM_G)
            # Adam moments of D keyed by the PLAIN names (weight_orig <-> weight), in the SN module's parameter order
            mm, vv, step = I.gen_adam(plain, SEED_ADAM_D)
            names = [k for k, _ in m.discriminator.named_parameters()]
            sd = m.d_optimizer.state_dict()
            sd["state"] = {i: {"step": torch.tensor(float(step)),
                               "exp_avg": torch.from_numpy(mm[k.replace("weight_orig", "weight")]).clone(),
                               "exp_avg_sq": torch.from_numpy(vv[k.replace("weight_orig", "weight")]).clone()}
                           for i, k in enumerate(names)}
            m.d_optimizer.load_state_dict(sd)
            return m
        tag = f"s{size}_b{B}"
        m = model()
        tap = MaskTap(m.discriminator)
        acts = ActTap(m.discriminator)                      # D(real) blocks, then D(fake) blocks
        torch.manual_seed(SEED_TORCH + 7)
        met = m.train_discriminator_step(real, noise=z)
        tap.close(); acts.close()
        acts.store(f"{tag}/d", out)
        out[f"{tag}/masks"] = I.pack_masks(tap.masks)
        record_step(f"{tag}/d", m, m.discriminator, m.d_optimizer, met, out, extra_buffers=True)
        # the G step follows on the SAME model: with D.eval() no power iteration runs, and the un-iterated random (u, v) of a
        # fresh state would give a sigma far from the spectral norm (saturated predictions, zero gradients)
        acts = ActTap(m.generator, m.discriminator)         # fc, G blocks, then D blocks
        met = m.train_generator_step(B, noise=z2)
        acts.close()
        acts.store(f"{tag}/g", out)
        record_step(f"{tag}/g", m, m.generator, m.g_optimizer, met, out, extra_buffers=True)
        probes(f"{tag}/g/dbuf", {k: v.float() for k, v in m.discriminator.state_dict().items() if k.endswith(("_u", "_v"))}, out)
    out["meta"] = np.array(json.dumps({"torch": torch.__version__, "threads": torch.get_num_threads(),
                                       "seeds": dict(z=SEED_Z, real=SEED_REAL, torch=SEED_TORCH + 7)}))
    path = os.path.join(HERE, "golden_sn_steps.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def make_ablation_step():
    """(xi) one iteration of AblationGANTrainer.train_epoch (ablation_vanilla_gan_signatures.py:397-467) replayed statement by
    statement on the reference's own Generator / Discriminator / BCELoss / Adam objects (that file itself cannot be
    imported here: torchvision absent; its ConfigurableGenerator with activation='relu' is layer for layer the standard
    Generator, ablation...py:216-328 vs generator_vanilla_gan.py:124-163)."""
    out = {}
    for size, latent, B in ((64, 100, 8), (128, 128, 4)):
        m = fresh_model(size, latent, warm=True)
        G, D, crit = m.generator, m.discriminator, m.criterion
        real = torch.from_numpy(I.gen_real(B, size, SEED_REAL))
        z = torch.from_numpy(I.gen_z(B, latent, SEED_Z))
        ls = 0.9
        G.train(); D.train()                                               # :403-404
        tap = MaskTap(D)
        # census in the harness' CALL order: D(real) blocks, Generator (fc + blocks), D(fake.detach()) blocks, D(fake) blocks of
        # the G update (hooks on D first, then G: ActTap appends per call, not per module)
        acts = ActTap(D, G)
        torch.manual_seed(SEED_TORCH + 5)
        real_labels, fake_labels = torch.full((B, 1), ls), torch.zeros(B, 1)   # :419-420
        m.d_optimizer.zero_grad()                                           # :423
        d_real_output = D(real)                                             # :426
        d_real_loss = crit(d_real_output, real_labels)
        fake_images = G(z)                                                  # :431
        d_fake_output = D(fake_images.detach())                             # :432
        d_fake_loss = crit(d_fake_output, fake_labels)
        d_loss = d_real_loss + d_fake_loss
        d_loss.backward()
        m.d_optimizer.step()                                                # :437
        tag = f"s{size}_b{B}"
        record_step(f"{tag}/d", m, D, m.d_optimizer,
                    {"d_loss": d_loss.item(), "d_loss_real": d_real_loss.item(), "d_loss_fake": d_fake_loss.item(),
                     "d_real_mean": d_real_output.mean().item(), "d_fake_mean": d_fake_output.mean().item()}, out)
        m.g_optimizer.zero_grad()                                           # :440
        d_output_for_g = D(fake_images)                                     # :442
        g_loss = crit(d_output_for_g, real_labels)                          # :443
        g_loss.backward()
        m.g_optimizer.step()                                                # :446
        tap.close(); acts.close()
        acts.store(f"{tag}", out)
        record_step(f"{tag}/g", m, G, m.g_optimizer, {"g_loss": g_loss.item(), "g_fake_mean": d_output_for_g.mean().item()},
                    out, extra_buffers=True)
        nb = 4 if size == 64 else 5
        assert len(tap.masks) == 3 * nb
        out[f"{tag}/masks"] = I.pack_masks(tap.masks)
    out["meta"] = np.array(json.dumps({"torch": torch.__version__, "threads": torch.get_num_threads(), "label_smoothing": 0.9,
                                       "seeds": dict(z=SEED_Z, real=SEED_REAL, torch=SEED_TORCH + 5)}))
    path = os.path.join(HERE, "golden_ablation_step.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def checkpoint_manifests():
    """(ix) structure of the two checkpoint layouts (keys, shapes, dtypes) -- no weights."""
    man = {}
    for size, latent in ((64, 100), (128, 128)):
        m = fresh_model(size, latent, warm=False)
        m.train_step(torch.from_numpy(I.gen_real(2, size, 1)))
        with tempfile.TemporaryDirectory() as d:
            m.save(os.path.join(d, "ck"))                             # layout B (vanilla_gan_model.py:450-472)
            b = torch.load(os.path.join(d, "ck.pt"), map_location="cpu", weights_only=False)
            cfg_json = json.load(open(os.path.join(d, "ck_config.json")))
        a = {  # layout A as assembled at train_vanilla_gan_signatures.py:417-427
            "epoch": 3, "global_step": 7,
            "generator_state_dict": m.generator.state_dict(),
            "discriminator_state_dict": m.discriminator.state_dict(),
            "g_optimizer_state_dict": m.g_optimizer.state_dict(),
            "d_optimizer_state_dict": m.d_optimizer.state_dict(),
            "config": {"latent_dim": latent, "image_size": size},
            "fixed_noise": torch.randn(64, latent), "best_g_loss": 1.0,
        }
        man[f"s{size}"] = {"layout_A": manifest(a), "layout_B": manifest(b),
                           "layout_B_config_json_keys": sorted(cfg_json),
                           "g_params": m.generator.get_num_params(),
                           "d_params": m.discriminator.get_num_params()}
    with open(os.path.join(HERE, "checkpoint_manifest.json"), "w") as f:
        json.dump(man, f, indent=None, separators=(",", ":"))   # key ORDER is part of the pin
    print("wrote checkpoint_manifest.json")


if __name__ == "__main__":
    torch.set_num_threads(8)
    if "--manifest-only" in sys.argv:
        checkpoint_manifests()
        sys.exit(0)
    if "--spectral-norm" in sys.argv:
        make_spectral_norm()
        sys.exit(0)
    if "--ablation" in sys.argv:
        make_ablation_step()
        sys.exit(0)
    if "--sn-steps" in sys.argv:
        make_spectral_norm_steps()
        sys.exit(0)
    if "--case" in sys.argv:                      # one more fixture: --case SIZE LATENT BATCH
        k = sys.argv.index("--case")
        make(int(sys.argv[k + 1]), int(sys.argv[k + 2]), int(sys.argv[k + 3]), full_image=False)
        sys.exit(0)
    make(64, 100, 4, full_image=True)
    make(64, 100, 64, full_image=False)
    make(128, 128, 4, full_image=False)
    make(128, 128, 32, full_image=False)
    make(64, 100, 128, full_image=False)          # BASELINE configs[3]: conv G/D 64x64, batch 128
    make(64, 100, 5, full_image=False)            # an odd batch (ragged tiles)
    make(128, 128, 5, full_image=False)           # ... at 128x128 (batch 3 was tried: three samples per BatchNorm statistic put the rounding
                                                  # noise of the exactly-zero fc bias gradient over the fixed bounds)
    make(64, 50, 8, full_image=False)             # a latent size of the ablation grid (ablation_vanilla_gan_signatures.py:597) that is not a
                                                  # multiple of 4: the Generator fc's generic kernels (generator_vanilla_gan.py:99,125)
    if "--cases-only" in sys.argv:                # the five step fixtures only (e.g. after adding a record to them)
        sys.exit(0)
    make_spectral_norm()
    make_ablation_step()
    make_spectral_norm_steps()
    checkpoint_manifests()
