"""Drop-in for the reference's ``vanilla_gan_model.VanillaGAN`` (vanilla_gan_model.py:28-633).

G, D, BCE loss and the two Adam optimisers as one object; ``train_discriminator_step`` /
``train_generator_step`` are single calls into the HIP engine (``siggan_d_step`` / ``siggan_g_step``)
instead of an autograd graph.  Checkpoints (``save`` / ``load`` / ``from_checkpoint``) keep the
reference's layout B, including the ``<path>_config.json`` side-car."""
import json
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, Optional, Union

import torch
import torch.nn as nn

from ._modules import EngineAdam
from .discriminator_vanilla_gan import Discriminator
from .engine import Engine
from .generator_vanilla_gan import Generator


class _PendingMetrics:
    """Metrics of a step that is still running: an asynchronous copy into pinned host memory + an event."""

    def __init__(self, model, engine, with_norms: bool):
        self.model, self.with_norms = model, with_norms
        self.keys = engine.D_KEYS + engine.G_KEYS
        self.host = torch.empty(engine.metrics.numel(), dtype=torch.float32, pin_memory=True)
        self.host.copy_(engine.metrics, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record()
        self.value = None

    def get(self) -> Dict[str, float]:
        if self.value is None:
            from ._lib import METRIC_INDEX
            self.event.synchronize()
            out = {k: float(self.host[METRIC_INDEX[k]]) for k in self.keys}
            if self.with_norms:         # the trainer's variant reports the pre-clip norms (train...py:330-337, 370-376)
                for k in ("d_grad_norm", "g_grad_norm"):
                    out[k] = float(self.host[METRIC_INDEX[k]])
            self.model.d_losses.append(out["d_loss"]); self.model.g_losses.append(out["g_loss"])
            self.value = out
        return self.value


class VanillaGAN(nn.Module):
    def __init__(self, latent_dim: int = 100, image_size: int = 64, image_channels: int = 1, g_lr: float = 2e-4,
                 d_lr: float = 2e-4, beta1: float = 0.5, beta2: float = 0.999, label_smoothing: float = 0.9,
                 use_spectral_norm: bool = False, device: Optional[str] = None, max_batch: int = 64,
                 seed: Optional[int] = None) -> None:
        super().__init__()
        self.latent_dim, self.image_size, self.image_channels = latent_dim, image_size, image_channels
        self.g_lr, self.d_lr, self.beta1, self.beta2 = g_lr, d_lr, beta1, beta2
        self.label_smoothing, self.use_spectral_norm = label_smoothing, use_spectral_norm
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise RuntimeError("VanillaGAN (HIP engine) needs a ROCm device; there is no CPU path")
        if self._device.index is None:
            self._device = torch.device("cuda", torch.cuda.current_device())
        if seed is None:
            # the reference draws z / dropout from torch's global generator (never seeded by the trainer): follow
            # whatever seed that generator was given (torch.manual_seed, or the per-process random default)
            seed = torch.initial_seed() & ((1 << 63) - 1)
        self.engine = Engine(latent_dim=latent_dim, image_size=image_size, max_batch=max_batch,
                             device=str(self._device), seed=seed, image_channels=image_channels,
                             spectral_norm=use_spectral_norm)
        self.generator = Generator(latent_dim=latent_dim, output_size=image_size, output_channels=image_channels,
                                   _engine=self.engine)
        self.discriminator = Discriminator(input_size=image_size, input_channels=image_channels,
                                           use_spectral_norm=use_spectral_norm, _engine=self.engine)
        self.criterion = nn.BCELoss()            # kept for API parity; the loss is fused in the engine
        self.g_optimizer = EngineAdam(self.generator, lr=g_lr, betas=(beta1, beta2))
        self.d_optimizer = EngineAdam(self.discriminator, lr=d_lr, betas=(beta1, beta2))
        self.current_epoch = 0
        self.global_step = 0
        self.d_losses: list = []
        self.g_losses: list = []

    @property
    def device(self) -> torch.device:
        return self._device

    def to(self, device: Union[str, torch.device]) -> "VanillaGAN":
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self._device:
            raise RuntimeError(f"this VanillaGAN is bound to {self._device}; create a new one for {device}")
        return self

    # ---- training steps (vanilla_gan_model.py:180-336) ------------------------------------------
    def train_discriminator_step(self, real_images: torch.Tensor, noise: Optional[torch.Tensor] = None,
                                 clip: Optional[float] = None) -> Dict[str, float]:
        self.discriminator.train()
        self.generator.eval()
        h = self.d_optimizer.hyper()
        self.d_optimizer._sync_views()
        real = real_images.to(self._device, torch.float32)
        z = None if noise is None else noise.to(self._device, torch.float32)
        m = self.engine.d_step(real, z, None, h["lr"], h["beta1"], h["beta2"], h["eps"], self.label_smoothing, clip)
        self.d_losses.append(m["d_loss"])
        self.global_step += 1
        return m

    def train_generator_step(self, batch_size: int, noise: Optional[torch.Tensor] = None,
                             clip: Optional[float] = None) -> Dict[str, float]:
        self.generator.train()
        self.discriminator.eval()
        h = self.g_optimizer.hyper()
        self.g_optimizer._sync_views()
        z = None if noise is None else noise.to(self._device, torch.float32)
        m = self.engine.g_step(batch_size, z, h["lr"], h["beta1"], h["beta2"], h["eps"], clip)
        self.g_losses.append(m["g_loss"])
        return m

    def train_step(self, real_images: torch.Tensor, n_critic: int = 1, next_real: Optional[torch.Tensor] = None,
                   clip: Optional[float] = None, deferred: bool = False):
        """vanilla_gan_model.py:308-336.  With n_critic == 1 the two steps run as ONE pipelined engine
        step (the G step's forward beside the D step's backward; bit-identical results) with a single
        metrics read-back.  Extras: next_real -- the following step's batch, if the loop already holds it (its
        D(real) forward then runs beside this step's Generator backward); clip -- the trainer's
        gradient_clip_value (train_vanilla_gan_signatures.py:262-279); deferred -- return a handle whose
        .get() yields the metric dict, so the caller can enqueue the next step before it waits for this one."""
        if n_critic != 1:
            metrics: Dict[str, float] = {}
            for i in range(n_critic):
                d = self.train_discriminator_step(real_images, clip=clip)
                if i == n_critic - 1:
                    metrics.update(d)
            metrics.update(self.train_generator_step(real_images.size(0), clip=clip))
            return metrics
        self.discriminator.train()
        self.generator.train()          # the mode the reference leaves G in after train_generator_step
        hd, hg = self.d_optimizer.hyper(), self.g_optimizer.hyper()
        self.d_optimizer._sync_views(); self.g_optimizer._sync_views()
        real = real_images.to(self._device, torch.float32)
        e = self.engine
        e.step_begin(real, None, None, None, self.label_smoothing)
        e.d_apply(hd["lr"], hd["beta1"], hd["beta2"], hd["eps"], clip, 1.0, sync=False)
        if next_real is not None:
            e.stage_real(next_real.to(self._device, torch.float32))
        e.g_compute_grads(real.shape[0])
        e.g_apply(hg["lr"], hg["beta1"], hg["beta2"], hg["eps"], clip, 1.0, sync=False)
        self.global_step += 1
        self.discriminator.eval()       # reference: train_generator_step ends with D in eval mode
        pending = _PendingMetrics(self, e, bool(clip))
        return pending if deferred else pending.get()


    # ---- generation (vanilla_gan_model.py:338-407) ---------------------------------------------------
    @torch.no_grad()
    def generate(self, n_samples: int, device=None, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.generator.eval()
        if noise is None:
            noise = torch.randn(n_samples, self.latent_dim, device=self._device)
        return self.generator(noise.to(self._device, torch.float32))

    @torch.no_grad()
    def generate_interpolation(self, n_steps: int = 10, z_start=None, z_end=None) -> torch.Tensor:
        self.generator.eval()
        if z_start is None:
            z_start = torch.randn(1, self.latent_dim, device=self._device)
        if z_end is None:
            z_end = torch.randn(1, self.latent_dim, device=self._device)
        alphas = torch.linspace(0, 1, n_steps, device=self._device).view(-1, 1)
        return self.generator((z_start.to(self._device) * (1 - alphas) + z_end.to(self._device) * alphas).contiguous())

    # ---- configuration / checkpoints (vanilla_gan_model.py:409-560) -------------------------------------
    def get_config(self) -> Dict[str, Any]:
        g, d = self.generator.get_num_params(), self.discriminator.get_num_params()
        return {"latent_dim": self.latent_dim, "image_size": self.image_size, "image_channels": self.image_channels,
                "g_lr": self.g_lr, "d_lr": self.d_lr, "beta1": self.beta1, "beta2": self.beta2,
                "label_smoothing": self.label_smoothing, "use_spectral_norm": self.use_spectral_norm,
                "current_epoch": self.current_epoch, "global_step": self.global_step,
                "g_params": g, "d_params": d, "total_params": g + d}

    def _snapshot(self, sd):
        return {k: v.detach().clone() for k, v in sd.items()}

    def save(self, path: Union[str, Path], save_optimizer: bool = True, save_history: bool = True) -> None:
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        ck = {"config": self.get_config(),
              "generator_state_dict": self._snapshot(self.generator.state_dict()),
              "discriminator_state_dict": self._snapshot(self.discriminator.state_dict()),
              "current_epoch": self.current_epoch, "global_step": self.global_step,
              "saved_at": datetime.now().isoformat(),
              # extra key (the reference's loaders ignore it): where the library's z / dropout stream stands
              "engine_rng_state": list(self.engine.rng_state())}
        if save_optimizer:
            for key, opt in (("g_optimizer_state_dict", self.g_optimizer), ("d_optimizer_state_dict", self.d_optimizer)):
                sd = opt.state_dict()
                sd["state"] = {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                               for i, st in sd["state"].items()}
                ck[key] = sd
        if save_history:
            ck["d_losses"], ck["g_losses"] = list(self.d_losses), list(self.g_losses)
        torch.save(ck, f"{path}.pt")
        with open(f"{path}_config.json", "w") as f:
            json.dump(self.get_config(), f, indent=2)
        print(f"Model saved to {path}.pt")

    def load(self, path: Union[str, Path], load_optimizer: bool = True, load_history: bool = True,
             map_location: Optional[str] = None) -> None:
        path = Path(path)
        if not path.suffix:
            path = Path(f"{path}.pt")
        # checkpoints hold tensors and primitive containers only, so the safe loader suffices
        ck = torch.load(path, map_location=map_location or str(self._device), weights_only=True)
        self.generator.load_state_dict(ck["generator_state_dict"])
        self.discriminator.load_state_dict(ck["discriminator_state_dict"])
        self.current_epoch = ck.get("current_epoch", 0)
        self.global_step = ck.get("global_step", 0)
        # continue the z / dropout stream where the saved run stood; a checkpoint without the key (written by the reference)
        # gets the estimate "two optimiser updates per global step" (exact for train_step with n_critic = 1)
        if "engine_rng_state" in ck:
            self.engine.seed(int(ck["engine_rng_state"][0]), offset=int(ck["engine_rng_state"][1]))
        else:
            self.engine.seed(self.engine._seed, offset=2 * int(self.global_step))
        if load_optimizer and "g_optimizer_state_dict" in ck:
            self.g_optimizer.load_state_dict(ck["g_optimizer_state_dict"])
            self.d_optimizer.load_state_dict(ck["d_optimizer_state_dict"])
        if load_history and "d_losses" in ck:
            self.d_losses, self.g_losses = list(ck.get("d_losses", [])), list(ck.get("g_losses", []))
        print(f"Model loaded from {path}")
        print(f"  Epoch: {self.current_epoch}, Global Step: {self.global_step}")

    @classmethod
    def from_checkpoint(cls, path: Union[str, Path], device: Optional[str] = None) -> "VanillaGAN":
        path = Path(path)
        if not path.suffix:
            path = Path(f"{path}.pt")
        cfg = torch.load(path, map_location="cpu", weights_only=True)["config"]
        model = cls(latent_dim=cfg["latent_dim"], image_size=cfg["image_size"], image_channels=cfg["image_channels"],
                    g_lr=cfg["g_lr"], d_lr=cfg["d_lr"], beta1=cfg["beta1"], beta2=cfg["beta2"],
                    label_smoothing=cfg["label_smoothing"], use_spectral_norm=cfg["use_spectral_norm"], device=device)
        model.load(path, load_optimizer=True, load_history=True)
        return model

    def set_learning_rates(self, g_lr: float, d_lr: float) -> None:
        for opt, lr in ((self.g_optimizer, g_lr), (self.d_optimizer, d_lr)):
            for group in opt.param_groups:
                group["lr"] = lr
        self.g_lr, self.d_lr = g_lr, d_lr

    def get_recent_losses(self, n: int = 100) -> Dict[str, float]:
        d, g = self.d_losses[-n:] or [0], self.g_losses[-n:] or [0]
        return {"avg_d_loss": sum(d) / len(d), "avg_g_loss": sum(g) / len(g)}

    def summary(self) -> str:
        c = self.get_config()
        bar = "=" * 60
        return "\n".join([bar, "VanillaGAN Model Summary (MI355X HIP engine)", bar, f"Device: {self._device}",
                          f"Latent Dimension: {c['latent_dim']}", f"Image Size: {c['image_size']}x{c['image_size']}",
                          f"Generator parameters: {c['g_params']:,} (lr {c['g_lr']})",
                          f"Discriminator parameters: {c['d_params']:,} (lr {c['d_lr']})",
                          f"Label Smoothing: {c['label_smoothing']}  Adam betas: ({c['beta1']}, {c['beta2']})",
                          f"Epoch {c['current_epoch']}  Global step {c['global_step']}", bar])


def create_vanilla_gan(latent_dim: int = 100, image_size: int = 64, use_spectral_norm: bool = False,
                       device: Optional[str] = None) -> VanillaGAN:
    return VanillaGAN(latent_dim=latent_dim, image_size=image_size, image_channels=1,
                      use_spectral_norm=use_spectral_norm, device=device)
