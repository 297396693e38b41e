"""Drop-in for the reference's ``train_vanilla_gan_signatures.py`` on the MI355X HIP engine.

Keeps the reference's CLI flags (train_vanilla_gan_signatures.py:638-798), ``TrainingConfig``
fields (:39-101), the per-step metric dictionaries of ``GANTrainer._train_discriminator`` /
``_train_generator`` (:281-376, one ``siggan_d_step`` / ``siggan_g_step`` call each, clipping
included), the layout-A checkpoint dictionary and file names (:402-484), the stdout lines the
reference's UI parses (``[Trainer] ...``, the tqdm postfix ``D=, G=, D(r)=, D(f)=`` and
``[Epoch NNNN] G_loss: ... | D_loss: ... | D(real): ... | D(fake): ...``), the
``epoch_%04d.png`` sample grids, CSV/JSON logs and the cooperative ``--stop_file``.

Data: ``data_loader_signatures.create_data_loader`` of this package -- images decoded once
into an HBM-resident uint8 cache, the reference's augmentation chain applied per batch by one kernel,
sample order and random draws those of the reference's DataLoader.
"""
import argparse
import csv
import json
import os
from dataclasses import asdict, dataclass
from datetime import datetime
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import torch

from .vanilla_gan_model import VanillaGAN


@dataclass
class TrainingConfig:
    latent_dim: int = 100
    image_size: int = 64
    image_channels: int = 1
    batch_size: int = 64
    epochs: int = 200
    g_lr: float = 2e-4
    d_lr: float = 2e-4
    beta1: float = 0.5
    beta2: float = 0.999
    label_smoothing: float = 0.9
    gradient_clip_value: Optional[float] = None
    n_critic: int = 1
    sample_interval: int = 5
    checkpoint_interval: int = 10
    num_workers: int = 4
    fixed_noise_samples: int = 64
    mode_collapse_threshold: float = 0.1
    mode_collapse_window: int = 50
    data_dir: str = ""
    checkpoint_dir: str = "./checkpoints"
    sample_dir: str = "./samples"
    log_dir: str = "./logs"

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)


class ModeCollapseDetector:
    """Training-health heuristic of the reference (:104-170): same three checks on a sliding
    window of generator losses and mean D(fake)."""

    def __init__(self, threshold: float = 0.1, window_size: int = 50) -> None:
        self.threshold, self.window_size = threshold, window_size
        self.g_losses: List[float] = []
        self.d_fake_outputs: List[float] = []

    def update(self, g_loss: float, d_fake_mean: float) -> None:
        self.g_losses.append(g_loss)
        self.d_fake_outputs.append(d_fake_mean)
        del self.g_losses[:-self.window_size], self.d_fake_outputs[:-self.window_size]

    def check_collapse(self) -> Tuple[bool, str]:
        if len(self.g_losses) < self.window_size:
            return False, "Insufficient data"
        fake = torch.tensor(self.d_fake_outputs)
        gl = torch.tensor(self.g_losses)
        d_var, d_mean = fake.var().item(), fake.mean().item()
        if d_var < self.threshold * 0.1:
            return True, f"D(fake) variance too low: {d_var:.6f}"
        if gl.var().item() < self.threshold and gl.mean().item() < 0.5:
            return True, f"G_loss stuck: mean={gl.mean().item():.4f}, var={gl.var().item():.6f}"
        if abs(d_mean - 0.5) < 0.05 and d_var < self.threshold:
            return True, f"D(fake) stuck at ~0.5: mean={d_mean:.4f}"
        return False, "Training appears stable"

    def reset(self) -> None:
        self.g_losses.clear()
        self.d_fake_outputs.clear()


class RunLogger:
    """Per-epoch metrics -> console line, CSV and JSON with the file names and fields of the
    reference's utils/logger.py (the UI discovers runs by these names)."""

    def __init__(self, log_dir: str, experiment_name: str) -> None:
        self.log_dir = Path(log_dir)
        self.log_dir.mkdir(parents=True, exist_ok=True)
        self.experiment_name = experiment_name
        self.timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        self._base = f"{experiment_name}_{self.timestamp}"
        self.metrics: List[Dict[str, Any]] = []
        self.config: Dict[str, Any] = {}

    def log_config(self, cfg: Dict[str, Any]) -> None:
        self.config = {"experiment_name": self.experiment_name, "timestamp": self.timestamp, **cfg}

    def log_metrics(self, epoch: int, g_loss: float, d_loss: float, d_real: float, d_fake: float) -> None:
        self.metrics.append({"epoch": epoch, "g_loss": g_loss, "d_loss": d_loss, "d_real": d_real, "d_fake": d_fake,
                             "timestamp": datetime.now().isoformat()})
        print(f"[Epoch {epoch:04d}] G_loss: {g_loss:.4f} | D_loss: {d_loss:.4f} | "
              f"D(real): {d_real:.4f} | D(fake): {d_fake:.4f}", flush=True)

    def save_to_csv(self) -> Path:
        path = self.log_dir / f"{self._base}_metrics.csv"
        if self.metrics:
            with open(path, "w", newline="", encoding="utf-8") as f:
                w = csv.DictWriter(f, fieldnames=list(self.metrics[0]))
                w.writeheader()
                w.writerows(self.metrics)
        return path

    def save_to_json(self) -> Path:
        path = self.log_dir / f"{self._base}_log.json"
        with open(path, "w", encoding="utf-8") as f:
            json.dump({"config": self.config, "metrics": self.metrics}, f, indent=2)
        return path

    def get_summary(self) -> Dict[str, Any]:
        if not self.metrics:
            return {"status": "no_data"}
        g = [m["g_loss"] for m in self.metrics]
        d = [m["d_loss"] for m in self.metrics]
        return {"experiment_name": self.experiment_name, "total_epochs": len(self.metrics), "final_g_loss": g[-1],
                "final_d_loss": d[-1], "min_g_loss": min(g), "min_d_loss": min(d), "avg_g_loss": sum(g) / len(g),
                "avg_d_loss": sum(d) / len(d)}


def save_sample_grid(images: torch.Tensor, path, nrow: int = 8, padding: int = 2) -> None:
    """8-per-row grid PNG, values mapped from [-1, 1] to uint8 by truncation (the reference's
    save_sample_grid(normalize=True, value_range=(-1, 1)) + tensor_to_pil_image rule)."""
    from PIL import Image
    x = images.detach().float().cpu().clamp(-1, 1)
    x = (x + 1.0) / 2.0
    n, _, h, w = x.shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = torch.zeros(rows * (h + padding) + padding, cols * (w + padding) + padding)
    for i in range(n):
        r, c = divmod(i, cols)
        grid[padding + r * (h + padding): padding + r * (h + padding) + h,
             padding + c * (w + padding): padding + c * (w + padding) + w] = x[i, 0]
    arr = (grid * 255.0).clamp(0, 255).to(torch.uint8).numpy()
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(arr, mode="L").save(str(path))


def create_data_loader(data_dir: str, batch_size: int, num_workers: int, image_size: int, **kw):
    """The device input pipeline (data_loader_signatures.py of this package): decoded images cached in HBM, the
    reference's augmentation chain as one kernel per batch, its DataLoader's sample order and random draws."""
    from .data_loader_signatures import create_data_loader as device_loader
    return device_loader(data_dir=data_dir, batch_size=batch_size, num_workers=num_workers, image_size=image_size, **kw)


class GANTrainer:
    """Training manager with the reference's method surface (:173-635)."""

    def __init__(self, config: TrainingConfig, device: Optional[str] = None, stop_file: Optional[str] = None) -> None:
        self.config = config
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self.stop_file = Path(stop_file).resolve() if stop_file else None
        for d in (config.checkpoint_dir, config.sample_dir, config.log_dir):
            Path(d).mkdir(parents=True, exist_ok=True)
        self.model = VanillaGAN(latent_dim=config.latent_dim, image_size=config.image_size,
                                image_channels=config.image_channels, g_lr=config.g_lr, d_lr=config.d_lr,
                                beta1=config.beta1, beta2=config.beta2, label_smoothing=config.label_smoothing,
                                device=self.device, max_batch=max(config.batch_size, config.fixed_noise_samples))
        self.logger = RunLogger(config.log_dir, "vanilla_gan_signatures")
        self.logger.log_config(config.to_dict())
        self.collapse_detector = ModeCollapseDetector(config.mode_collapse_threshold, config.mode_collapse_window)
        self.fixed_noise = torch.randn(config.fixed_noise_samples, config.latent_dim, device=self.model.device)
        self.start_epoch, self.global_step, self.best_g_loss = 0, 0, float("inf")
        self.data_loader = None
        print(f"[Trainer] Initialized on device: {self.device}")
        print(f"[Trainer] Generator params: {self.model.generator.get_num_params():,}")
        print(f"[Trainer] Discriminator params: {self.model.discriminator.get_num_params():,}")

    def _stop_requested(self) -> bool:
        try:
            return self.stop_file is not None and self.stop_file.exists()
        except OSError:
            return False

    # ---- the two steps the headline metric times ---------------------------------------------------
    def _train_discriminator(self, real_batch: torch.Tensor) -> Dict[str, float]:
        m = self.model.train_discriminator_step(real_batch, clip=self.config.gradient_clip_value)
        return {k: m[k] for k in ("d_loss", "d_loss_real", "d_loss_fake", "d_real_mean", "d_fake_mean", "d_grad_norm")}

    def _train_generator(self, batch_size: int) -> Dict[str, float]:
        m = self.model.train_generator_step(batch_size, clip=self.config.gradient_clip_value)
        return {k: m[k] for k in ("g_loss", "g_fake_mean", "g_grad_norm")}

    @torch.no_grad()
    def _generate_samples(self, epoch: int) -> None:
        self.model.generator.eval()
        path = Path(self.config.sample_dir) / f"epoch_{epoch:04d}.png"
        save_sample_grid(self.model.generator(self.fixed_noise), path, nrow=8)
        print(f"[Trainer] Saved samples to {path}")

    # ---- layout-A checkpoints (:402-484) -------------------------------------------------------------
    def _checkpoint_dict(self, epoch: int) -> Dict[str, Any]:
        snap = lambda sd: {k: v.detach().clone() for k, v in sd.items()}

        def opt_sd(opt):
            sd = opt.state_dict()
            sd["state"] = {i: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()}
                           for i, st in sd["state"].items()}
            return sd
        return {"epoch": epoch, "global_step": self.global_step,
                "generator_state_dict": snap(self.model.generator.state_dict()),
                "discriminator_state_dict": snap(self.model.discriminator.state_dict()),
                "g_optimizer_state_dict": opt_sd(self.model.g_optimizer),
                "d_optimizer_state_dict": opt_sd(self.model.d_optimizer),
                "config": self.config.to_dict(), "fixed_noise": self.fixed_noise.cpu(), "best_g_loss": self.best_g_loss,
                # extra key (the reference's loaders ignore it): where the library's z / dropout stream stands
                "engine_rng_state": list(self.model.engine.rng_state())}

    def _save_checkpoint(self, epoch: int, is_best: bool = False) -> Path:
        ck = self._checkpoint_dict(epoch)
        path = Path(self.config.checkpoint_dir) / f"checkpoint_epoch_{epoch:04d}.pt"
        torch.save(ck, path)
        print(f"[Trainer] Saved checkpoint: {path}")
        torch.save(ck, Path(self.config.checkpoint_dir) / "checkpoint_latest.pt")
        if is_best:
            torch.save(ck, Path(self.config.checkpoint_dir) / "checkpoint_best.pt")
            print("[Trainer] New best model saved!")
        return path

    def load_checkpoint(self, checkpoint_path: Optional[str] = None) -> int:
        path = Path(checkpoint_path) if checkpoint_path else Path(self.config.checkpoint_dir) / "checkpoint_latest.pt"
        if not path.exists():
            print(f"[Trainer] No checkpoint found at {path}")
            return 0
        print(f"[Trainer] Loading checkpoint: {path}")
        ck = torch.load(path, map_location=str(self.model.device), weights_only=True)
        self.model.generator.load_state_dict(ck["generator_state_dict"])
        self.model.discriminator.load_state_dict(ck["discriminator_state_dict"])
        self.model.g_optimizer.load_state_dict(ck["g_optimizer_state_dict"])
        self.model.d_optimizer.load_state_dict(ck["d_optimizer_state_dict"])
        self.start_epoch = ck["epoch"] + 1
        self.global_step = ck["global_step"]
        # a resumed run must not replay the z / dropout draws of steps 0..N: position the library RNG behind them
        # (one tick per optimiser update = n_critic + 1 per batch)
        eng = self.model.engine
        if "engine_rng_state" in ck:            # written by this trainer / VanillaGAN.save: the exact position
            eng.seed(int(ck["engine_rng_state"][0]), offset=int(ck["engine_rng_state"][1]))
        else:                                   # a reference-written checkpoint: the estimate
            eng.seed(eng._seed, offset=(int(getattr(self.config, "n_critic", 1)) + 1) * int(self.global_step))
        self.best_g_loss = ck.get("best_g_loss", float("inf"))
        if "fixed_noise" in ck:
            self.fixed_noise = ck["fixed_noise"].to(self.model.device)
        print(f"[Trainer] Resumed from epoch {ck['epoch']}")
        return self.start_epoch

    # ---- epoch loop (:486-635) ----------------------------------------------------------------------------
    def train(self, data_loader=None) -> Dict[str, Any]:
        cfg = self.config
        if data_loader is None:
            print(f"[Trainer] Loading data from: {cfg.data_dir}")
            data_loader = create_data_loader(data_dir=cfg.data_dir, batch_size=cfg.batch_size, num_workers=cfg.num_workers,
                                             image_size=cfg.image_size, shuffle=True, augment=True, drop_last=True,
                                             device=str(self.model.device))
        self.data_loader = data_loader
        print(f"[Trainer] Data loaded: {len(data_loader)} batches per epoch")
        if self.start_epoch == 0:
            self._generate_samples(epoch=0)
        try:
            from tqdm import tqdm
        except ImportError:                                   # pragma: no cover
            tqdm = None
        stopped = False
        try:
            for epoch in range(self.start_epoch, cfg.epochs):
                self.model.current_epoch = epoch
                if self._stop_requested():
                    print("\n[Trainer] Stop requested. Exiting before starting next epoch...")
                    break
                sums, seen = [0.0, 0.0, 0.0, 0.0], 0
                bar = tqdm(data_loader, desc=f"Epoch {epoch + 1}/{cfg.epochs}", leave=True, ncols=120) if tqdm else data_loader
                def account(dm, gm):
                    nonlocal seen
                    seen += 1
                    for i, v in enumerate((dm["d_loss"], gm["g_loss"], dm["d_real_mean"], dm["d_fake_mean"])):
                        sums[i] += v
                    self.collapse_detector.update(g_loss=gm["g_loss"], d_fake_mean=dm["d_fake_mean"])
                    if tqdm:
                        bar.set_postfix({"D": f"{dm['d_loss']:.4f}", "G": f"{gm['g_loss']:.4f}",
                                         "D(r)": f"{dm['d_real_mean']:.3f}", "D(f)": f"{dm['d_fake_mean']:.3f}"})

                it = iter(bar)
                nxt = next(it, None)
                pending = None                                    # metrics of the step still running on the device
                while nxt is not None:
                    real_batch, nxt = nxt, next(it, None)         # one batch of look-ahead: the next D(real) runs early
                    if isinstance(real_batch, (list, tuple)):
                        real_batch = real_batch[0]
                    if cfg.n_critic == 1:
                        # both steps as one pipelined engine step (same results as the two calls below); its metrics are
                        # read one step late, so the host prepares the next step while this one runs
                        ahead = nxt[0] if isinstance(nxt, (list, tuple)) else nxt
                        now = self.model.train_step(real_batch, next_real=ahead if ahead is not None and ahead.shape == real_batch.shape else None,
                                                    clip=cfg.gradient_clip_value, deferred=True)
                        if pending is not None:
                            m = pending.get()
                            account(m, m)
                        pending = now
                    else:
                        for _ in range(cfg.n_critic):
                            dm = self._train_discriminator(real_batch)
                        gm = self._train_generator(real_batch.size(0))
                        account(dm, gm)
                    self.global_step += 1
                    if self._stop_requested():
                        print("\n[Trainer] Stop requested. Stopping after current batch...")
                        stopped = True
                        break
                if pending is not None:
                    m = pending.get()
                    account(m, m)
                if seen == 0:
                    print("[Trainer] No batches processed for this epoch. Stopping.")
                    break
                avg_d, avg_g, avg_r, avg_f = (s / seen for s in sums)
                self.logger.log_metrics(epoch=epoch + 1, g_loss=avg_g, d_loss=avg_d, d_real=avg_r, d_fake=avg_f)
                collapsed, reason = self.collapse_detector.check_collapse()
                if collapsed:
                    print(f"\n[WARNING] Potential mode collapse detected: {reason}")
                is_best = avg_g < self.best_g_loss
                if is_best:
                    self.best_g_loss = avg_g
                if (epoch + 1) % cfg.sample_interval == 0 or stopped:
                    self._generate_samples(epoch + 1)
                if (epoch + 1) % cfg.checkpoint_interval == 0 or stopped:
                    self._save_checkpoint(epoch + 1, is_best=is_best)
                if stopped:
                    break
        except KeyboardInterrupt:
            print("\n[Trainer] KeyboardInterrupt received. Stopping training...")
        finally:
            self.logger.save_to_csv()
            self.logger.save_to_json()
        summary = self.logger.get_summary()
        print("\n" + "=" * 60 + "\nTraining Complete!\n" + "=" * 60)
        print(f"Total Epochs: {summary.get('total_epochs', cfg.epochs)}")
        print(f"Best G Loss: {self.best_g_loss:.4f}")
        return summary


def parse_arguments(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Train Vanilla GAN for Signature Generation (MI355X HIP engine)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--data_dir", type=str, default="./data/signatures/train", help="Path to training data directory")
    p.add_argument("--checkpoint_dir", type=str, default="./checkpoints", help="Directory to save checkpoints")
    p.add_argument("--sample_dir", type=str, default="./samples", help="Directory to save generated samples")
    p.add_argument("--log_dir", type=str, default="./logs", help="Directory to save training logs")
    p.add_argument("--run_dir", type=str, default=None, help="If set, overrides checkpoint/sample/log dirs under this run directory")
    p.add_argument("--stop_file", type=str, default=None, help="If set, training will stop when this file exists")
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--latent_dim", type=int, default=100)
    p.add_argument("--image_size", type=int, default=64, choices=[64, 128])
    p.add_argument("--g_lr", type=float, default=2e-4)
    p.add_argument("--d_lr", type=float, default=2e-4)
    p.add_argument("--beta1", type=float, default=0.5)
    p.add_argument("--label_smoothing", type=float, default=0.9)
    p.add_argument("--gradient_clip", type=float, default=None)
    p.add_argument("--n_critic", type=int, default=1)
    p.add_argument("--sample_interval", type=int, default=5)
    p.add_argument("--checkpoint_interval", type=int, default=10)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--resume_from", type=str, default=None)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--num_workers", type=int, default=4)
    return p.parse_args(argv)


def main(argv=None) -> Dict[str, Any]:
    a = parse_arguments(argv)
    if a.run_dir:
        run = Path(a.run_dir).resolve()
        run.mkdir(parents=True, exist_ok=True)
        a.checkpoint_dir, a.sample_dir, a.log_dir = str(run / "checkpoints"), str(run / "samples"), str(run / "logs")
    cfg = TrainingConfig(latent_dim=a.latent_dim, image_size=a.image_size, batch_size=a.batch_size, epochs=a.epochs,
                         g_lr=a.g_lr, d_lr=a.d_lr, beta1=a.beta1, label_smoothing=a.label_smoothing,
                         gradient_clip_value=a.gradient_clip, n_critic=a.n_critic, sample_interval=a.sample_interval,
                         checkpoint_interval=a.checkpoint_interval, num_workers=a.num_workers, data_dir=a.data_dir,
                         checkpoint_dir=a.checkpoint_dir, sample_dir=a.sample_dir, log_dir=a.log_dir)
    print("\n" + "=" * 60 + "\nVanilla GAN Training - Signature Generation\n" + "=" * 60)
    print(f"Data Directory: {cfg.data_dir}\nEpochs: {cfg.epochs}\nBatch Size: {cfg.batch_size}")
    print(f"Image Size: {cfg.image_size}x{cfg.image_size}\nLatent Dim: {cfg.latent_dim}")
    print(f"Gradient Clipping: {cfg.gradient_clip_value or 'Disabled'}\n" + "=" * 60 + "\n")
    trainer = GANTrainer(config=cfg, device=a.device, stop_file=a.stop_file)
    if a.resume or a.resume_from:
        trainer.load_checkpoint(a.resume_from)
    return trainer.train()


if __name__ == "__main__":
    main()
