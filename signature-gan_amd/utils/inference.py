"""Drop-in for the reference's ``utils/inference.py`` (checkpoint -> Generator, batched generation,
tensor -> PIL) on the MI355X HIP engine.

Checkpoint tolerance is the reference's (utils/inference.py:20-104): a dict with
``generator_state_dict`` (trainer layout A / VanillaGAN layout B), a dict with ``state_dict``, or a
bare Generator state_dict whose architecture is inferred from ``fc.*weight.shape[1]`` and the count
of ``upsample_blocks.N.block.0.weight`` keys (>= 5 -> 128x128)."""
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from ..generator_vanilla_gan import Generator

DEFAULT_LATENT_DIM, DEFAULT_IMAGE_SIZE, DEFAULT_IMAGE_CHANNELS = 100, 64, 1


def infer_architecture_from_state_dict(state_dict: Dict[str, Any]) -> Tuple[int, int]:
    latent_dim, blocks = DEFAULT_LATENT_DIM, 0
    for key, t in state_dict.items():
        if "fc" in key and "weight" in key and getattr(t, "dim", lambda: 0)() == 2:
            latent_dim = int(t.shape[1])
            break
    for key in state_dict:
        if "upsample_blocks" in key and ".0.weight" in key:
            try:
                blocks = max(blocks, int(key.split(".")[1]) + 1)
            except ValueError:
                pass
    return latent_dim, (128 if blocks >= 5 else 64)


def load_generator(checkpoint_path: str, device: torch.device) -> Tuple[Generator, int]:
    """(Generator in eval mode on ``device``, latent_dim).  Only the safe loader is used."""
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    channels = DEFAULT_IMAGE_CHANNELS
    if isinstance(ck, dict) and ("generator_state_dict" in ck or "state_dict" in ck):
        cfg = ck.get("config", {}) or {}
        latent_dim = cfg.get("latent_dim", DEFAULT_LATENT_DIM)
        size = cfg.get("image_size", DEFAULT_IMAGE_SIZE)
        channels = cfg.get("image_channels", DEFAULT_IMAGE_CHANNELS)
        sd = ck["generator_state_dict"] if "generator_state_dict" in ck else ck["state_dict"]
    else:
        sd = ck
        latent_dim, size = infer_architecture_from_state_dict(sd)
    g = Generator(latent_dim=latent_dim, output_size=size, output_channels=channels)
    g.load_state_dict(sd)
    g.to(device)
    g.eval()
    return g, latent_dim


def tensor_to_uint8(images: torch.Tensor) -> np.ndarray:
    """(B,1,H,W) in [-1,1] -> (B,H,W) uint8 with the reference's rule ((x+1)*127.5, clip, TRUNCATE)."""
    x = images.detach().float().cpu().numpy()
    return ((x[:, 0] + 1) * 127.5).clip(0, 255).astype(np.uint8)


def tensor_to_pil_image(tensor: torch.Tensor):
    from PIL import Image
    return Image.fromarray(tensor_to_uint8(tensor.unsqueeze(0))[0], mode="L")


def generate_signatures_batch(generator: Generator, n_samples: int, latent_dim: int, device: torch.device,
                              seed: Optional[int] = None, batch_size: int = 32, progress_callback=None,
                              noise_scale: float = 1.0) -> List[Any]:
    """Reference semantics (utils/inference.py:136-194): optional torch.manual_seed, then per batch
    z = randn(b, latent, device=device) * noise_scale -> generator(z) -> one PIL image per sample."""
    from PIL import Image
    if seed is not None:
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)
        np.random.seed(seed)
    out: List[Any] = []
    done = 0
    while done < n_samples:
        b = min(batch_size, n_samples - done)
        z = torch.randn(b, latent_dim, device=device) * noise_scale
        for arr in tensor_to_uint8(generator(z)):
            out.append(Image.fromarray(arr, mode="L"))
        done += b
        if progress_callback is not None:
            progress_callback(done / n_samples)
    return out
