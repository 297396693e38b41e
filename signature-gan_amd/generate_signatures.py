"""Drop-in for the reference's ``generate_signatures.py`` CLI: checkpoint in -> PNG files out, with the
Generator forward on the MI355X HIP engine (``siggan_g_forward``).  Flags, file naming
(``<prefix>_%06d.png``) and the ``--seed`` semantics follow generate_signatures.py:50-249."""
import argparse
import os
from typing import Any, Dict, Optional

import torch

from .utils.inference import generate_signatures_batch, load_generator


def generate_signatures(generator, n_samples: int, output_dir: str, batch_size: int = 64,
                        device: torch.device = torch.device("cuda"), seed: Optional[int] = None,
                        prefix: str = "signature") -> None:
    os.makedirs(output_dir, exist_ok=True)
    print(f"Output directory: {output_dir}")
    print(f"Generating {n_samples} signatures...")
    images = generate_signatures_batch(generator=generator, n_samples=n_samples, latent_dim=generator.latent_dim,
                                       device=device, seed=seed, batch_size=batch_size)
    print(f"Saving {len(images)} images...")
    for i, img in enumerate(images):
        img.save(os.path.join(output_dir, f"{prefix}_{i + 1:06d}.png"), "PNG")
    print("\nGeneration complete!")
    print(f"Generated {len(images)} signatures saved to: {output_dir}")


def get_checkpoint_info(checkpoint_path: str) -> Dict[str, Any]:
    if not os.path.exists(checkpoint_path):
        return {"error": f"Checkpoint not found: {checkpoint_path}"}
    ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    info: Dict[str, Any] = {"path": checkpoint_path, "type": type(ck).__name__}
    if isinstance(ck, dict):
        info["keys"] = list(ck.keys())
        for k in ("epoch", "config", "g_loss", "d_loss"):
            if k in ck:
                info[k] = ck[k]
    return info


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(description="Generate synthetic signatures (MI355X HIP engine)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--checkpoint", type=str, required=True, help="Path to the generator checkpoint file")
    p.add_argument("--n_samples", type=int, default=100, help="Number of signatures to generate")
    p.add_argument("--output_dir", type=str, default="./generated_signatures", help="Output directory")
    p.add_argument("--batch_size", type=int, default=64, help="Batch size for generation")
    p.add_argument("--seed", type=int, default=None, help="Random seed for reproducibility (optional)")
    p.add_argument("--prefix", type=str, default="signature", help="Filename prefix for generated images")
    p.add_argument("--device", type=str, default="auto", help="Device to use for inference")
    p.add_argument("--info", action="store_true", help="Display checkpoint information and exit")
    return p.parse_args(argv)


def main(argv=None) -> None:
    a = parse_args(argv)
    device = torch.device("cuda" if a.device == "auto" else a.device)
    print(f"Using device: {device}")
    if a.info:
        print("\nCheckpoint Information:")
        for k, v in get_checkpoint_info(a.checkpoint).items():
            print(f"  {k}: {v}")
        return
    generator, _ = load_generator(a.checkpoint, device)
    generate_signatures(generator, a.n_samples, a.output_dir, a.batch_size, device, a.seed, a.prefix)
    print("\n" + "=" * 50 + "\nGeneration Summary:")
    print(f"  Checkpoint: {a.checkpoint}\n  Samples generated: {a.n_samples}\n  Output directory: {a.output_dir}")
    print(f"  Seed: {a.seed if a.seed is not None else 'Random'}\n  Device: {device}\n" + "=" * 50)


if __name__ == "__main__":
    main()
