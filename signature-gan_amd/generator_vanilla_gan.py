"""Drop-in for the reference's ``generator_vanilla_gan`` on the MI355X HIP engine.

Same constructor, attributes, ``state_dict`` keys and return shapes as the reference's Generator
(generator_vanilla_gan.py:69-237); ``forward`` runs fc+BatchNorm1d+ReLU, the transposed-convolution
blocks and the 3x3 conv + tanh as HIP kernels through ``siggan_g_forward``."""
from typing import Optional, Tuple

import torch

from . import layout
from ._modules import EngineBacked, build_tree, reference_init


class Generator(EngineBacked):
    which = "g"

    def __init__(self, latent_dim: int = 100, output_size: int = 64, output_channels: int = 1,
                 base_features: int = 256, _engine=None) -> None:
        super().__init__()
        layout.check_size(output_size, "output_size")            # ValueError like the reference (:106-107)
        if output_channels != 1:
            raise ValueError("the HIP engine builds grayscale (output_channels == 1) generators only")
        if base_features != 256:
            # the reference hard-codes the block widths (:133-147), so any other value breaks its forward
            raise ValueError("base_features must be 256 (the reference's block widths are fixed)")
        self.latent_dim, self.output_size = latent_dim, output_size
        self.output_channels, self.base_features = output_channels, base_features
        self.init_size = 4
        self.init_channels = layout.G_CHAIN[output_size][0]
        build_tree(self, layout.generator_entries(latent_dim, output_size, output_channels), reference_init)
        if _engine is not None:
            self._shared_engine = True
            self._attach(_engine, copy_in=True)

    def _engine_kwargs(self):
        return dict(latent_dim=self.latent_dim, image_size=self.output_size)

    @torch.no_grad()
    def forward(self, z: torch.Tensor) -> torch.Tensor:
        """z (B, latent_dim) -> images (B, 1, S, S) in [-1, 1].  ``self.training`` selects BatchNorm
        batch statistics (+ running-stat update) vs running statistics, as nn.Module.train()/eval()."""
        eng = self._require_engine()
        return eng.g_forward(z, training=self.training)

    def generate_latent(self, n_samples: int, device: Optional[torch.device] = None) -> torch.Tensor:
        if device is None:
            device = next(self.parameters()).device
        return torch.randn(n_samples, self.latent_dim, device=device)

    def get_output_shape(self) -> Tuple[int, int, int]:
        return (self.output_channels, self.output_size, self.output_size)


def create_generator(latent_dim: int = 100, output_size: int = 64, output_channels: int = 1) -> Generator:
    return Generator(latent_dim=latent_dim, output_size=output_size, output_channels=output_channels)
