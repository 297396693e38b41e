"""Drop-in for the reference's ``discriminator_vanilla_gan`` on the MI355X HIP engine.

Mirrors Discriminator (discriminator_vanilla_gan.py:84-282): same constructor, ``state_dict`` keys,
``forward`` -> probabilities (B, 1), ``forward_features`` -> (B, 8192).  Spectral normalisation is
only reachable from the ablation script in the reference and is not built."""
from typing import Tuple

import torch

from . import layout
from ._modules import EngineBacked, build_tree, reference_init


class Discriminator(EngineBacked):
    which = "d"

    def __init__(self, input_size: int = 64, input_channels: int = 1, use_spectral_norm: bool = False,
                 dropout: float = 0.25, leaky_slope: float = 0.2, _engine=None) -> None:
        super().__init__()
        layout.check_size(input_size, "input_size")              # ValueError like the reference (:121-122)
        if input_channels != 1:
            raise ValueError("the HIP engine builds grayscale (input_channels == 1) discriminators only")
        if use_spectral_norm:
            raise NotImplementedError("spectral normalisation is outside the accelerated path (SURVEY 8f-4)")
        self.input_size, self.input_channels = input_size, input_channels
        self.use_spectral_norm, self.dropout, self.leaky_slope = use_spectral_norm, dropout, leaky_slope
        build_tree(self, layout.discriminator_entries(input_size, input_channels), reference_init)
        if _engine is not None:
            if abs(_engine.dropout - dropout) > 1e-12 or abs(_engine.leaky_slope - leaky_slope) > 1e-12:
                raise ValueError("shared engine was created with different dropout / leaky_slope")
            self._shared_engine = True
            self._attach(_engine, copy_in=True)

    def _engine_kwargs(self):
        return dict(image_size=self.input_size, dropout=self.dropout, leaky_slope=self.leaky_slope)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, 1, S, S) -> P(real) (B, 1).  In train() mode Dropout2d is active (masks from the
        library RNG), in eval() mode it is off -- nn.Module semantics."""
        eng = self._require_engine()
        return eng.d_forward(x, training=self.training)

    @torch.no_grad()
    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        eng = self._require_engine()
        return eng.d_forward(x, training=self.training, want_features=True)[1]

    def get_input_shape(self) -> Tuple[int, int, int]:
        return (self.input_channels, self.input_size, self.input_size)


def create_discriminator(input_size: int = 64, input_channels: int = 1, use_spectral_norm: bool = False,
                         dropout: float = 0.25) -> Discriminator:
    return Discriminator(input_size=input_size, input_channels=input_channels,
                         use_spectral_norm=use_spectral_norm, dropout=dropout)
