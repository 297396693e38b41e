"""Drop-in for the reference's ``discriminator_vanilla_gan`` on the MI355X HIP engine.

Mirrors Discriminator (discriminator_vanilla_gan.py:84-282): same constructor, ``state_dict`` keys,
``forward`` -> probabilities (B, 1), ``forward_features`` -> (B, 8192).

Spectral normalisation (``use_spectral_norm=True``, :60-62, :200-202 -- ``torch.nn.utils.spectral_norm`` on every
conv and on the classifier): the module carries the reference's SN keys (``weight_orig`` parameter, ``weight_u`` /
``weight_v`` buffers, 20 keys for the 64x64 net) as views of the engine's storage, and the hook's arithmetic -- one power
iteration per ``train()``-mode forward, ``weight_orig / sigma`` with sigma = u . (W v) in every forward, the D step's
gradient through sigma -- runs inside the library (``csrc/sn.hip``)."""
from typing import Tuple

import torch

from . import layout
from ._modules import EngineBacked, build_tree, reference_init

_SN_SUFFIXES = ("weight_orig", "weight_u", "weight_v")


def _sn_entries(entries):
    """spectral_norm() deletes `weight` and registers `weight_orig` behind the existing `bias`: per layer the reference's
    key order is bias, weight_orig (parameters), weight_u, weight_v (buffers)."""
    out, pending = [], None
    for key, shape, kind in entries:
        if key.endswith(".weight"):
            base = key[:-len("weight")]
            cols = 1
            for d in shape[1:]:
                cols *= d
            pending = [(base + "weight_orig", shape, "param"), (base + "weight_u", (shape[0],), "sn_vec"), (base + "weight_v", (cols,), "sn_vec")]
        else:
            out.append((key, shape, kind))
            if pending and key.endswith(".bias"):
                out += pending
                pending = None
    return out


def _sn_init(key, shape, kind):
    if kind == "sn_vec":                                        # torch's spectral_norm: normalize(randn)
        return torch.nn.functional.normalize(torch.randn(shape), dim=0, eps=1e-12)
    return reference_init(key.replace("weight_orig", "weight"), shape, kind)


class Discriminator(EngineBacked):
    which = "d"

    def __init__(self, input_size: int = 64, input_channels: int = 1, use_spectral_norm: bool = False,
                 dropout: float = 0.25, leaky_slope: float = 0.2, _engine=None) -> None:
        super().__init__()
        layout.check_size(input_size, "input_size")              # ValueError like the reference (:121-122)
        if input_channels != 1:
            raise ValueError("the HIP engine builds grayscale (input_channels == 1) discriminators only")
        self.input_size, self.input_channels = input_size, input_channels
        self.use_spectral_norm, self.dropout, self.leaky_slope = use_spectral_norm, dropout, leaky_slope
        entries = layout.discriminator_entries(input_size, input_channels)
        if use_spectral_norm:
            build_tree(self, _sn_entries(entries), _sn_init)
        else:
            build_tree(self, entries, reference_init)
        if _engine is not None:
            if abs(_engine.dropout - dropout) > 1e-12 or abs(_engine.leaky_slope - leaky_slope) > 1e-12:
                raise ValueError("shared engine was created with different dropout / leaky_slope")
            if bool(_engine.spectral_norm) != bool(use_spectral_norm):
                raise ValueError("shared engine was created with a different spectral_norm setting")
            self._shared_engine = True
            self._attach(_engine, copy_in=True)

    # ---- spectral norm -------------------------------------------------------------------------------
    # The module keeps the reference's keys (weight_orig parameter, weight_u / weight_v buffers); the engine keeps weight_orig in
    # its parameter arena under the plain layer name and u / v in its own flat buffers.  All of torch's hook -- one power
    # iteration per training forward, sigma = u . (W v), weight = weight_orig / sigma, the gradient through sigma -- runs
    # inside the library (csrc/sn.hip).
    @staticmethod
    def _engine_name(name):
        return name[:-len("_orig")] if name.endswith("weight_orig") else name

    def _remap_views(self, engine, views, gviews):
        if not self.use_spectral_norm:
            return views, gviews
        if not engine.spectral_norm:
            raise ValueError("a spectral-norm Discriminator needs an Engine created with spectral_norm=True")
        v2 = {(k + "_orig" if k.endswith(".weight") else k): t for k, t in views.items()}
        g2 = {(k + "_orig" if k.endswith(".weight") else k): t for k, t in gviews.items()}
        v2.update(engine.sn_views())
        return v2, g2

    def _engine_kwargs(self):
        return dict(image_size=self.input_size, dropout=self.dropout, leaky_slope=self.leaky_slope,
                    spectral_norm=self.use_spectral_norm)

    def _forward(self, x, want_features):
        eng = self._require_engine()
        return eng.d_forward(x, training=self.training, want_features=want_features)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B, 1, S, S) -> P(real) (B, 1).  In train() mode Dropout2d is active (masks from the
        library RNG), in eval() mode it is off -- nn.Module semantics."""
        out = self._forward(x, False)
        return out[0] if isinstance(out, tuple) else out

    @torch.no_grad()
    def forward_features(self, x: torch.Tensor) -> torch.Tensor:
        return self._forward(x, True)[1]

    def get_input_shape(self) -> Tuple[int, int, int]:
        return (self.input_channels, self.input_size, self.input_size)


def create_discriminator(input_size: int = 64, input_channels: int = 1, use_spectral_norm: bool = False,
                         dropout: float = 0.25) -> Discriminator:
    return Discriminator(input_size=input_size, input_channels=input_channels,
                         use_spectral_norm=use_spectral_norm, dropout=dropout)
