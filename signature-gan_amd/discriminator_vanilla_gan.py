"""Drop-in for the reference's ``discriminator_vanilla_gan`` on the MI355X HIP engine.

Mirrors Discriminator (discriminator_vanilla_gan.py:84-282): same constructor, ``state_dict`` keys,
``forward`` -> probabilities (B, 1), ``forward_features`` -> (B, 8192).

Spectral normalisation (``use_spectral_norm=True``, :60-62, :200-202 -- ``torch.nn.utils.spectral_norm`` on every
conv and on the classifier) is supported for INFERENCE: the module then carries the reference's SN keys
(``weight_orig`` parameter, ``weight_u`` / ``weight_v`` buffers, 20 keys for the 64x64 net), and in ``eval()`` mode --
where torch's hook runs no power iteration -- the engine is handed ``weight_orig / sigma``, sigma = u . (W v), so a
checkpoint trained with the reference's ablation harness can be loaded and scored.  Training with SN (one power
iteration per training forward, gradients through sigma) is not built: ``train()``-mode forward raises."""
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
        self._sn_seen = None
        if use_spectral_norm:
            build_tree(self, _sn_entries(entries), _sn_init)
        else:
            build_tree(self, entries, reference_init)
        if _engine is not None:
            if abs(_engine.dropout - dropout) > 1e-12 or abs(_engine.leaky_slope - leaky_slope) > 1e-12:
                raise ValueError("shared engine was created with different dropout / leaky_slope")
            self._shared_engine = True
            self._attach(_engine, copy_in=True)

    def _engine_kwargs(self):
        return dict(image_size=self.input_size, dropout=self.dropout, leaky_slope=self.leaky_slope)

    # ---- spectral norm (inference) -------------------------------------------------------------------
    def _named_leaves(self):
        """Leaves that live in the engine's arenas: with SN only the biases (weight_orig / u / v stay module tensors)."""
        for name, t in super()._named_leaves():
            if not (self.use_spectral_norm and name.endswith(_SN_SUFFIXES)):
                yield name, t

    def _sn_sync(self, eng):
        """engine weight <- weight_orig / (u . (W v)), when weight_orig / u / v changed since the last call."""
        sd = dict(self.named_parameters())
        sd.update(dict(self.named_buffers()))
        stamp = tuple((k, t.data_ptr(), t._version) for k, t in sd.items() if k.endswith(_SN_SUFFIXES))
        if stamp == self._sn_seen:
            return
        views = eng.views("d")
        with torch.no_grad():
            for key, v in views.items():
                if key.endswith(".weight"):
                    base = key[:-len("weight")]
                    w, u, vv = sd[base + "weight_orig"], sd[base + "weight_u"], sd[base + "weight_v"]
                    sigma = torch.dot(u.to(v.device), torch.mv(w.to(v.device).reshape(w.shape[0], -1), vv.to(v.device)))
                    v.copy_(w.to(v.device) / sigma)
        eng.params_changed()
        self._sn_seen = stamp

    def _forward(self, x, want_features):
        eng = self._require_engine()
        if self.use_spectral_norm:
            if self.training:
                raise NotImplementedError("spectral-norm training (power iteration + gradients through sigma) is not built "
                                          "(SURVEY 8f-4); call .eval() to score with a spectral-norm checkpoint")
            self._sn_sync(eng)
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
