"""Fully-connected ("MLP") vanilla GAN on the MI355X HIP engine -- a BUILD-DEFINED extension, PARITY UNPINNED.

BASELINE.json's configs[0] (z=100 -> 28x28 MLP G 100-256-512-784 / mirror D, batch 32) and the wording of configs[1] name a
fully-connected model; the reference has none (its "vanilla" Generator / Discriminator are convolutional and reject sizes
other than 64 / 128: generator_vanilla_gan.py:106-107, discriminator_vanilla_gan.py:121-122).  So there is no reference
class this module mirrors and no reference output to pin it to; the checker is the build's own CPU restatement
(oracle/mlp_oracle.py).  Conventions follow the conv drop-ins: BCELoss on probabilities, label smoothing 0.9 on the D step's
real labels, Adam(2e-4, (0.5, 0.999)), D step with G in eval mode, G step with BatchNorm batch statistics.

    G: z -> [Linear(h) -> BatchNorm1d -> ReLU for h in hidden] -> Linear(S*S) -> Tanh -> (B, 1, S, S)
    D: (B, 1, S, S) -> [Linear(h) -> LeakyReLU(0.2) for h in reversed(hidden)] -> Linear(1) -> Sigmoid -> (B, 1)

All compute is in ``libsiggan_hip.so`` (csrc/mlp.hip: every dense product on the fp32 matrix cores); torch holds the flat
parameter / gradient / Adam arenas the C ABI borrows (include/siggan_mlp.h).  No CPU path."""
import ctypes as C
from collections import OrderedDict

import torch

from . import _lib


def g_entries(latent, size, hidden):
    """state_dict-style entries of the generator in parameters() order: (key, shape)."""
    out, k = [], latent
    for i, h in enumerate(hidden):
        out += [(f"net.{i}.linear.weight", (h, k)), (f"net.{i}.linear.bias", (h,)), (f"net.{i}.bn.weight", (h,)), (f"net.{i}.bn.bias", (h,))]
        k = h
    return out + [("out.weight", (size * size, k)), ("out.bias", (size * size,))]


def d_entries(size, hidden):
    out, k = [], size * size
    for j, h in enumerate(reversed(hidden)):
        out += [(f"net.{j}.weight", (h, k)), (f"net.{j}.bias", (h,))]
        k = h
    return out + [("out.weight", (1, k)), ("out.bias", (1,))]


def _spans(entries):
    sp, off = OrderedDict(), 0
    for key, shape in entries:
        n = 1
        for d in shape:
            n *= d
        sp[key] = (off, n, shape)
        off += n
    return sp, off


class MLPGAN:
    """Generator + Discriminator + BCE + two Adam optimisers of the fully-connected variant (see module docstring)."""

    def __init__(self, latent_dim=100, image_size=28, hidden=(256, 512), max_batch=32, device="cuda:0", seed=0, leaky_slope=0.2,
                 g_lr=2e-4, d_lr=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("MLPGAN (HIP engine) needs a ROCm device; there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load()
        self.latent_dim, self.image_size, self.hidden, self.max_batch = int(latent_dim), int(image_size), tuple(int(h) for h in hidden), int(max_batch)
        self.hp = dict(g_lr=g_lr, d_lr=d_lr, beta1=beta1, beta2=beta2, ls=label_smoothing)
        hid = (C.c_int32 * 4)(*(list(self.hidden) + [0] * (4 - len(self.hidden))))
        cfg = _lib.MlpConfig(self.device.index, self.latent_dim, self.image_size, len(self.hidden), hid, self.max_batch, leaky_slope, int(seed))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.mlpgan_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.g_spans, g_total = _spans(g_entries(self.latent_dim, self.image_size, self.hidden))
        self.d_spans, d_total = _spans(d_entries(self.image_size, self.hidden))
        assert g_total == self.lib.mlpgan_param_count(h, 0) and d_total == self.lib.mlpgan_param_count(h, 1)
        assert len(self.g_spans) == self.lib.mlpgan_param_tensors(h, 0) and len(self.d_spans) == self.lib.mlpgan_param_tensors(h, 1)
        bn_total = sum(self.hidden)
        assert bn_total == self.lib.mlpgan_bn_count(h)
        z = lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device)
        self.g_params, self.g_grads, self.g_exp_avg, self.g_exp_avg_sq = z(g_total), z(g_total), z(g_total), z(g_total)
        self.d_params, self.d_grads, self.d_exp_avg, self.d_exp_avg_sq = z(d_total), z(d_total), z(d_total), z(d_total)
        self.g_adam_steps, self.d_adam_steps = z(len(self.g_spans)), z(len(self.d_spans))
        self.g_bn_mean, self.g_bn_var = z(bn_total), torch.ones(bn_total, dtype=torch.float32, device=self.device)
        self.g_bn_batches = z(len(self.hidden), torch.int64)
        self.metrics = z(_lib.M_COUNT)
        st = _lib.MlpStorage(*[t.data_ptr() for t in (
            self.g_params, self.g_grads, self.g_exp_avg, self.g_exp_avg_sq, self.g_adam_steps, self.g_bn_mean, self.g_bn_var,
            self.g_bn_batches, self.d_params, self.d_grads, self.d_exp_avg, self.d_exp_avg_sq, self.d_adam_steps)])
        _lib.check(self.lib.mlpgan_bind(h, C.byref(st)))
        self.init_weights(seed)

    def views(self, which, arena="params"):
        spans = self.g_spans if which == "g" else self.d_spans
        flat = getattr(self, f"{which}_{arena}")
        return OrderedDict((k, flat[o:o + n].view(shape)) for k, (o, n, shape) in spans.items())

    def bn_views(self):
        out, off = OrderedDict(), 0
        for i, h in enumerate(self.hidden):
            out[f"net.{i}.bn.running_mean"] = self.g_bn_mean[off:off + h]
            out[f"net.{i}.bn.running_var"] = self.g_bn_var[off:off + h]
            out[f"net.{i}.bn.num_batches_tracked"] = self.g_bn_batches[i]
            off += h
        return out

    def init_weights(self, seed=0):
        """The conv models' init distribution (generator_vanilla_gan.py:168-187): weights N(0, 0.02), biases 0, BatchNorm weight
        N(1, 0.02)."""
        gen = torch.Generator().manual_seed(int(seed))
        for which in ("g", "d"):
            for k, v in self.views(which).items():
                if k.endswith("bias"):
                    v.zero_()
                else:
                    v.copy_(torch.empty(v.shape).normal_(1.0 if ".bn.weight" in k else 0.0, 0.02, generator=gen))
            for a in ("grads", "exp_avg", "exp_avg_sq", "adam_steps"):
                getattr(self, f"{which}_{a}").zero_()
        self.g_bn_mean.zero_(); self.g_bn_var.fill_(1.0); self.g_bn_batches.zero_()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _hyper(self, lr, clip=None):
        return _lib.Hyper(lr, self.hp["beta1"], self.hp["beta2"], 1e-8, self.hp["ls"], clip if clip else 0.0, 1.0)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def _f32(self, t, what):
        if t is None:
            return None
        if t.device != self.device or t.dtype != torch.float32:
            raise ValueError(f"{what} must be a float32 tensor on {self.device}")
        return t.contiguous()

    def generate(self, z, training=False):
        z = self._f32(z, "z")
        if z.dim() != 2 or z.shape[1] != self.latent_dim or z.shape[0] > self.max_batch:
            raise ValueError(f"z must be (B <= {self.max_batch}, {self.latent_dim})")
        out = torch.empty(z.shape[0], 1, self.image_size, self.image_size, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mlpgan_g_forward(self._h, self._p(z), z.shape[0], int(training), self._p(out), self._stream()))
        return out

    def discriminate(self, x):
        x = self._f32(x, "x")
        s = self.image_size
        if x.dim() != 4 or tuple(x.shape[1:]) != (1, s, s) or x.shape[0] > self.max_batch:
            raise ValueError(f"x must be (B <= {self.max_batch}, 1, {s}, {s})")
        p = torch.empty(x.shape[0], 1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.mlpgan_d_forward(self._h, self._p(x), x.shape[0], self._p(p), self._stream()))
        return p

    def _metrics(self, keys):
        host = self.metrics.cpu()
        return {k: float(host[_lib.METRIC_INDEX[k]]) for k in keys}

    def train_discriminator_step(self, real, noise=None, clip=None, sync=True):
        real = self._f32(real, "real_images")
        hp = self._hyper(self.hp["d_lr"], clip)
        _lib.check(self.lib.mlpgan_d_step(self._h, self._p(real), real.shape[0], self._p(self._f32(noise, "noise")), C.byref(hp),
                                          self._p(self.metrics), self._stream()))
        return self._metrics(("d_loss", "d_loss_real", "d_loss_fake", "d_real_mean", "d_fake_mean")) if sync else None

    def train_generator_step(self, batch, noise=None, clip=None, sync=True):
        hp = self._hyper(self.hp["g_lr"], clip)
        _lib.check(self.lib.mlpgan_g_step(self._h, int(batch), self._p(self._f32(noise, "noise")), C.byref(hp), self._p(self.metrics),
                                          self._stream()))
        return self._metrics(("g_loss", "g_fake_mean")) if sync else None

    def train_step(self, real, sync=True):
        d = self.train_discriminator_step(real, sync=False)
        g = self.train_generator_step(real.shape[0], sync=False)
        return self._metrics(("d_loss", "d_loss_real", "d_loss_fake", "d_real_mean", "d_fake_mean", "g_loss", "g_fake_mean")) if sync else None

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mlpgan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def op_gemm(layout, a, b):
    """C = A . op(B) on the fp32 matrix cores (test hook): layout 'NT' A (M,K) B (N,K); 'NN' A (M,K) B (K,N); 'TN' A (K,M) B (K,N)."""
    lay = {"NT": 0, "NN": 1, "TN": 2}[layout]
    a, b = a.contiguous().float(), b.contiguous().float()
    if lay == 0:
        (m, k), n = a.shape, b.shape[0]
    elif lay == 1:
        (m, k), n = a.shape, b.shape[1]
    else:
        (k, m), n = a.shape, b.shape[1]
    c = torch.empty(m, n, dtype=torch.float32, device=a.device)
    _lib.check(_lib.load().mlpgan_op_gemm(a.device.index, lay, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(c.data_ptr()),
                                          m, n, k, C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)))
    return c
