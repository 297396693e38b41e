"""Host-side context of the HIP engine: owns the torch flat arenas the C ABI borrows and exposes
the step / forward calls.  PyTorch here is plumbing only (device memory, streams)."""
import ctypes as C

import torch

from . import _lib, layout


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32(t, device, what):
    if t is None:
        return None
    if t.device != device:
        raise ValueError(f"{what} must live on {device}, got {t.device}")
    if t.dtype != torch.float32:
        raise ValueError(f"{what} must be float32, got {t.dtype}")
    return t.contiguous()


class Engine:
    """One MI355X context for a (latent_dim, image_size) model.

    Arenas (all on ``device``): ``g_params/g_grads/g_exp_avg/g_exp_avg_sq`` and the ``d_*``
    equivalents are flat fp32 tensors in the reference's parameters() order; ``g_views`` /
    ``d_views`` map state_dict keys to views into them (so ``state_dict()`` / checkpoints need
    no conversion)."""

    def __init__(self, latent_dim=100, image_size=64, max_batch=64, device="cuda:0", seed=0,
                 dropout=0.25, leaky_slope=0.2, image_channels=1, dtype="f32", f16_grad_scale=0.0, spectral_norm=False):
        layout.check_size(image_size)
        if image_channels != 1:
            raise ValueError(f"only image_channels == 1 is built, got {image_channels}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP engine needs a ROCm device ('cuda:N'); there is no CPU path")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.load()
        self.latent_dim, self.image_size, self.max_batch = int(latent_dim), int(image_size), int(max_batch)
        self.dropout, self.leaky_slope, self._seed = float(dropout), float(leaky_slope), int(seed)
        # storage type of the library-owned activations and MFMA weight copies ("f32": the reference's arithmetic and
        # the parity path; "bf16" / "f16": 16-bit storage and MFMA operands, fp32 everything else -- BASELINE configs[2] / [4])
        if str(dtype) not in _lib.DTYPES:
            raise ValueError(f"dtype must be one of f32 / bf16 / f16, got {dtype!r}")
        self.dtype = {0: "f32", 1: "bf16", 2: "f16"}[_lib.DTYPES[str(dtype)]]
        self.act_dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[self.dtype]
        self.f16_grad_scale = float(f16_grad_scale)
        # torch.nn.utils.spectral_norm on every Discriminator conv + the classifier (discriminator_vanilla_gan.py:60-62,200-202):
        # d_params then holds weight_orig, d_sn_u / d_sn_v the weight_u / weight_v buffers
        self.spectral_norm = bool(spectral_norm)
        self._h = None
        self._staged = None
        self._comm = None
        self._mode = 2                             # SIGGAN_MODE_OVERLAP (set_mode); re-applied when the context is re-created
        self._create_context()
        h = self._h
        self.g_entries = layout.generator_entries(latent_dim, image_size)
        self.d_entries = layout.discriminator_entries(image_size)
        self.g_spans, g_total = layout.spans(self.g_entries)
        self.d_spans, d_total = layout.spans(self.d_entries)
        self.bn_spans, bn_total = layout.spans(self.g_entries, "bn_mean")
        assert g_total == self.lib.siggan_param_count(h, 0) and d_total == self.lib.siggan_param_count(h, 1)
        assert bn_total == self.lib.siggan_bn_count(h)
        for which, sp in ((0, self.g_spans), (1, self.d_spans)):
            assert len(sp) == self.lib.siggan_param_tensors(h, which)
            o, n = C.c_int64(), C.c_int64()
            for i, (off, num, _) in enumerate(sp.values()):
                _lib.check(self.lib.siggan_param_span(h, which, i, C.byref(o), C.byref(n)))
                assert (o.value, n.value) == (off, num), "host/library layout mismatch"
        dev, z = self.device, lambda n, dt=torch.float32: torch.zeros(n, dtype=dt, device=self.device)
        self.g_params, self.g_grads, self.g_exp_avg, self.g_exp_avg_sq = z(g_total), z(g_total), z(g_total), z(g_total)
        self.d_params, self.d_grads, self.d_exp_avg, self.d_exp_avg_sq = z(d_total), z(d_total), z(d_total), z(d_total)
        self.g_adam_steps, self.d_adam_steps = z(len(self.g_spans)), z(len(self.d_spans))
        self.g_bn_mean, self.g_bn_var = z(bn_total), torch.ones(bn_total, dtype=torch.float32, device=dev)
        self.g_bn_batches = z(len(self.bn_spans), torch.int64)
        self.metrics = z(_lib.M_COUNT)
        self.d_sn_u = self.d_sn_v = None
        if self.spectral_norm:
            # spans of weight_u (Cout) / weight_v (Cin*kh*kw) per layer, conv blocks then classifier; torch's init: normalize(randn)
            self.sn_spans, uo, vo = {}, 0, 0
            for key, (o, n, shape) in self.d_spans.items():
                if key.endswith(".weight"):
                    rows, cols = shape[0], n // shape[0]
                    self.sn_spans[key[:-len("weight")]] = (uo, rows, vo, cols)
                    uo, vo = uo + rows, vo + cols
            assert uo == self.lib.siggan_sn_count(h, 0) and vo == self.lib.siggan_sn_count(h, 1)
            self.d_sn_u, self.d_sn_v = z(uo), z(vo)
            gen = torch.Generator().manual_seed(self._seed & 0x7FFFFFFF)
            for base, (u0, rows, v0, cols) in self.sn_spans.items():
                self.d_sn_u[u0:u0 + rows] = torch.nn.functional.normalize(torch.randn(rows, generator=gen), dim=0, eps=1e-12).to(dev)
                self.d_sn_v[v0:v0 + cols] = torch.nn.functional.normalize(torch.randn(cols, generator=gen), dim=0, eps=1e-12).to(dev)
        self._bind()
        self.d_chans = list(layout.D_CHAIN[image_size])

    def _create_context(self):
        cfg = _lib.Config(self.device.index, self.latent_dim, self.image_size, 1, self.max_batch,
                          self.dropout, self.leaky_slope, self._seed, _lib.DTYPES[self.dtype], self.f16_grad_scale,
                          int(self.spectral_norm))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.siggan_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def _bind(self):
        st = _lib.Storage(*[t.data_ptr() for t in (
            self.g_params, self.g_grads, self.g_exp_avg, self.g_exp_avg_sq, self.g_adam_steps, self.g_bn_mean,
            self.g_bn_var, self.g_bn_batches, self.d_params, self.d_grads, self.d_exp_avg, self.d_exp_avg_sq,
            self.d_adam_steps)] + [t.data_ptr() if t is not None else 0 for t in (self.d_sn_u, self.d_sn_v)])
        _lib.check(self.lib.siggan_bind(self._h, C.byref(st)))
        self._staged = None

    def ensure_batch(self, batch):
        """Grow the library workspace (a new context bound to the SAME arenas) when a larger batch
        than ``max_batch`` arrives; parameters, moments and buffers are untouched."""
        if batch <= self.max_batch:
            return
        if self._comm is not None:
            raise RuntimeError(f"batch {batch} exceeds max_batch {self.max_batch}: a context that holds a communicator "
                               "cannot be re-created on one rank alone; size max_batch for the largest batch up front")
        torch.cuda.synchronize(self.device)
        seed, offset = self.rng_state()          # the z / dropout stream continues where the old context stood
        self.lib.siggan_destroy(self._h)
        self._h = None
        self._staged = None                      # a batch staged in the old context is gone
        self.max_batch = 1 << (int(batch) - 1).bit_length()
        self._create_context()
        self._bind()
        if self._mode != 2:
            _lib.check(self.lib.siggan_set_mode(self._h, self._mode))
        self.seed(seed, offset)

    # ---- views -------------------------------------------------------------------------------
    def views(self, which, arena="params"):
        """OrderedDict state_dict key -> view into the chosen flat arena ('params', 'grads',
        'exp_avg', 'exp_avg_sq')."""
        spans = self.g_spans if which == "g" else self.d_spans
        flat = getattr(self, f"{which}_{arena}")
        return {k: flat[o:o + n].view(shape) for k, (o, n, shape) in spans.items()}

    def sn_views(self):
        """state_dict key -> view for the spectral-norm buffers: '<layer>.weight_u' (Cout) and '<layer>.weight_v' (Cin*kh*kw)."""
        out = {}
        for base, (u0, rows, v0, cols) in self.sn_spans.items():
            out[base + "weight_u"] = self.d_sn_u[u0:u0 + rows]
            out[base + "weight_v"] = self.d_sn_v[v0:v0 + cols]
        return out

    def bn_views(self):
        out = {}
        for i, (k, (o, n, shape)) in enumerate(self.bn_spans.items()):
            out[k] = self.g_bn_mean[o:o + n]
            out[k.replace("running_mean", "running_var")] = self.g_bn_var[o:o + n]
            out[k.replace("running_mean", "num_batches_tracked")] = self.g_bn_batches[i]
        return out

    def init_reference(self, seed=0):
        """Fresh parameters drawn from the reference's init distribution (Conv/ConvT/Linear weight
        ~ N(0, 0.02), biases 0, BatchNorm weight ~ N(1, 0.02), bias 0: generator_vanilla_gan.py:168-187,
        discriminator_vanilla_gan.py:212-239); BN buffers reset; Adam state cleared.  Statistical,
        not bit-identical to torch's own init stream."""
        gen = torch.Generator().manual_seed(int(seed))
        for which in ("g", "d"):
            for k, v in self.views(which).items():
                if k.endswith("bias"):
                    v.zero_()
                else:
                    mean = 1.0 if ".1.weight" in k else 0.0
                    v.copy_(torch.empty(v.shape).normal_(mean, 0.02, generator=gen))
            for a in ("grads", "exp_avg", "exp_avg_sq", "adam_steps"):
                getattr(self, f"{which}_{a}").zero_()
        self.g_bn_mean.zero_(); self.g_bn_var.fill_(1.0); self.g_bn_batches.zero_()
        self.params_changed()

    def params_changed(self):
        """Call after writing parameters / BN buffers from outside (load_state_dict, init)."""
        _lib.check(self.lib.siggan_params_changed(self._h))

    def set_mode(self, graph=False, overlap=True):
        """Step-phase execution mode: hipGraph replay and/or side-stream overlap (default: overlap only)."""
        self._mode = (1 if graph else 0) | (2 if overlap else 0)
        _lib.check(self.lib.siggan_set_mode(self._h, self._mode))

    def set_step_variant(self, variant="trainer"):
        """'trainer': the reference's GANTrainer / VanillaGAN iteration (default).  'ablation': AblationGANTrainer.train_epoch's
        (ablation_vanilla_gan_signatures.py:397-467) -- both nets in train mode, one shared Generator forward, G target =
        smoothed label; see ablation_step."""
        v = {"trainer": 0, "ablation": 1}[variant]
        _lib.check(self.lib.siggan_set_step_variant(self._h, v))
        self.step_variant = variant

    def ablation_step(self, real, z=None, masks=None, lr_d=2e-4, lr_g=2e-4, beta1=0.5, beta2=0.999, eps=1e-8,
                      label_smoothing=0.9, sync=True):
        """One iteration of the ablation harness' loop (needs set_step_variant('ablation')): masks, if given, are the three
        Dropout2d mask sets (real pass, fake pass of the D update, fake pass of the G update), each one (B, C_l) per block."""
        if getattr(self, "step_variant", "trainer") != "ablation":
            raise RuntimeError("call set_step_variant('ablation') first")
        self.d_compute_grads(real, z, masks, label_smoothing, mask_passes=3)
        dm = self.d_apply(lr_d, beta1, beta2, eps, None, 1.0, sync)
        self.g_compute_grads(real.shape[0], label_smoothing=label_smoothing)
        gm = self.g_apply(lr_g, beta1, beta2, eps, None, 1.0, sync)
        if sync:
            dm.update(gm)
        return dm

    def seed(self, seed, offset=0):
        """(Re)position the library RNG (z, dropout tables): Philox key ``seed``, call counter ``offset`` (the counter
        ticks once per optimiser update, i.e. twice per G+D step)."""
        self._seed = int(seed)
        _lib.check(self.lib.siggan_seed(self._h, int(seed), int(offset)))

    def rng_state(self):
        """(seed, offset) of the library RNG as it stands (synchronises the device)."""
        s, o = C.c_uint64(), C.c_uint64()
        _lib.check(self.lib.siggan_rng_state(self._h, C.byref(s), C.byref(o)))
        return int(s.value), int(o.value)

    # ---- data parallelism: the library's own RCCL communicator ----------------------------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (rank 0 draws it, every rank passes the same bytes to comm_init)."""
        buf = C.create_string_buffer(128)
        _lib.check(_lib.load().siggan_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        """From here on d_apply / g_apply (hence d_step / g_step / train_step) sum-all-reduce the gradient bucket over
        RCCL inside the library and average it; world == 1 is allowed (the collective runs and changes nothing)."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of Engine.comm_unique_id()")
        self._comm_id = C.create_string_buffer(bytes(unique_id), 128)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.siggan_comm_init(self._h, int(rank), int(world), self._comm_id))
        self._comm = (int(rank), int(world), bytes(unique_id))

    def comm_destroy(self):
        _lib.check(self.lib.siggan_comm_destroy(self._h))
        self._comm = None

    @property
    def comm_world(self):
        return int(self.lib.siggan_comm_world(self._h))

    def comm_broadcast(self, tensor, root=0):
        """In-place broadcast of a device tensor from ``root`` over the library communicator (no-op without one)."""
        if tensor.device != self.device or not tensor.is_contiguous():
            raise ValueError("comm_broadcast needs a contiguous tensor on the engine's device")
        _lib.check(self.lib.siggan_comm_broadcast(self._h, _ptr(tensor), tensor.numel() * tensor.element_size(), int(root),
                                                  self._stream()))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _hyper(self, lr, beta1, beta2, eps=1e-8, label_smoothing=0.9, clip=None, grad_scale=1.0):
        return _lib.Hyper(lr, beta1, beta2, eps, label_smoothing, clip if clip else 0.0, grad_scale)

    def _check_batch(self, b):
        if b < 1:
            raise ValueError(f"batch must be >= 1, got {b}")
        self.ensure_batch(b)

    # ---- forward passes ------------------------------------------------------------------------
    def g_forward(self, z, training=False, out=None):
        z = _f32(z, self.device, "z")
        if z.dim() != 2 or z.shape[1] != self.latent_dim:
            raise ValueError(f"z must be (B, {self.latent_dim}), got {tuple(z.shape)}")
        b = z.shape[0]
        self._check_batch(b)
        if out is None:
            out = torch.empty(b, 1, self.image_size, self.image_size, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.siggan_g_forward(self._h, _ptr(z), b, int(training), _ptr(out), self._stream()))
        return out

    def _masks(self, masks, b, passes):
        if masks is None:
            return None
        if isinstance(masks, (list, tuple)):
            assert len(masks) == passes * len(self.d_chans)
            masks = torch.cat([m.to(self.device, torch.float32).reshape(-1) for m in masks])
        masks = _f32(masks, self.device, "masks")
        if masks.numel() != passes * b * sum(self.d_chans):
            raise ValueError("dropout masks have the wrong size")
        return masks

    def d_forward(self, x, training=False, masks=None, want_features=False):
        x = _f32(x, self.device, "x")
        s = self.image_size
        if x.dim() != 4 or tuple(x.shape[1:]) != (1, s, s):
            raise ValueError(f"x must be (B, 1, {s}, {s}), got {tuple(x.shape)}")
        b = x.shape[0]
        self._check_batch(b)
        masks = self._masks(masks, b, 1)
        probs = torch.empty(b, 1, dtype=torch.float32, device=self.device)
        feat = torch.empty(b, self.d_chans[-1] * 16, dtype=torch.float32, device=self.device) if want_features else None
        _lib.check(self.lib.siggan_d_forward(self._h, _ptr(x), b, int(training), _ptr(masks), _ptr(probs), _ptr(feat),
                                             self._stream()))
        return (probs, feat) if want_features else probs

    # ---- training steps --------------------------------------------------------------------------
    def _metrics(self, keys, sync):
        if not sync:
            return None
        host = self.metrics.cpu()
        return {k: float(host[_lib.METRIC_INDEX[k]]) for k in keys}

    D_KEYS = ("d_loss", "d_loss_real", "d_loss_fake", "d_real_mean", "d_fake_mean", "d_real_acc", "d_fake_acc")
    G_KEYS = ("g_loss", "g_fake_mean")

    def d_compute_grads(self, real, z=None, masks=None, label_smoothing=0.9, mask_passes=2):
        real = _f32(real, self.device, "real_images")
        s = self.image_size
        if real.dim() != 4 or tuple(real.shape[1:]) != (1, s, s):
            raise ValueError(f"real_images must be (B, 1, {s}, {s}), got {tuple(real.shape)}")
        b = real.shape[0]
        self._check_batch(b)
        z = _f32(z, self.device, "noise")
        if z is not None and tuple(z.shape) != (b, self.latent_dim):
            raise ValueError(f"noise must be ({b}, {self.latent_dim})")
        masks = self._masks(masks, b, mask_passes)
        hp = self._hyper(0.0, 0.5, 0.999, label_smoothing=label_smoothing)
        self._staged = None
        _lib.check(self.lib.siggan_d_grads(self._h, _ptr(real), b, _ptr(z), _ptr(masks), C.byref(hp), _ptr(self.metrics),
                                           self._stream()))

    def stage_real(self, next_real):
        """Hand over the NEXT step's real batch (siggan_stage_real): call between d_apply and g_compute_grads;
        that g_compute_grads runs the batch's D(real) forward beside the Generator backward, and the next
        step_begin / train_step given the same (unmodified) tensor picks it up.  Results do not change."""
        next_real = _f32(next_real, self.device, "real_images")
        s = self.image_size
        if next_real.dim() != 4 or tuple(next_real.shape[1:]) != (1, s, s):
            raise ValueError(f"real_images must be (B, 1, {s}, {s}), got {tuple(next_real.shape)}")
        self._check_batch(next_real.shape[0])
        _lib.check(self.lib.siggan_stage_real(self._h, _ptr(next_real), next_real.shape[0], self._stream()))
        # the tensor itself is kept: identity + version recognise it at the next step, and holding the reference keeps
        # the allocator from handing its address to a different batch in between
        self._staged = (next_real, next_real._version)

    def _take_staged(self, real):
        st, self._staged = self._staged, None
        return st is not None and st[0] is real and st[1] == real._version

    def step_begin(self, real, z=None, masks=None, z_g=None, label_smoothing=0.9):
        """d_compute_grads + the following G step's training forward on its own lane (pipelined
        step); follow with d_apply, then g_compute_grads(batch) WITHOUT z, then g_apply."""
        real = _f32(real, self.device, "real_images")
        b = real.shape[0]
        self._check_batch(b)
        z, z_g = _f32(z, self.device, "noise"), _f32(z_g, self.device, "noise")
        masks = self._masks(masks, b, 2)
        hp = self._hyper(0.0, 0.5, 0.999, label_smoothing=label_smoothing)
        real_ptr = None if self._take_staged(real) else _ptr(real)      # NULL: the batch staged by stage_real
        _lib.check(self.lib.siggan_step_begin(self._h, real_ptr, b, _ptr(z), _ptr(masks), _ptr(z_g), C.byref(hp),
                                              _ptr(self.metrics), self._stream()))

    def train_step(self, real, z_d=None, masks=None, z_g=None, lr_d=2e-4, lr_g=2e-4, beta1=0.5, beta2=0.999, eps=1e-8,
                   label_smoothing=0.9, clip=None, sync=True, next_real=None):
        """One pipelined G+D step (n_critic = 1): same results as d_step followed by g_step.  next_real: the
        following step's real batch, if the loop already has it (see stage_real)."""
        self.step_begin(real, z_d, masks, z_g, label_smoothing)
        dm = self.d_apply(lr_d, beta1, beta2, eps, clip, 1.0, sync)
        if next_real is not None:
            self.stage_real(next_real)
        self.g_compute_grads(real.shape[0])
        gm = self.g_apply(lr_g, beta1, beta2, eps, clip, 1.0, sync)
        if sync:
            dm.update(gm)
        return dm

    def d_apply(self, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8, clip=None, grad_scale=1.0, sync=True):
        hp = self._hyper(lr, beta1, beta2, eps, clip=clip, grad_scale=grad_scale)
        _lib.check(self.lib.siggan_d_apply(self._h, C.byref(hp), _ptr(self.metrics), None, self._stream()))
        m = self._metrics(self.D_KEYS + (("d_grad_norm",) if clip else ()), sync)
        if m is not None and not clip:
            m["d_grad_norm"] = None
        return m

    def d_step(self, real, z=None, masks=None, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8, label_smoothing=0.9,
               clip=None, sync=True):
        self.d_compute_grads(real, z, masks, label_smoothing)
        return self.d_apply(lr, beta1, beta2, eps, clip, 1.0, sync)

    def g_compute_grads(self, batch, z=None, label_smoothing=0.9):
        self._check_batch(batch)
        z = _f32(z, self.device, "noise")
        if z is not None and tuple(z.shape) != (batch, self.latent_dim):
            raise ValueError(f"noise must be ({batch}, {self.latent_dim})")
        hp = self._hyper(0.0, 0.5, 0.999, label_smoothing=label_smoothing)
        _lib.check(self.lib.siggan_g_grads(self._h, batch, _ptr(z), C.byref(hp), _ptr(self.metrics), self._stream()))

    def g_apply(self, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8, clip=None, grad_scale=1.0, sync=True):
        hp = self._hyper(lr, beta1, beta2, eps, clip=clip, grad_scale=grad_scale)
        _lib.check(self.lib.siggan_g_apply(self._h, C.byref(hp), _ptr(self.metrics), None, self._stream()))
        m = self._metrics(self.G_KEYS + (("g_grad_norm",) if clip else ()), sync)
        if m is not None and not clip:
            m["g_grad_norm"] = None
        return m

    def g_step(self, batch, z=None, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8, clip=None, sync=True):
        self.g_compute_grads(batch, z)
        return self.g_apply(lr, beta1, beta2, eps, clip, 1.0, sync)

    # ---- operator-level calls (tests / profiling) --------------------------------------------------
    def op_conv4x4s2(self, form, x_nhwc, w):
        """Activations in / out in the context's element type (``act_dtype``: the input is cast, i.e. rounded to
        nearest, if it is not already); weights fp32 in the torch layout."""
        b, h, _, cin = x_nhwc.shape
        cout = w.shape[0] if form == 0 else w.shape[1]
        ho = h // 2 if form == 0 else 2 * h
        x = x_nhwc.to(self.act_dtype).contiguous()
        out = torch.empty(b, ho, ho, cout, dtype=self.act_dtype, device=self.device)
        _lib.check(self.lib.siggan_op_conv4x4s2(self._h, form, _ptr(x), _ptr(w.float().contiguous()), _ptr(out),
                                                b, h, cin, cout, self._stream()))
        return out

    def op_wgrad(self, small_nhwc, large_nhwc):
        b, hs, _, cs = small_nhwc.shape
        cl = large_nhwc.shape[3]
        dw = torch.empty(cs, cl, 4, 4, dtype=torch.float32, device=self.device)
        sm, lg = small_nhwc.to(self.act_dtype).contiguous(), large_nhwc.to(self.act_dtype).contiguous()
        _lib.check(self.lib.siggan_op_conv4x4s2_wgrad(self._h, _ptr(sm), _ptr(lg), _ptr(dw), b, hs, cs, cl, self._stream()))
        return dw

    def op_adam(self, p, g, m, v, step, lr=2e-4, beta1=0.5, beta2=0.999, eps=1e-8, clip=None, grad_scale=1.0):
        hp = self._hyper(lr, beta1, beta2, eps, clip=clip, grad_scale=grad_scale)
        _lib.check(self.lib.siggan_op_adam(self._h, _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), step, C.byref(hp),
                                           self._stream()))

    def op_randn(self, n):
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.siggan_op_randn(self._h, _ptr(out), n, self._stream()))
        return out

    @staticmethod
    def device_info(index=0):
        """(compute units, shader clock in kHz, HBM bytes) of a device: what the roofline peaks are derived from."""
        cu, khz, mem = C.c_int32(), C.c_int32(), C.c_int64()
        _lib.check(_lib.load().siggan_device_info(int(index), C.byref(cu), C.byref(khz), C.byref(mem)))
        return cu.value, khz.value, mem.value

    def prof_enable(self, on=True):
        _lib.check(self.lib.siggan_prof_enable(self._h, int(on)))

    def prof_read(self):
        """[{name, launches, ms, flops}] for every MFMA kernel slot with launches (synchronises)."""
        out = []
        for i in range(self.lib.siggan_prof_slots()):
            name = C.create_string_buffer(64)
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _lib.check(self.lib.siggan_prof_read(self._h, i, name, 64, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            if n.value:
                out.append({"name": name.value.decode(), "launches": n.value, "ms": ms.value, "flops": fl.value,
                            "bytes": by.value})
        return out

    def debug_tensor(self, name, index, shape):
        """Copy of a workspace tensor (test hook): first prod(shape) floats, reshaped."""
        n = 1
        for s in shape:
            n *= s
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.siggan_debug_tensor(self._h, name.encode(), index, _ptr(out), n, self._stream()))
        return out.view(shape)

    def close(self):
        self._staged = None
        if getattr(self, "_h", None):
            self.lib.siggan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
