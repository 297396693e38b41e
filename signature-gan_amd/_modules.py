"""Shared plumbing of the drop-in ``Generator`` / ``Discriminator`` modules.

The modules are real ``nn.Module`` objects whose parameters and buffers carry the reference's
``state_dict`` keys, but hold no layers: once the module sits on a ROCm device its parameters
are *views* into the Engine's flat arenas (the storage the C ABI borrows), ``.grad`` are views
into the gradient arena, and ``forward`` calls the HIP engine.  There is no CPU or autograd path:
calling ``forward`` on a CPU module raises."""
import torch
import torch.nn as nn

from .engine import Engine


class _Node(nn.Module):
    """Name-only container (gives ``fc.0.weight``-style keys)."""

    def forward(self, *a, **k):   # pragma: no cover
        raise RuntimeError("this container holds parameters only; call the owning model")


def build_tree(root, entries, init):
    """Register every (key, shape, kind) under nested _Node containers of ``root``."""
    for key, shape, kind in entries:
        *path, leaf = key.split(".")
        mod = root
        for name in path:
            if name not in mod._modules:
                mod.add_module(name, _Node())
            mod = mod._modules[name]
        t = init(key, shape, kind)
        if kind == "param":
            mod.register_parameter(leaf, nn.Parameter(t))
        else:
            mod.register_buffer(leaf, t)


def reference_init(key, shape, kind):
    """generator_vanilla_gan.py:168-187 / discriminator_vanilla_gan.py:212-239: weights N(0, 0.02),
    biases 0, BatchNorm weight N(1, 0.02) bias 0; fresh BatchNorm buffers."""
    if kind == "bn_count":
        return torch.zeros((), dtype=torch.int64)
    if kind == "bn_mean":
        return torch.zeros(shape)
    if kind == "bn_var":
        return torch.ones(shape)
    if key.endswith("bias"):
        return torch.zeros(shape)
    return torch.empty(shape).normal_(1.0 if ".1.weight" in key else 0.0, 0.02)


class EngineBacked(nn.Module):
    """Base of the drop-in networks: ``which`` is 'g' or 'd'."""

    which = None

    def _engine_kwargs(self):
        raise NotImplementedError

    def __init__(self):
        super().__init__()
        self._engine = None
        self._shared_engine = False

    # ---- engine attachment -----------------------------------------------------------------
    def _named_leaves(self):
        for name, p in self.named_parameters():
            yield name, p
        for name, b in self.named_buffers():
            yield name, b

    def _attach(self, engine, copy_in=True):
        """Re-point every parameter / buffer at its view inside ``engine``'s arenas."""
        views = dict(engine.views(self.which))
        gviews = dict(engine.views(self.which, "grads"))
        if self.which == "g":
            views.update(engine.bn_views())
        views, gviews = self._remap_views(engine, views, gviews)
        mods = dict(self.named_modules())
        with torch.no_grad():
            for name, t in list(self._named_leaves()):
                v = views[name]
                if copy_in:
                    v.copy_(t.to(v.device))
                owner, leaf = name.rsplit(".", 1)
                m = mods[owner]
                if leaf in m._parameters:
                    p = m._parameters[leaf]
                    p.data = v
                    p.grad = gviews[name]
                else:
                    m._buffers[leaf] = v
        self._engine = engine
        engine.params_changed()

    def _remap_views(self, engine, views, gviews):
        """Hook: modules whose state_dict keys differ from the engine's (spectral norm) rename / add views here."""
        return views, gviews

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        leaves = [t for _, t in self._named_leaves()]
        dev = leaves[0].device
        if dev.type == "cuda":
            eng = self._engine
            if eng is None or eng.device != dev:
                eng = Engine(device=str(dev), **self._engine_kwargs())
                self._shared_engine = False
            self._attach(eng, copy_in=True)
        else:
            self._engine = None if not self._shared_engine else self._engine
        return out

    def _require_engine(self):
        if self._engine is None or next(self.parameters()).device.type != "cuda":
            raise RuntimeError(
                f"{type(self).__name__} runs on the MI355X HIP engine only: move it to a ROCm device "
                "(.to('cuda')) first; there is no CPU path")
        return self._engine

    def load_state_dict(self, state_dict, strict=True, assign=False):
        if assign:
            raise ValueError("assign=True would detach the parameters from the engine's arenas")
        out = super().load_state_dict(state_dict, strict=strict)
        if self._engine is not None:
            self._engine.params_changed()
        return out

    def get_num_params(self):
        """Total number of trainable parameters (reference: get_num_params)."""
        return sum(p.numel() for p in self.parameters() if p.requires_grad)


class EngineAdam(torch.optim.Adam):
    """torch.optim.Adam whose state lives in the Engine's flat arenas (so ``state_dict()`` is the
    reference's optimizer checkpoint format) and whose ``step()`` is the fused HIP update."""

    def __init__(self, module, lr, betas):
        params = list(module.parameters())
        super().__init__(params, lr=lr, betas=betas)
        self._module = module
        self._populated_for = None

    def _sync_views(self):
        eng = self._module._engine
        if eng is None or self._populated_for is eng:
            return eng
        w = self._module.which
        m, v = eng.views(w, "exp_avg"), eng.views(w, "exp_avg_sq")
        steps_flat = getattr(eng, f"{w}_adam_steps")
        order = list(m)                                       # the engine's tensor order (index into *_adam_steps)
        ename = getattr(self._module, "_engine_name", lambda n: n)
        names = [n for n, _ in self._module.named_parameters()]
        m = {n: m[ename(n)] for n in names}; v = {n: v[ename(n)] for n in names}
        steps = [steps_flat[order.index(ename(n))] for n in names]
        old = {n: self.state.get(p) for n, p in zip(names, self._module.parameters())}
        self.state.clear()
        with torch.no_grad():
            for i, (n, p) in enumerate(zip(names, self._module.parameters())):
                st = {"step": steps[i], "exp_avg": m[n], "exp_avg_sq": v[n]}
                if old[n]:                                    # carry over a state loaded on the CPU
                    st["step"].copy_(torch.as_tensor(old[n]["step"], dtype=torch.float32))
                    st["exp_avg"].copy_(old[n]["exp_avg"]); st["exp_avg_sq"].copy_(old[n]["exp_avg_sq"])
                self.state[p] = st
        self._populated_for = eng
        if any(old.values()):
            eng.params_changed()                              # step tensors were written: the library re-reads them
        return eng

    def state_dict(self):
        self._sync_views()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        eng = self._sync_views()
        if eng is None:
            return super().load_state_dict(state_dict)
        groups = state_dict["param_groups"]
        for g, new in zip(self.param_groups, groups):
            for k, val in new.items():
                if k != "params":
                    g[k] = val
        params = list(self._module.parameters())
        with torch.no_grad():
            for idx, st in state_dict["state"].items():
                mine = self.state[params[int(idx)]]
                mine["step"].copy_(torch.as_tensor(st["step"], dtype=torch.float32))
                mine["exp_avg"].copy_(st["exp_avg"]); mine["exp_avg_sq"].copy_(st["exp_avg_sq"])
            if not state_dict["state"]:
                for mine in self.state.values():
                    mine["step"].zero_(); mine["exp_avg"].zero_(); mine["exp_avg_sq"].zero_()
        eng.params_changed()                                  # the step tensors were written: the library re-reads them (siggan.h)

    def zero_grad(self, set_to_none=True):
        eng = self._module._engine
        if eng is not None:
            getattr(eng, f"{self._module.which}_grads").zero_()

    def hyper(self):
        g = self.param_groups[0]
        return dict(lr=float(g["lr"]), beta1=float(g["betas"][0]), beta2=float(g["betas"][1]), eps=float(g["eps"]))

    @torch.no_grad()
    def step(self, closure=None):
        """Apply the fused HIP Adam update to the gradient the engine just produced
        (Engine.*_compute_grads); raises if there is none."""
        eng = self._sync_views()
        if eng is None:
            raise RuntimeError("EngineAdam.step needs the module on a ROCm device")
        h = self.hyper()
        apply = eng.g_apply if self._module.which == "g" else eng.d_apply
        apply(h["lr"], h["beta1"], h["beta2"], h["eps"], sync=False)
