"""Which stored-16-bit tensor costs the bf16 gradient fidelity?  CPU only (no GPU, no reference needed).

    python profiles/bf16_rounding_sites.py SIZE LATENT BATCH > profiles/r03_bf16_rounding_sites_s<SIZE>_b<BATCH>.txt

The storage-rounding oracle (oracle.Quant: fp32 arithmetic, values rounded to bf16 exactly where the HIP path stores 16-bit
tensors -- 'a': a stored activation, 'g': a stored activation gradient, 'w': the weight copy an MFMA convolution reads) is run
on the G step with single rounding sites, or a whole class of them, switched off, and its gradients are compared with the fp32
oracle's: relative L2 error of the whole arena ('all') and per parameter (listed when > 0.05).  It answers the round-2
verdict's question (weak #5): which tensor is 42 % off, and would keeping some activation gradient in fp32 fix it."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import torch, numpy as np
from common import *

class SiteQuant(O.Quant):
    """Quant that numbers its call sites (in call order, per kind) and can skip some."""
    def __init__(self, dtype, skip=()):
        super().__init__(dtype, 1.0)
        self.skip = set(skip); self.n = {"a": 0, "g": 0, "w": 0}; self.log = []
    def _site(self, kind, x, fn):
        i = self.n[kind]; self.n[kind] += 1
        self.log.append((kind, i, tuple(x.shape)))
        return x if (kind, i) in self.skip else fn(x)
    def a(self, x): return self._site("a", x, super().a)
    def g(self, x): return self._site("g", x, super().g)
    def w(self, w): return self._site("w", w, super().w)

def rel(a, b): return float((a - b).norm() / (b.norm() + 1e-30))

size, latent, batch = [int(v) for v in sys.argv[1:4]]
dt = torch.bfloat16
z2 = torch.from_numpy(I.gen_z(batch, latent, SEED["z"] + 1))
def run(q):
    g_sd, d_sd, g_opt, d_opt = oracle_states(size, latent, warm=True)
    met, grads, _, _ = O.g_grads(dict(g_sd), d_sd, z2, size, q=q)
    return grads
ref = run(None)
q = SiteQuant(dt); full = run(q)
names = list(ref)
print("sites:", [(k, i, s) for k, i, s in q.log])
def report(tag, g):
    errs = {k: rel(g[k], ref[k]) for k in names}
    allr = float(torch.cat([(g[k] - ref[k]).reshape(-1) for k in names]).norm() / torch.cat([ref[k].reshape(-1) for k in names]).norm())
    worst = max(errs, key=errs.get)
    print(f"{tag:28s} all {allr:.4f}  worst {worst} {errs[worst]:.4f}   " + " ".join(f"{k.split('.')[0][:2]}{k.split('.')[1] if k[0]=='u' else ''}.{k.split('.')[-2]}{k.split('.')[-1][0]}={v:.3f}" for k, v in errs.items() if v > 0.05))
report("all sites rounded", full)
for kind in ("a", "g", "w"):
    report(f"no {kind} sites at all", run(SiteQuant(dt, skip=[(kind, i) for i in range(40)])))
for kind, cnt in q.n.items():
    for i in range(cnt):
        report(f"skip {kind}{i}", run(SiteQuant(dt, skip=[(kind, i)])))
