import sys
sys.path.insert(0, "/root/repo")
import torch
import signature_gan_amd
from signature_gan_amd.engine import Engine
B = 64
eng = Engine(latent_dim=100, image_size=64, max_batch=B, device="cuda:0", seed=1, dtype=__import__("os").environ.get("TRACE_DTYPE", "f32"))
eng.init_reference(0)
import os

real = (torch.rand(B, 1, 64, 64, device="cuda") * 2 - 1)
for _ in range(40):
    eng.train_step(real, sync=False, next_real=real)
torch.cuda.synchronize()
