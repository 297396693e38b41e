"""The fully-connected GAN extension (include/siggan_mlp.h; BASELINE.json configs[0]) on the MI355X against the build's own
CPU restatement (oracle/mlp_oracle.py).  PARITY UNPINNED: the reference has no such model, so these tests show that the HIP
path computes what the extension is defined to compute, nothing about the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("layout,m,n,k", [("NT", 32, 256, 100), ("NT", 64, 784, 512), ("NT", 5, 1, 256), ("NN", 32, 512, 784),
                                          ("NN", 64, 100, 256), ("TN", 784, 512, 32), ("TN", 256, 100, 64), ("TN", 1, 256, 37)])
def test_gemm_layouts(layout, m, n, k):
    from signature_gan_amd.mlp_gan import op_gemm
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    if layout == "NT":
        a, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g); want = a @ b.t()
    elif layout == "NN":
        a, b = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g); want = a @ b
    else:
        a, b = torch.randn(k, m, generator=g), torch.randn(k, n, generator=g); want = a.t() @ b
    got = op_gemm(layout, a.cuda(), b.cuda()).cpu()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-6


def _states(model, seed):
    from oracle import mlp_oracle as M
    from oracle.siggan_oracle import AdamState
    gen = torch.Generator().manual_seed(seed)
    g_sd, d_sd = {}, {}
    for which, sd in (("g", g_sd), ("d", d_sd)):
        for k, v in model.views(which).items():
            t = torch.empty(v.shape).normal_(1.0 if ".bn.weight" in k else 0.0, 0.05, generator=gen)
            v.copy_(t); sd[k] = t.clone()
    for k, v in model.bn_views().items():
        if "num_batches" in k:
            g_sd[k] = torch.zeros((), dtype=torch.int64)
        else:
            t = torch.rand(v.shape, generator=gen) * 0.5 + (0.75 if "var" in k else -0.25)
            v.copy_(t); g_sd[k] = t.clone()
    return g_sd, d_sd, AdamState(list(model.g_spans), g_sd), AdamState(list(model.d_spans), d_sd), M


@pytest.mark.parametrize("size,hidden,batch", [(28, (256, 512), 32), (64, (256, 512, 1024), 64), (28, (128,), 5)])
def test_mlp_steps_vs_own_oracle(size, hidden, batch):
    """configs[0] (28x28, 100-256-512-784, batch 32) and two other geometries: forward, one D step, one G step -- losses,
    predictions, gradients, BatchNorm buffers, updated weights -- against oracle/mlp_oracle.py (parity unpinned)."""
    from signature_gan_amd.mlp_gan import MLPGAN
    m = MLPGAN(latent_dim=100, image_size=size, hidden=hidden, max_batch=batch, device="cuda:0", seed=1)
    g_sd, d_sd, g_opt, d_opt, M = _states(m, 7)
    gen = torch.Generator().manual_seed(3)
    z, z2 = torch.randn(batch, 100, generator=gen), torch.randn(batch, 100, generator=gen)
    real = torch.rand(batch, 1, size, size, generator=gen) * 2 - 1

    img = m.generate(z.cuda()).cpu()
    want = M.g_forward(dict(g_sd), z, hidden, size, training=False)
    assert float((img - want).abs().max()) <= 2e-4
    p = m.discriminate(real.cuda()).cpu()
    assert float((p - M.d_forward(d_sd, real, len(hidden))).abs().max()) <= 2e-5

    met = m.train_discriminator_step(real.cuda(), noise=z.cuda())
    o_met, o_grads = M.d_step(g_sd, d_sd, d_opt, real, z, hidden, size)
    for k, v in o_met.items():
        assert abs(met[k] - v) <= 2e-4 * abs(v) + 2e-6, (k, met[k], v)
    gs = max(float(g.abs().max()) for g in o_grads.values())
    for k, g in m.views("d", "grads").items():
        assert float((g.cpu() - o_grads[k]).abs().max()) <= 2e-4 * max(float(o_grads[k].abs().max()), 1e-3 * gs), k
    for k, w in m.views("d").items():
        assert float((w.cpu() - d_sd[k]).abs().max()) <= 2.5 * 2e-4, k

    met = m.train_generator_step(batch, noise=z2.cuda())
    o_met, o_grads = M.g_step(g_sd, d_sd, g_opt, z2, hidden, size)
    for k, v in o_met.items():
        assert abs(met[k] - v) <= 2e-4 * abs(v) + 2e-6, (k, met[k], v)
    gs = max(float(g.abs().max()) for g in o_grads.values())
    for k, g in m.views("g", "grads").items():
        # (a ReLU whose pre-activation is within rounding of zero may fall on either side: bounded as in the conv tests)
        assert float((g.cpu() - o_grads[k]).abs().max()) <= 5e-3 * max(float(o_grads[k].abs().max()), 1e-2 * gs), k
    for k, v in m.bn_views().items():
        ref = g_sd[k].float()
        assert float((v.float().cpu() - ref).abs().max()) <= 2e-4 * float(ref.abs().max()) + 1e-6, k
    # a few more steps from the library RNG: finite, losses move
    for _ in range(3):
        t = m.train_step(real.cuda())
    assert all(np.isfinite(v) for v in t.values())
    m.close()
