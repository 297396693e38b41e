"""Operator-level parity on the MI355X: each MFMA / elementwise kernel against the same op
evaluated by torch on the CPU in fp32 (tolerance: fp32 accumulation-order noise, rtol 1e-4 of the
output scale unless stated)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from hipcommon import Engine
    e = Engine(latent_dim=100, image_size=64, max_batch=8, device="cuda:0", seed=1)
    yield e
    e.close()


def _close(got, want, rtol=2e-4, what=""):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = want.abs().max().item() + 1e-30
    err = (got - want).abs().max().item()
    assert err <= rtol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


# (batch, h_in, c_in, c_out) -- every conv layer shape of both model sizes, plus ragged batches
DOWN = [(2, 32, 64, 128), (3, 16, 128, 256), (5, 8, 256, 512), (2, 8, 512, 512), (1, 64, 64, 128),
        (2, 8, 128, 256), (2, 16, 64, 128), (2, 32, 32, 64), (1, 64, 32, 32), (128, 32, 64, 128),
        # 256 tiles of 64x64: the two-way K split that runs inside 8-wave workgroups (no slabs)
        (32, 32, 64, 128)]
UP = [(2, 4, 256, 128), (3, 8, 128, 64), (2, 16, 64, 32), (1, 32, 32, 32), (2, 4, 512, 256), (1, 64, 32, 32),
      (3, 4, 512, 256), (2, 8, 256, 128), (2, 16, 128, 64), (64, 16, 128, 64),
      # the in-workgroup two-way K split in the "up" form (64 tiles x 4 classes); four K-tiles only (the short, one-tile-ahead loop)
      (64, 8, 128, 64), (2, 16, 32, 64),
      # large enough for k_gconv_up4 (all four parity classes per workgroup, input patch in LDS): row widths 16 / 32 / 64,
      # 32 and 64 input channels
      (192, 16, 32, 32), (48, 32, 32, 32), (12, 64, 32, 32), (96, 32, 64, 32), (24, 64, 64, 32)]


@pytest.mark.parametrize("b,h,ci,co", DOWN)
def test_conv_down(eng, b, h, ci, co):
    g = torch.Generator().manual_seed(b * 1000 + h + ci + co)
    x = torch.randn(b, ci, h, h, generator=g)
    w = torch.randn(co, ci, 4, 4, generator=g) * 0.05
    want = F.conv2d(x, w, None, stride=2, padding=1)
    got = eng.op_conv4x4s2(0, x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda())
    _close(got.permute(0, 3, 1, 2), want, what="down")


@pytest.mark.parametrize("b,h,ci,co", UP)
def test_conv_up(eng, b, h, ci, co):
    g = torch.Generator().manual_seed(b * 1000 + h + ci + co + 7)
    x = torch.randn(b, ci, h, h, generator=g)
    w = torch.randn(ci, co, 4, 4, generator=g) * 0.05
    want = F.conv_transpose2d(x, w, None, stride=2, padding=1)
    got = eng.op_conv4x4s2(1, x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda())
    _close(got.permute(0, 3, 1, 2), want, what="up")


@pytest.mark.parametrize("b,hs,cs,cl", [(2, 16, 128, 64), (3, 8, 256, 128), (2, 4, 512, 256), (2, 4, 256, 128),
                                        (2, 8, 128, 64), (2, 16, 64, 32), (1, 32, 32, 32), (1, 4, 512, 512),
                                        (64, 16, 128, 64), (1, 4, 32, 32)])
def test_conv_wgrad(eng, b, hs, cs, cl):
    """dw[cs,cl,kh,kw] = sum small[n,p,q,cs] * large[n,2p-1+kh,2q-1+kw,cl]  == Conv2d weight-grad with
    (out=cs, in=cl) == ConvTranspose2d weight-grad with (in=cs, out=cl)."""
    g = torch.Generator().manual_seed(b + hs + cs + cl)
    small = torch.randn(b, cs, hs, hs, generator=g)
    large = torch.randn(b, cl, 2 * hs, 2 * hs, generator=g)
    w = torch.zeros(cs, cl, 4, 4, requires_grad=True)
    (F.conv2d(large, w, None, stride=2, padding=1) * small).sum().backward()
    got = eng.op_wgrad(small.permute(0, 2, 3, 1).contiguous().cuda(), large.permute(0, 2, 3, 1).contiguous().cuda())
    _close(got, w.grad, what="wgrad")


@pytest.mark.parametrize("n,step,clip", [(1000003, 1, None), (4096, 6, None), (77, 3, 0.5), (300001, 2, 1e-3)])
def test_adam(eng, n, step, clip):
    from common import O
    g = torch.Generator().manual_seed(n)
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-2
    m, v = torch.randn(n, generator=g) * 1e-3, torch.rand(n, generator=g) * 1e-5
    if step == 1:
        m.zero_(); v.zero_()
    dp, dg, dm, dv = p.cuda(), gr.cuda(), m.cuda(), v.cuda()
    eng.op_adam(dp, dg, dm, dv, step, clip=clip)
    if clip is not None:
        O.clip_grad_norm([gr], clip)
        _close(dg, gr, 1e-5, "clipped grad written back")
    O.adam_update(p, gr, m, v, step, 2e-4, 0.5, 0.999)
    _close(dm, m, 1e-5, "exp_avg"); _close(dv, v, 1e-5, "exp_avg_sq")
    assert (dp.cpu() - p).abs().max().item() <= 2e-7 + 1e-6 * 2e-4, "param"


def test_randn_moments(eng):
    x = eng.op_randn(1 << 20).cpu().double()
    y = eng.op_randn(1 << 20).cpu().double()
    assert abs(x.mean().item()) < 5e-3 and abs(x.std().item() - 1) < 5e-3
    assert abs((x ** 4).mean().item() - 3) < 0.1
    assert (x - y).abs().max().item() > 1.0, "successive calls must differ"
    assert torch.isfinite(x).all()


# ---- 16-bit operand kernels (bf16 / f16 storage + MFMA operands, fp32 accumulate; BASELINE configs[2] / [4]) --------------
# The check is exact up to fp32 accumulation order: the CPU evaluates the same op in fp32 on the inputs ROUNDED to the
# narrow type (what the kernel reads).  Weight gradients are fp32 results (rtol 2e-4 of scale, as for the fp32 kernels);
# conv outputs are stored in the narrow type, so each element additionally carries one rounding of the output (half an ulp:
# 2^-9 relative for bf16, 2^-12 for f16).
NARROW = {"bf16": (torch.bfloat16, 2.0 ** -8), "f16": (torch.float16, 2.0 ** -11)}


@pytest.fixture(scope="module", params=["bf16", "f16"])
def eng16(request):
    from hipcommon import Engine
    e = Engine(latent_dim=100, image_size=64, max_batch=8, device="cuda:0", seed=1, dtype=request.param)
    yield e
    e.close()


def _close16(got, want, ulp, what):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = want.abs().max().item() + 1e-30
    tol = ulp * want.abs() + 2e-4 * scale
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} elements off, worst {(got - want).abs().max().item():.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("b,h,ci,co", DOWN)
def test_conv_down_16(eng16, b, h, ci, co):
    td, ulp = NARROW[eng16.dtype]
    g = torch.Generator().manual_seed(b * 1000 + h + ci + co)
    x = torch.randn(b, ci, h, h, generator=g).to(td).float()
    w = (torch.randn(co, ci, 4, 4, generator=g) * 0.05)
    want = F.conv2d(x, w.to(td).float(), None, stride=2, padding=1)
    got = eng16.op_conv4x4s2(0, x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda())
    assert got.dtype == td
    _close16(got.float().permute(0, 3, 1, 2), want, ulp, f"down {eng16.dtype}")


@pytest.mark.parametrize("b,h,ci,co", UP)
def test_conv_up_16(eng16, b, h, ci, co):
    td, ulp = NARROW[eng16.dtype]
    g = torch.Generator().manual_seed(b * 1000 + h + ci + co + 7)
    x = torch.randn(b, ci, h, h, generator=g).to(td).float()
    w = (torch.randn(ci, co, 4, 4, generator=g) * 0.05)
    want = F.conv_transpose2d(x, w.to(td).float(), None, stride=2, padding=1)
    got = eng16.op_conv4x4s2(1, x.permute(0, 2, 3, 1).contiguous().cuda(), w.cuda())
    _close16(got.float().permute(0, 3, 1, 2), want, ulp, f"up {eng16.dtype}")


@pytest.mark.parametrize("b,hs,cs,cl", [(2, 16, 128, 64), (3, 8, 256, 128), (2, 4, 512, 256), (2, 16, 64, 32), (1, 32, 32, 32),
                                        (1, 4, 512, 512), (64, 16, 128, 64), (1, 4, 32, 32), (5, 8, 64, 32)])
def test_conv_wgrad_16(eng16, b, hs, cs, cl):
    td, _ = NARROW[eng16.dtype]
    g = torch.Generator().manual_seed(b + hs + cs + cl)
    small = torch.randn(b, cs, hs, hs, generator=g).to(td).float()
    large = torch.randn(b, cl, 2 * hs, 2 * hs, generator=g).to(td).float()
    w = torch.zeros(cs, cl, 4, 4, requires_grad=True)
    (F.conv2d(large, w, None, stride=2, padding=1) * small).sum().backward()
    got = eng16.op_wgrad(small.permute(0, 2, 3, 1).contiguous().cuda(), large.permute(0, 2, 3, 1).contiguous().cuda())
    assert got.dtype == torch.float32
    _close(got, w.grad, what=f"wgrad {eng16.dtype}")


def test_wgrad_16_fragment_map_exact(eng16):
    """Exact-integer data with an asymmetric pattern: a wrong lane <-> element map of the transposing LDS reads (or a
    swapped operand) cannot produce the right integers."""
    b, hs, cs, cl = 1, 8, 64, 32
    n = torch.arange(b * hs * hs * cs, dtype=torch.float32)
    small = ((n * 7 + 3) % 5 - 2).view(b, hs, hs, cs)                      # values in {-2..2}, NHWC
    m = torch.arange(b * 4 * hs * hs * cl, dtype=torch.float32)
    large = ((m * 11 + 1) % 7 - 3).view(b, 2 * hs, 2 * hs, cl)             # values in {-3..3}
    w = torch.zeros(cs, cl, 4, 4, requires_grad=True)
    (F.conv2d(large.permute(0, 3, 1, 2), w, None, stride=2, padding=1) * small.permute(0, 3, 1, 2)).sum().backward()
    got = eng16.op_wgrad(small.cuda(), large.cuda())
    assert torch.equal(got.cpu(), w.grad)
