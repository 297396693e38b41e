"""The bf16 / f16 variants of the HIP path (BASELINE.json configs[2] / configs[4]) against the oracle.

The reference is fp32-only (vanilla_gan_model.py:107-120: no autocast / half anywhere), so a narrow path has no reference
output of its own.  Two comparisons, both through the C ABI on the per-GPU workloads of those configs -- (64, z=100, B=64)
and (128, z=128, B=32) -- plus a small case:

  (1) vs the fp32 oracle (the reference's arithmetic): the bar a user cares about.  16-bit storage of ~10 chained layers
      carries 2^-9 (bf16) / 2^-12 (f16) relative rounding per stored element; summed over a layer's fan-in the errors add in
      quadrature, and the backward chain doubles the depth.  Tolerances below are ~3x what was measured on the MI355X
      (profiles/r02_parity_margins.json) and are stated relative to each tensor's scale.
  (2) vs the oracle with the SAME storage roundings modelled (oracle.Quant: stored activations, stored activation
      gradients and the MFMA weight copies rounded to the narrow type, all arithmetic fp32) AND handed the HIP path's
      (Leaky)ReLU sign decisions, as the fp32 parity tests do: two implementations that round to 16 bits at every layer drift
      apart by ~1e-3 of a layer's scale, so ~1e-4 of the pre-activations land on different sides of zero, and each such
      decision changes a gradient element by a factor of 5 (LeakyReLU) or switches it off (ReLU) -- measured, that alone is
      0.5-1 % of a gradient's L2 norm in BOTH narrow types, i.e. it hides the arithmetic.  With the decisions shared (their
      number and their distance from zero are bounded below) what is left is fp32 summation order plus elements whose
      pre-rounding value straddles a rounding boundary -- an order of magnitude tighter than (1), tight enough to expose a
      wrong kernel that (1)'s looser bar could hide.
"""
import json
import os

import numpy as np
import pytest
import torch

from common import I, O, SEED, d_chans, oracle_states

pytestmark = pytest.mark.gpu

CASES = [(64, 100, 4), (64, 100, 64), (128, 128, 32), (64, 100, 5),      # (64, 100, 5): an odd batch (ragged row tiles)
         (64, 50, 8)]       # latent_dim % 4 != 0: the Generator fc's generic kernels in 16-bit storage
TORCH_T = {"bf16": torch.bfloat16, "f16": torch.float16}
# metric: relative (losses, mean predictions); image: absolute (pixels in [-1, 1]); grad_t: relative L2 error of the worst
# parameter tensor (NAMED in the report: it is final_conv.0.bias -- one scalar, a near-cancelling sum over every pixel -- at
# the large cases, a bias / fc weight at the small ones); grad_all: relative L2 error of the whole gradient arena; bn: BatchNorm
# running statistics, relative to scale.
# (1) vs the fp32 oracle, free-running: 1.6 x the largest value measured on the MI355X over the four cases
# (profiles/r03_narrow_parity.json: bf16 metric 1.02e-2, image 4.7e-3, grad_t 0.259, grad_all 0.130, bn 8.9e-4 -- the five-sample batch; f16 7e-4, 6e-4,
# 0.0945, 0.0463, 1e-4).  Forward quantities track the storage precision (8x between the types); the gradients are ~2.7x apart:
# they are dominated by what 16-bit operands do to heavily cancelling sums, activation storage and weight copies alike, and
# NOT by the stored activation gradients (profiles/bf16_rounding_sites.py).
TOL_FP32 = {"bf16": dict(metric=1.6e-2, image=7.5e-3, grad_t=0.42, grad_all=0.21, bn=1.45e-3),
            "f16": dict(metric=1.2e-3, image=1e-3, grad_t=0.15, grad_all=0.075, bn=1.6e-4)}
# (2) vs the storage-rounding oracle with shared sign decisions: 8x apart, as the precisions are.
# (bn: ONE pre-BatchNorm element that rounds the other way moves a 4-sample feature variance by ~1 ulp / 4: the bound is one
# such element, not the ~1e-6 the statistics agree to otherwise)
TOL_Q = {"bf16": dict(metric=1.5e-2, image=6e-3, grad_t=3e-2, grad_all=2e-2, bn=8e-4),
         "f16": dict(metric=2e-3, image=1.2e-3, grad_t=5e-3, grad_all=2.5e-3, bn=1e-4)}
# sign decisions that may differ between the HIP path and the Quant oracle: at most this fraction of a layer's elements,
# each with |pre-activation| below this fraction of the layer's largest
SIGN_FRAC, SIGN_DIST = {"bf16": 5e-3, "f16": 8e-4}, {"bf16": 1.5e-2, "f16": 2.5e-3}


def _sign_stats(signs, recorded, keep=None):
    """(largest fraction of differing sign decisions in a layer, largest |x| / max|x| among them)"""
    frac, dist = 0.0, 0.0
    for i, (s_, x) in enumerate(zip(signs, recorded)):
        bad = s_.reshape(x.shape) != (x > 0)
        if keep is not None and keep[i] is not None:
            bad &= keep[i][:, :, None, None] > 0
        if bad.any():
            frac = max(frac, float(bad.float().mean()))
            dist = max(dist, float(x[bad].abs().max()) / float(x.abs().max()))
    return frac, dist
REPORT = {}


def _rel_l2(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def _grad_errs(views, ref):
    per = {k: _rel_l2(views[k].cpu(), ref[k]) for k in views}
    # tensors whose gradient is a negligible part of the arena (e.g. a conv bias that BatchNorm cancels) are pure noise
    tot = float(torch.cat([g.reshape(-1) for g in ref.values()]).double().norm())
    per = {k: v for k, v in per.items() if float(ref[k].double().norm()) > 1e-4 * tot}
    flat_got = torch.cat([views[k].cpu().reshape(-1) for k in views])
    flat_ref = torch.cat([ref[k].reshape(-1) for k in views])
    worst = max(per, key=per.get)
    return per[worst], _rel_l2(flat_got, flat_ref), worst, {k: round(v, 5) for k, v in per.items()}


def _metric_err(got, ref, keys):
    return max(abs(got[k] - ref[k]) / (abs(ref[k]) + 1e-3) for k in keys)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("size,latent,batch", CASES)
def test_narrow_steps(dtype, size, latent, batch):
    from hipcommon import cuda, hip_signs_d, hip_signs_g, make_engine
    z = torch.from_numpy(I.gen_z(batch, latent, SEED["z"]))
    z2 = torch.from_numpy(I.gen_z(batch, latent, SEED["z"] + 1))
    real = torch.from_numpy(I.gen_real(batch, size, SEED["real"]))
    masks = [torch.from_numpy(m) for m in I.gen_masks(batch, d_chans(size) * 2, 5)]
    nb = len(masks) // 2
    row = {}

    eng = make_engine(size, latent, batch, warm=True, dtype=dtype)
    gs = 1024.0 if dtype == "f16" else 1.0
    q = O.Quant(TORCH_T[dtype], gs)

    # ---- generation (G eval) ----
    img = eng.g_forward(cuda(z), training=False).cpu()
    g_sd, d_sd, _, _ = oracle_states(size, latent, warm=True)
    row["image_vs_fp32"] = float((img - O.g_forward(g_sd, z, False, size)).abs().max())
    with torch.no_grad():
        row["image_vs_q"] = float((img - O.g_forward(g_sd, z, False, size, q=q)).abs().max())

    # ---- D step ----
    met = eng.d_step(cuda(real), cuda(z), masks)
    signs = hip_signs_d(eng, size, batch, 2)
    dkeys = ("d_loss", "d_loss_real", "d_loss_fake", "d_real_mean", "d_fake_mean")
    for tag, qq in (("fp32", None), ("q", q)):
        g_sd, d_sd, g_opt, d_opt = oracle_states(size, latent, warm=True)
        rec = []
        o_met, o_grads = O.d_step(g_sd, d_sd, d_opt, real, z, masks[:nb], masks[nb:], size, q=qq,
                                  signs=signs if qq else None, record=rec)
        if qq:
            row["d_sign_frac"], row["d_sign_dist"] = _sign_stats(signs, rec, keep=masks)
        row[f"d_metric_vs_{tag}"] = _metric_err(met, o_met, dkeys)
        row[f"d_grad_t_vs_{tag}"], row[f"d_grad_all_vs_{tag}"], row[f"d_grad_worst_tensor_vs_{tag}"], per = _grad_errs(eng.views("d", "grads"), o_grads)
        if not qq:
            row["d_grad_per_tensor_vs_fp32"] = per
        row[f"d_exp_avg_vs_{tag}"] = max(_rel_l2(v.cpu(), d_opt.m[k]) for k, v in eng.views("d", "exp_avg").items())
    eng.close()

    # ---- G step (fresh engine: same starting state as the oracle's) ----
    eng = make_engine(size, latent, batch, warm=True, dtype=dtype)
    met = eng.g_step(batch, cuda(z2))
    signs = hip_signs_g(eng, size, batch) + hip_signs_d(eng, size, batch, 1)
    for tag, qq in (("fp32", None), ("q", q)):
        g_sd, d_sd, g_opt, d_opt = oracle_states(size, latent, warm=True)
        rec = []
        o_met, o_grads = O.g_step(g_sd, d_sd, g_opt, z2, size, q=qq, signs=signs if qq else None, record=rec)
        if qq:
            row["g_sign_frac"], row["g_sign_dist"] = _sign_stats(signs, rec)
        row[f"g_metric_vs_{tag}"] = _metric_err(met, o_met, ("g_loss", "g_fake_mean"))
        row[f"g_grad_t_vs_{tag}"], row[f"g_grad_all_vs_{tag}"], row[f"g_grad_worst_tensor_vs_{tag}"], per = _grad_errs(eng.views("g", "grads"), o_grads)
        if not qq:
            row["g_grad_per_tensor_vs_fp32"] = per
        bn = eng.bn_views()
        row[f"bn_vs_{tag}"] = max(float((t.float().cpu() - g_sd[k].float()).abs().max() / (g_sd[k].float().abs().max() + 1e-30))
                                  for k, t in bn.items() if "num_batches" not in k)
        assert all(int(t) == int(g_sd[k]) for k, t in bn.items() if "num_batches" in k)
    steps = eng.g_adam_steps.cpu()
    assert float(steps.min()) == float(steps.max()) == g_opt.step
    eng.close()

    REPORT[f"{dtype}/s{size}_b{batch}"] = row
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "narrow_parity.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)

    for tag, tol in (("fp32", TOL_FP32[dtype]), ("q", TOL_Q[dtype])):
        assert row[f"image_vs_{tag}"] <= tol["image"], (tag, "image", row)
        for net in ("d", "g"):
            assert row[f"{net}_metric_vs_{tag}"] <= tol["metric"], (tag, net, "metric", row)
            assert row[f"{net}_grad_t_vs_{tag}"] <= tol["grad_t"], (tag, net, "grad_t", row)
            assert row[f"{net}_grad_all_vs_{tag}"] <= tol["grad_all"], (tag, net, "grad_all", row)
        assert row[f"bn_vs_{tag}"] <= tol["bn"], (tag, "bn", row)
    for net in ("d", "g"):
        assert row[f"{net}_sign_frac"] <= SIGN_FRAC[dtype] and row[f"{net}_sign_dist"] <= SIGN_DIST[dtype], (net, "signs", row)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_narrow_training_runs_and_matches_modes(dtype):
    """A few pipelined steps in a narrow type: finite, losses move like the fp32 run's, and the execution modes
    (overlap on / off, pipelined or not) stay bit-identical to each other within the type."""
    from hipcommon import cuda, make_engine
    size, latent, batch = 64, 100, 16
    real = cuda(torch.from_numpy(I.gen_real(batch, size, SEED["real"])))
    masks = [torch.from_numpy(m) for m in I.gen_masks(batch, d_chans(size) * 2, 3)]

    def run(dt, overlap, pipelined):
        eng = make_engine(size, latent, batch, warm=True, dtype=dt)
        eng.set_mode(graph=False, overlap=overlap)
        mets = []
        for s in range(4):
            z1 = cuda(torch.from_numpy(I.gen_z(batch, latent, 50 + s)))
            z2 = cuda(torch.from_numpy(I.gen_z(batch, latent, 60 + s)))
            if pipelined:
                mets.append(eng.train_step(real, z1, masks, z2, clip=0.5))
            else:
                m = eng.d_step(real, z1, masks, clip=0.5)
                m.update(eng.g_step(batch, z2, clip=0.5))
                mets.append(m)
        state = [t.clone() for t in (eng.g_params, eng.d_params, eng.g_exp_avg_sq, eng.d_exp_avg, eng.g_bn_mean, eng.g_bn_var)]
        eng.close()
        return mets, state

    ref_m, ref_s = run(dtype, False, False)
    for overlap, pipelined in ((True, False), (True, True), (False, True)):
        m, s = run(dtype, overlap, pipelined)
        assert m == ref_m, (overlap, pipelined)
        for a, b in zip(ref_s, s):
            assert torch.equal(a, b)
    f32_m, _ = run("f32", True, True)
    tol = 5e-2 if dtype == "bf16" else 1e-2
    for a, b in zip(ref_m, f32_m):
        for k in ("d_loss", "g_loss", "d_real_mean", "d_fake_mean"):
            assert np.isfinite(a[k]) and abs(a[k] - b[k]) <= tol * (abs(b[k]) + 0.1), (k, a[k], b[k])


def test_f16_overflow_skips_the_update():
    """fp16 carries a STATIC gradient scale (include/siggan.h, f16_grad_scale): an activation gradient that overflows reaches
    the weight gradients as inf / NaN.  *_apply must then skip the whole update -- parameters, both moments and the step
    counts bit-identical -- raise the skipped flag in the metrics, and the following step must run normally again."""
    from hipcommon import cuda, make_engine
    from signature_gan_amd import _lib
    size, latent, batch = 64, 100, 8
    eng = make_engine(size, latent, batch, warm=True, dtype="f16")
    real = cuda(torch.from_numpy(I.gen_real(batch, size, SEED["real"])))
    z = cuda(torch.from_numpy(I.gen_z(batch, latent, SEED["z"])))
    for which, grads, apply, flag in (("d", lambda: eng.d_compute_grads(real, z), eng.d_apply, "d_skipped"),
                                      ("g", lambda: eng.g_compute_grads(batch, z), eng.g_apply, "g_skipped")):
        state = lambda: [getattr(eng, f"{which}_{a}").clone() for a in ("params", "exp_avg", "exp_avg_sq", "adam_steps")]
        grads()
        before = state()
        getattr(eng, f"{which}_grads")[1234] = float("inf")
        apply(clip=0.5)
        assert float(eng.metrics[_lib.METRIC_INDEX[flag]]) == 1.0
        for a, b in zip(before, state()):
            assert torch.equal(a, b), f"{which}: a skipped update changed the state"
        grads()
        apply()
        assert float(eng.metrics[_lib.METRIC_INDEX[flag]]) == 0.0
        after = state()
        assert not torch.equal(before[0], after[0]) and float(after[3][0]) == float(before[3][0]) + 1.0
        assert all(bool(torch.isfinite(t).all()) for t in after)
    eng.close()
