"""Data-parallel step on the device: two ranks (gloo, both on cuda:0 -- RCCL refuses two ranks on one GPU, the
driver's multi-GPU runs use nccl) run DataParallelStep.step on their shards; rank 0's parameters, Adam moments and
BatchNorm buffers must equal, bit for bit, a single-process emulation with the same kernels: per-shard gradients
from the same weights, summed, applied with grad_scale = 1/W.  Parity of the kernels themselves against
the oracle is tests/test_engine_gpu.py's job."""
import os

import pytest
import torch

from common import I, O, SEED

pytestmark = pytest.mark.gpu
SIZE, LATENT, GB, WORLD = 64, 100, 8, 2


def _inputs():
    real = torch.from_numpy(I.gen_real(GB, SIZE, SEED["real"]))
    z_d = torch.from_numpy(I.gen_z(GB, LATENT, 71))
    z_g = torch.from_numpy(I.gen_z(GB, LATENT, 72))
    masks = [torch.from_numpy(m) for m in I.gen_masks(GB, list(O.D_CHAIN[SIZE]) * 2, 9)]
    return real, z_d, z_g, masks


def _state(eng):
    names = ("g_params", "d_params", "g_exp_avg", "g_exp_avg_sq", "d_exp_avg", "d_exp_avg_sq", "g_bn_mean", "g_bn_var",
             "g_adam_steps", "d_adam_steps")
    return {n: getattr(eng, n).detach().cpu().clone() for n in names}


def _worker(rank, port, out):
    import torch.distributed as dist
    from hipcommon import cuda, make_engine
    from signature_gan_amd.dp import DataParallelStep, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    torch.cuda.set_device(0)
    lo, hi = shard_bounds(GB, rank, WORLD)
    real, z_d, z_g, masks = _inputs()
    eng = make_engine(SIZE, LATENT, hi - lo, warm=True)
    dp = DataParallelStep(eng, clip=0.5)
    dp.sync_initial_state()
    for _ in range(2):
        m = dp.step(cuda(real[lo:hi]), cuda(z_d[lo:hi]), cuda(z_g[lo:hi]), [x[lo:hi] for x in masks], sync=True)
    if rank == 0:
        torch.save({"state": _state(eng), "metrics": m}, out)
    eng.close()
    dist.destroy_process_group()


def test_two_rank_step_equals_single_process_emulation(tmp_path):
    import torch.multiprocessing as mp
    from hipcommon import cuda, make_engine
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(29700 + os.getpid() % 200, out), nprocs=WORLD, join=True)
    got = torch.load(out, weights_only=True)

    real, z_d, z_g, masks = _inputs()
    h = GB // WORLD
    e0, e1 = make_engine(SIZE, LATENT, h, warm=True), make_engine(SIZE, LATENT, h, warm=True)
    sh = lambda t, r: cuda(t[r * h:(r + 1) * h])
    msk = lambda r: [x[r * h:(r + 1) * h] for x in masks]
    for _ in range(2):
        e1.d_compute_grads(sh(real, 1), sh(z_d, 1), msk(1))
        e0.d_compute_grads(sh(real, 0), sh(z_d, 0), msk(0))
        e0.d_grads.add_(e1.d_grads)
        e0.d_apply(clip=0.5, grad_scale=1.0 / WORLD)
        for n in ("d_params", "d_exp_avg", "d_exp_avg_sq", "d_adam_steps"):       # rank 1 holds the same D after the all-reduce
            getattr(e1, n).copy_(getattr(e0, n))
        e1.params_changed()
        e1.g_compute_grads(h, sh(z_g, 1))
        e0.g_compute_grads(h, sh(z_g, 0))
        e0.g_grads.add_(e1.g_grads)
        e0.g_apply(clip=0.5, grad_scale=1.0 / WORLD)
        for n in ("g_params", "g_exp_avg", "g_exp_avg_sq", "g_adam_steps"):
            getattr(e1, n).copy_(getattr(e0, n))
        e1.params_changed()
    want = _state(e0)
    for k, v in want.items():
        assert torch.equal(got["state"][k], v), f"{k}: the two-rank run differs from the emulation"
    assert got["metrics"]["d_loss"] > 0 and got["metrics"]["g_loss"] > 0
    e0.close(); e1.close()


def test_library_rccl_communicator_world_1_is_bit_identical():
    """The collective inside the library (siggan_comm_init -> ncclCommInitRank, ncclAllReduce of the gradient bucket in
    siggan_d_apply / siggan_g_apply, ncclBroadcast of the initial state) on ONE rank: a real RCCL communicator on cuda:0
    whose all-reduce must leave every result bit-identical to the run without a communicator."""
    from hipcommon import cuda, make_engine
    from signature_gan_amd.dp import DataParallelStep
    real, z_d, z_g, masks = _inputs()

    def run(comm):
        eng = make_engine(SIZE, LATENT, GB, warm=True)
        if comm:
            eng.comm_init(0, 1, eng.comm_unique_id())
            assert eng.comm_world == 1
        dp = DataParallelStep(eng, clip=0.5, transport="lib" if comm else "host")
        dp.sync_initial_state()
        mets = [dp.step(cuda(real), cuda(z_d), cuda(z_g), masks, sync=True) for _ in range(3)]
        st = _state(eng)
        if comm:
            eng.comm_destroy()
        eng.close()
        return mets, st

    m0, s0 = run(False)
    m1, s1 = run(True)
    assert m0 == m1
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


# ---- the data-parallel step against the ORACLE's emulation ("emulate W replicas on the oracle by running the D/G
# backward on each shard from the same weights with that shard's masks / noise, averaging gradients, and applying one Adam
# step; the GPU run must match that") ---------------------------------------------------------------------------------------
GB_O = 16


def _inputs_o():
    real = torch.from_numpy(I.gen_real(GB_O, SIZE, SEED["real"]))
    z_d = torch.from_numpy(I.gen_z(GB_O, LATENT, 81))
    z_g = torch.from_numpy(I.gen_z(GB_O, LATENT, 82))
    masks = [torch.from_numpy(m) for m in I.gen_masks(GB_O, list(O.D_CHAIN[SIZE]) * 2, 19)]
    return real, z_d, z_g, masks


def _oracle_worker(rank, world, port, outdir):
    import torch.distributed as dist
    from hipcommon import cuda, hip_signs_d, hip_signs_g, make_engine
    from signature_gan_amd.dp import allreduce_sum_, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    lo, hi = shard_bounds(GB_O, rank, world)
    b = hi - lo
    real, z_d, z_g, masks = _inputs_o()
    e = make_engine(SIZE, LATENT, b, warm=True)
    cp = lambda d: {k: t.detach().float().cpu().clone() for k, t in d.items()}
    # DataParallelStep.step (transport "host"), opened up between its halves to read this replica's sign decisions
    e.step_begin(cuda(real[lo:hi]), cuda(z_d[lo:hi]), [x[lo:hi] for x in masks], cuda(z_g[lo:hi]), 0.9)
    rec = {"signs_d": hip_signs_d(e, SIZE, b, 2)}
    allreduce_sum_(e.d_grads)
    rec["d_metrics"] = e.d_apply(grad_scale=1.0 / world, sync=True)
    rec["d_grads_avg"] = cp(e.views("d", "grads"))           # (written back scaled: the averaged gradient, as torch leaves .grad)
    e.g_compute_grads(b)
    rec["signs_g"] = hip_signs_g(e, SIZE, b) + hip_signs_d(e, SIZE, b, 1)
    allreduce_sum_(e.g_grads)
    rec["g_metrics"] = e.g_apply(grad_scale=1.0 / world, sync=True)
    rec["g_grads_avg"] = cp(e.views("g", "grads"))
    rec["d_w"], rec["g_w"], rec["bn"] = cp(e.views("d")), cp(e.views("g")), cp(e.bn_views())
    torch.save(rec, os.path.join(outdir, f"rank{rank}.pt"))
    e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_step_equals_the_oracles_emulation(world, tmp_path):
    import torch.multiprocessing as mp
    from common import oracle_states
    from hipcommon import count_sign_flips
    from test_engine_gpu import _adam_envelope
    mp.spawn(_oracle_worker, args=(world, 29900 + os.getpid() % 90 + world, str(tmp_path)), nprocs=world, join=True)
    ranks = [torch.load(str(tmp_path / f"rank{r}.pt"), weights_only=True) for r in range(world)]
    real, z_d, z_g, masks = _inputs_o()
    h = GB_O // world
    nb = len(masks) // 2
    g_sd, d_sd, g_opt, d_opt = oracle_states(SIZE, LATENT, warm=True)
    w0 = {k: v.clone() for k, v in d_sd.items()}
    m0, v0 = {k: v.clone() for k, v in d_opt.m.items()}, {k: v.clone() for k, v in d_opt.v.items()}

    def close(got, want, what, noise=()):
        gscale = max(float(want[k].abs().max()) for k in want)
        for k in want:
            sc = max(float(want[k].abs().max()), (1e-2 if k in noise else 1e-3) * gscale)
            err = float((got[k] - want[k]).abs().max()) / sc
            assert err <= 1e-4, f"{what} {k}: {err:.3e} of scale {sc:.3e}"
        return {k: max(float(want[k].abs().max()), (1e-2 if k in noise else 1e-3) * gscale) for k in want}

    # ---- D half: per-shard oracle gradients from the SAME weights, that shard's masks / noise and sign decisions ----
    per = []
    for r in range(world):
        sl = slice(r * h, (r + 1) * h)
        ms = [x[sl] for x in masks]
        rec = []
        met, grads, _, _, _ = O.d_grads(g_sd, d_sd, real[sl], z_d[sl], ms[:nb], ms[nb:], SIZE, signs=ranks[r]["signs_d"], record=rec)
        count_sign_flips(ranks[r]["signs_d"], rec, keep=ms)
        for k in ("d_loss", "d_real_mean", "d_fake_mean"):
            assert abs(ranks[r]["d_metrics"][k] - met[k]) <= 2e-4 * abs(met[k]) + 2e-6, (r, k)
        per.append(grads)
    avg = O.average_grads(per)
    for r in range(world):                       # every rank holds the same averaged bucket
        scale = close(ranks[r]["d_grads_avg"], avg, f"world {world} rank {r} averaged D gradient")
    d_opt.apply(d_sd, avg, 2e-4, 0.5, 0.999)
    for k in avg:
        lo, hi = _adam_envelope(w0[k], m0[k], v0[k], avg[k], 1e-4 * scale[k], d_opt.step)
        assert bool(((ranks[0]["d_w"][k] >= lo) & (ranks[0]["d_w"][k] <= hi)).all()), f"D weight {k} outside the Adam envelope"
        assert all(torch.equal(ranks[r]["d_w"][k], ranks[0]["d_w"][k]) for r in range(1, world)), f"replicas diverged: {k}"
    # ---- G half: through the updated Discriminator; per-replica BatchNorm (each rank = the reference's step on its shard) ----
    gw0 = {k: g_sd[k].clone() for k in g_opt.names}
    gm0, gv0 = {k: v.clone() for k, v in g_opt.m.items()}, {k: v.clone() for k, v in g_opt.v.items()}
    per, bufs0 = [], None
    for r in range(world):
        sl = slice(r * h, (r + 1) * h)
        gr = {k: v.clone() for k, v in g_sd.items()}
        rec = []
        met, grads, _, _ = O.g_grads(gr, d_sd, z_g[sl], SIZE, signs=ranks[r]["signs_g"], record=rec)
        count_sign_flips(ranks[r]["signs_g"], rec)
        for k in ("g_loss", "g_fake_mean"):
            assert abs(ranks[r]["g_metrics"][k] - met[k]) <= 2e-4 * abs(met[k]) + 2e-6, (r, k)
        for k, t in ranks[r]["bn"].items():      # this replica's own BatchNorm buffers
            assert float((t - gr[k].float()).abs().max()) <= 2e-4 * float(gr[k].float().abs().max()) + 1e-7, (r, k)
        per.append(grads)
    avg = O.average_grads(per)
    for r in range(world):
        scale = close(ranks[r]["g_grads_avg"], avg, f"world {world} rank {r} averaged G gradient", noise=("fc.0.bias",))
    g_opt.apply(g_sd, avg, 2e-4, 0.5, 0.999)
    for k in avg:
        lo, hi = _adam_envelope(gw0[k], gm0[k], gv0[k], avg[k], 1e-4 * scale[k], g_opt.step)
        assert bool(((ranks[0]["g_w"][k] >= lo) & (ranks[0]["g_w"][k] <= hi)).all()), f"G weight {k} outside the Adam envelope"
        assert all(torch.equal(ranks[r]["g_w"][k], ranks[0]["g_w"][k]) for r in range(1, world)), f"replicas diverged: {k}"
