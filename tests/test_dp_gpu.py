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
