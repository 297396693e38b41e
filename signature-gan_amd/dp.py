"""Data-parallel host logic: one process per GPU, per-replica BatchNorm, gradient buckets averaged
with one all-reduce per network per step (RCCL over xGMI through torch.distributed's "nccl"
backend on the GPU box; "gloo" in the CPU tests).  The reference has no distributed path; this
is the definition SURVEY 8(e) fixes: each rank runs the reference's step on its contiguous shard
of the global batch, gradients are averaged, clipping acts on the averaged gradient."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_bounds(global_batch, rank, world):
    """Contiguous split: rank r owns samples [r*B/W, (r+1)*B/W); the global batch must divide
    evenly (the reference's loader drops ragged batches, data_loader_signatures.py:313)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def allreduce_sum_(flat, group=None):
    """In-place SUM all-reduce of one flat gradient bucket (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def broadcast_state_(tensors, src=0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)


class DataParallelStep:
    """G+D train step of the reference sharded over the ranks of the default process group."""

    def __init__(self, engine, lr_g=2e-4, lr_d=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9, clip=None):
        self.e = engine
        self.hp = dict(lr_g=lr_g, lr_d=lr_d, beta1=beta1, beta2=beta2, ls=label_smoothing, clip=clip)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0

    def sync_initial_state(self):
        e = self.e
        broadcast_state_([e.g_params, e.d_params, e.g_bn_mean, e.g_bn_var, e.g_bn_batches, e.g_exp_avg, e.g_exp_avg_sq,
                          e.d_exp_avg, e.d_exp_avg_sq, e.g_adam_steps, e.d_adam_steps])
        e.params_changed()

    def step(self, real_local, z_d=None, z_g=None, masks=None, sync=False, next_real=None):
        """next_real: this rank's shard of the FOLLOWING step's real batch, when the loop already holds it
        (a prefetching loader does): its D(real) forward then runs beside this step's Generator backward."""
        e, hp, inv = self.e, self.hp, 1.0 / self.world
        e.step_begin(real_local, z_d, masks, z_g, hp["ls"])       # D grads + the G step's forward beside them
        allreduce_sum_(e.d_grads)
        dm = e.d_apply(hp["lr_d"], hp["beta1"], hp["beta2"], clip=hp["clip"], grad_scale=inv, sync=sync)
        if next_real is not None:
            e.stage_real(next_real)
        e.g_compute_grads(real_local.shape[0])
        allreduce_sum_(e.g_grads)
        gm = e.g_apply(hp["lr_g"], hp["beta1"], hp["beta2"], clip=hp["clip"], grad_scale=inv, sync=sync)
        if sync:
            dm.update(gm)
            return dm
        return None
