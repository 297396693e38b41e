"""Data-parallel host logic: one process per GPU, per-replica BatchNorm, gradient buckets averaged with ONE sum
all-reduce per network per step.  The reference has no distributed path; this is the definition SURVEY 8(e) fixes: each
rank runs the reference's step on its contiguous shard of the global batch, gradients are averaged, clipping acts on
the averaged gradient.

Two transports for the bucket:
  * "lib"  -- the library's own RCCL communicator (siggan_comm_init): the all-reduce is issued inside siggan_d_apply /
              siggan_g_apply on the step's stream, over xGMI; torch.distributed is used ONCE, to hand rank 0's 128-byte
              RCCL id to the other ranks (any launcher channel would do).  Default whenever the ranks own distinct GPUs.
  * "host" -- torch.distributed.all_reduce between the grads / apply halves of the C ABI; used by the gloo tests
              (CPU emulation, or several ranks sharing one GPU, which RCCL refuses)."""
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


def init_library_comm(engine, rank=None, world=None):
    """Create the library's RCCL communicator over the ranks of the default process group (or a single-rank one):
    rank 0 draws the id, torch.distributed carries its 128 bytes to the others -- the only use of torch.distributed on
    this path.  ncclCommInitRank is collective, so a rank that cannot take part must be known BEFORE any rank enters it:
    every rank first does the rank-independent part (resolve the RCCL entry points; drawing an id does that and is
    harmless to throw away), the ranks vote with one MIN all-reduce, and only a unanimous group goes on -- otherwise
    EVERY rank raises the same RuntimeError and the caller can fall back (bench.py does)."""
    grouped = dist.is_available() and dist.is_initialized()
    rank = (dist.get_rank() if grouped else 0) if rank is None else rank
    world = (dist.get_world_size() if grouped else 1) if world is None else world
    ok, uid, why = 1, None, ""
    try:
        uid = engine.comm_unique_id()
    except Exception as exc:                                     # noqa: BLE001 -- voted on below
        ok, why = 0, f"{type(exc).__name__}: {exc}"
    if grouped and world > 1:
        on_gpu = dist.get_backend() == "nccl"
        flag = torch.tensor([ok], dtype=torch.int32, device=engine.device if on_gpu else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            raise RuntimeError("the library RCCL communicator cannot be created on every rank"
                               + (f" (this rank: {why})" if why else " (another rank failed)"))
        box = [uid if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = box[0]
    elif not ok:
        raise RuntimeError(why)
    engine.comm_init(rank, world, uid)
    return engine


class DataParallelStep:
    """G+D train step of the reference sharded over the ranks of the default process group."""

    def __init__(self, engine, lr_g=2e-4, lr_d=2e-4, beta1=0.5, beta2=0.999, label_smoothing=0.9, clip=None, transport=None):
        self.e = engine
        self.hp = dict(lr_g=lr_g, lr_d=lr_d, beta1=beta1, beta2=beta2, ls=label_smoothing, clip=clip)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if transport is None:
            transport = "lib" if engine.comm_world > 1 or getattr(engine, "_comm", None) is not None else "host"
        if transport not in ("lib", "host"):
            raise ValueError("transport must be 'lib' or 'host'")
        if transport == "lib" and getattr(engine, "_comm", None) is None:
            init_library_comm(engine)
        self.transport = transport

    def sync_initial_state(self):
        e = self.e
        state = [e.g_params, e.d_params, e.g_bn_mean, e.g_bn_var, e.g_bn_batches, e.g_exp_avg, e.g_exp_avg_sq,
                 e.d_exp_avg, e.d_exp_avg_sq, e.g_adam_steps, e.d_adam_steps]
        if e.spectral_norm:        # weight_u / weight_v: each rank drew its own; sigma (hence W / sigma) must agree across replicas
            state += [e.d_sn_u, e.d_sn_v]
        if self.transport == "lib":
            for t in state:
                e.comm_broadcast(t, 0)
        else:
            broadcast_state_(state)
        e.params_changed()

    def step(self, real_local, z_d=None, z_g=None, masks=None, sync=False, next_real=None):
        """next_real: this rank's shard of the FOLLOWING step's real batch, when the loop already holds it
        (a prefetching loader does): its D(real) forward then runs beside this step's Generator backward."""
        e, hp = self.e, self.hp
        lib = self.transport == "lib"
        inv = 1.0 if lib else 1.0 / self.world          # the library divides by its communicator's world size itself
        e.step_begin(real_local, z_d, masks, z_g, hp["ls"])       # D grads + the G step's forward beside them
        if not lib:
            allreduce_sum_(e.d_grads)
        dm = e.d_apply(hp["lr_d"], hp["beta1"], hp["beta2"], clip=hp["clip"], grad_scale=inv, sync=sync)
        if next_real is not None:
            e.stage_real(next_real)
        e.g_compute_grads(real_local.shape[0])
        if not lib:
            allreduce_sum_(e.g_grads)
        gm = e.g_apply(hp["lr_g"], hp["beta1"], hp["beta2"], clip=hp["clip"], grad_scale=inv, sync=sync)
        if sync:
            dm.update(gm)
            return dm
        return None
