#!/usr/bin/env python3
"""Headline benchmark: signature images/sec for one G+D train step (n_critic = 1).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): reference conv G/D, 64x64x1, z = 100, batch 64 PER GPU, fp32,
dropout 0.25 active in the D step, label smoothing 0.9, Adam lr 2e-4 betas (0.5, 0.999), clipping
off; the real batch is synthetic U[-1,1] resident in HBM, z and the dropout masks are drawn by the
library's RNG inside the step, weights are random-init from the reference's init distribution.
A "step" = siggan_d_grads/apply + siggan_g_grads/apply through the C ABI (with N > 1 the two flat
gradient buckets are all-reduced over RCCL between grads and apply: weak scaling, per-replica BN).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant MFMA kernel: algorithmic FLOPs per launch / its mean launch duration
                  measured with HIP events on the launch stream over a second, instrumented pass of
                  the same K steps, run one kernel at a time (the timed pass carries no instrumentation
                  and overlaps independent kernels on side streams; `--serialize` turns that off for the
                  whole run, which is how the rocprofv3 summaries under profiles/ are taken)
  cpu_baseline -- the oracle's same step (torch CPU, fp32) timed on the host cores (rank 0, N = 1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SIZE, LATENT, BATCH = 64, 100, 64
FLOP_PER_IMAGE = 1975.8e6          # 8*D_fwd + 4*G_fwd, SURVEY 8(d) (conv/convT/linear MACs x 2)
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz


def cpu_baseline(threads, budget_s=12.0):
    """The oracle (kind "port") on the host cores: same step, same shapes, fp32."""
    import torch
    for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from common import I, O, oracle_states
    torch.set_num_threads(threads)
    g_sd, d_sd, g_opt, d_opt = oracle_states(SIZE, LATENT, warm=False)
    real = torch.from_numpy(I.gen_real(BATCH, SIZE, 22))
    chans = list(O.D_CHAIN[SIZE])
    gen = torch.Generator().manual_seed(3)

    def one():
        z1, z2 = torch.randn(BATCH, LATENT, generator=gen), torch.randn(BATCH, LATENT, generator=gen)
        masks = [(torch.rand(BATCH, c, generator=gen) < 0.75).float() for c in chans * 2]
        O.d_step(g_sd, d_sd, d_opt, real, z1, masks[:len(chans)], masks[len(chans):], SIZE)
        O.g_step(g_sd, d_sd, g_opt, z2, SIZE)

    one(); one()
    n, t0 = 0, time.perf_counter()
    while True:
        one(); n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 64:
            break
    return BATCH * n / dt, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true", help="skip the instrumented pass")
    ap.add_argument("--serialize", action="store_true",
                    help="run every pass with side-lane overlap OFF (one kernel at a time): the mode the roofline "
                         "pass always uses, and the one to profile with rocprofv3 so per-kernel durations agree")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import signature_gan_amd  # noqa: F401
    from signature_gan_amd.dp import DataParallelStep, env_rank
    from signature_gan_amd.engine import Engine

    rank, world, local = env_rank()
    if args.gpus > 1 or world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # SIGGAN_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (ranks then
        # share devices round-robin; RCCL itself refuses two ranks on one GPU).  The driver's runs use nccl = RCCL.
        backend = os.environ.get("SIGGAN_DIST_BACKEND", "nccl")
        local = local % torch.cuda.device_count() if backend != "nccl" else local
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
        local = 0
    dev = torch.device("cuda", local)

    eng = Engine(latent_dim=LATENT, image_size=SIZE, max_batch=BATCH, device=str(dev), seed=2 + rank)
    eng.init_reference(seed=0)                                  # identical initial weights on every rank
    if args.serialize:
        eng.set_mode(graph=False, overlap=False)
    dp = DataParallelStep(eng)
    dp.sync_initial_state()
    gen = torch.Generator(device="cpu").manual_seed(1 + rank)
    real = (torch.rand(BATCH, 1, SIZE, SIZE, generator=gen) * 2 - 1).to(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The loop hands each step the batch of the step after it as well (a prefetching loader has it):
    # that batch's D(real) forward then runs beside this step's Generator backward (siggan_stage_real).
    # Every step still does one D(real) forward -- for its successor instead of for itself.
    for _ in range(args.warmup):
        dp.step(real, next_real=real)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        dp.step(real, next_real=real)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    m = eng.metrics.cpu()
    assert torch.isfinite(m).all(), "non-finite training metrics"

    roofline = None
    if rank == 0 and not args.no_roofline:
        # one kernel at a time: a launch's event-bracketed time is then the kernel's own duration
        # (with overlap on, two MFMA kernels share the chip and each looks proportionally longer)
        eng.set_mode(graph=False, overlap=False)
        eng.prof_enable(True)
        for _ in range(args.steps):
            dp.step(real, next_real=real) if world == 1 else (eng.d_step(real, sync=False), eng.g_step(BATCH, sync=False))
        recs = eng.prof_read()
        eng.prof_enable(False)
        eng.set_mode(graph=False, overlap=not args.serialize)
        top = max(recs, key=lambda r: r["ms"])
        ach = top["flops"] / (top["ms"] * 1e-3) / 1e12
        traffic = None                  # HBM bytes per launch from the committed PMC passes (separate rocprofv3 runs)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                traffic = json.load(f)["per_launch_bytes"][top["name"]]["total"]
        except (OSError, KeyError, ValueError):
            pass
        fam_ms, fam_fl = sum(r["ms"] for r in recs), sum(r["flops"] for r in recs)
        roofline = {
            "bound": "mfma", "kernel": top["name"], "achieved": round(ach, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": traffic,
            "launches_per_step": top["launches"] / args.steps,
            "avg_launch_us": round(1e3 * top["ms"] / top["launches"], 2),
            "gflop_per_launch": round(top["flops"] / top["launches"] / 1e9, 4),
            "mfma_family": {"ms_per_step": round(fam_ms / args.steps, 4),
                            "achieved": round(fam_fl / (fam_ms * 1e-3) / 1e12, 3),
                            "kernels": {r["name"]: {"launches_per_step": r["launches"] / args.steps,
                                                    "ms_per_step": round(r["ms"] / args.steps, 4),
                                                    "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2)} for r in recs}},
        }
    if world > 1:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        # two thread counts (this job's CPU share on the GPU box is 16; torch's oneDNN convs do not always scale past
        # 8): the faster one is the baseline, the other is quoted in `sample`
        runs = []
        for threads in sorted({min(16, os.cpu_count() or 1), min(8, os.cpu_count() or 1)}, reverse=True):
            v, n = cpu_baseline(threads, budget_s=10.0)
            runs.append((v, threads, n))
        (v, threads, n), rest = max(runs), [r for r in runs if r != max(runs)]
        other = "; ".join(f"{round(r[0], 1)} images/s with {r[1]} threads ({r[2]} steps)" for r in rest)
        cpu = {"value": round(v, 1), "unit": "images/s", "cores": threads, "kind": "port",
               "sample": f"{n} G+D steps of the same workload (batch {BATCH}, 64x64, fp32) by oracle/siggan_oracle.py "
                         f"on torch CPU with {threads} threads" + (f"; {other}" if other else "")}

    if rank == 0:
        imgs = BATCH * world * args.steps
        out = {
            "metric": "signature images/sec (G+D train step, bs64 64x64 z=100)",
            "value": round(imgs / dt, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: reference conv G/D 64x64x1, z=100, batch 64 per GPU, fp32, n_critic=1",
                       "global_batch": BATCH * world, "parallelism": f"dp{world}"},
            "achieved_tflops_whole_step": round(imgs / dt * FLOP_PER_IMAGE / 1e12, 3),
            "frac_of_fp32_mfma_peak_whole_step": round(imgs / dt * FLOP_PER_IMAGE / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
            "final_metrics": {"d_loss": round(float(m[0]), 4), "g_loss": round(float(m[8]), 4)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
