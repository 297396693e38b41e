"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares, the product's layouts agree with the reference's manifest, the drop-in modules carry
the reference's state_dict keys and fail loudly without a GPU, the CLI keeps the reference's
flags, and the data-parallel host logic averages gradient buckets correctly (gloo, world 2)."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from common import GOLDEN, I, O, ROOT, SEED

import signature_gan_amd  # noqa: F401
from signature_gan_amd import _lib, layout


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "siggan.h")).read() + open(os.path.join(ROOT, "include", "siggan_mlp.h")).read()
    declared = set(re.findall(r"\b((?:siggan|mlpgan)_[a-z0-9_]+)\s*\(", header))
    declared -= {"siggan_ctx", "siggan_config", "siggan_storage", "siggan_hyper", "mlpgan_ctx", "mlpgan_config", "mlpgan_storage"}
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert declared == set(_lib.EXPORTS), (declared ^ set(_lib.EXPORTS))
    assert lib.siggan_abi_version() == _lib.ABI_VERSION


def test_create_rejects_bad_geometry_without_touching_a_gpu():
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    cfg = _lib.Config(0, 100, 32, 1, 8, 0.25, 0.2, 0)           # image_size 32: ValueError in the reference
    assert lib.siggan_create(C.byref(cfg), C.byref(h)) == -1
    assert b"64 or 128" in lib.siggan_last_error()
    with pytest.raises(ValueError):
        _lib.check(-1)
    cfg = _lib.Config(0, 100, 64, 3, 8, 0.25, 0.2, 0)
    assert lib.siggan_create(C.byref(cfg), C.byref(h)) == -1


@pytest.mark.parametrize("size,latent", [(64, 100), (128, 128)])
def test_layout_matches_reference_manifest_and_oracle(size, latent):
    man = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))[f"s{size}"]
    for entries, ref, specs in ((layout.generator_entries(latent, size), man["layout_A"]["generator_state_dict"],
                                 O.g_state_specs(latent, size)),
                                (layout.discriminator_entries(size), man["layout_A"]["discriminator_state_dict"],
                                 O.d_state_specs(size))):
        assert [k for k, _, _ in entries] == list(ref) == list(specs)
        for k, shape, kind in entries:
            assert list(shape) == ref[k]["tensor"]
            assert ref[k]["dtype"] == ("int64" if kind == "bn_count" else "float32")
    assert layout.spans(layout.generator_entries(latent, size))[1] == man["g_params"]
    assert layout.spans(layout.discriminator_entries(size))[1] == man["d_params"]


def test_dropin_modules_on_cpu():
    from signature_gan_amd.discriminator_vanilla_gan import Discriminator
    from signature_gan_amd.generator_vanilla_gan import Generator
    man = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))["s64"]
    g, d = Generator(latent_dim=100, output_size=64), Discriminator(input_size=64)
    assert list(g.state_dict()) == list(man["layout_A"]["generator_state_dict"])
    assert list(d.state_dict()) == list(man["layout_A"]["discriminator_state_dict"])
    assert g.get_num_params() == man["g_params"] and d.get_num_params() == man["d_params"]
    assert g.get_output_shape() == (1, 64, 64) and d.get_input_shape() == (1, 64, 64)
    assert g.init_size == 4 and g.init_channels == 256
    w = g.state_dict()["upsample_blocks.0.block.0.weight"]
    assert abs(float(w.std()) - 0.02) < 2e-3 and abs(float(g.state_dict()["fc.1.weight"].mean()) - 1.0) < 2e-3
    with pytest.raises(RuntimeError, match="no CPU path"):
        g(torch.randn(2, 100))
    with pytest.raises(RuntimeError, match="no CPU path"):
        d(torch.randn(2, 1, 64, 64))
    for bad in (32, 256):
        with pytest.raises(ValueError):
            Generator(output_size=bad)
        with pytest.raises(ValueError):
            Discriminator(input_size=bad)
    assert g.generate_latent(3, torch.device("cpu")).shape == (3, 100)


def test_cli_keeps_the_reference_flags():
    from signature_gan_amd.train_vanilla_gan_signatures import TrainingConfig, parse_arguments
    ref_flags = ["data_dir", "checkpoint_dir", "sample_dir", "log_dir", "run_dir", "stop_file", "epochs", "batch_size",
                 "latent_dim", "image_size", "g_lr", "d_lr", "beta1", "label_smoothing", "gradient_clip", "n_critic",
                 "sample_interval", "checkpoint_interval", "resume", "resume_from", "device", "num_workers"]
    ns = parse_arguments([])
    assert sorted(vars(ns)) == sorted(ref_flags)
    assert (ns.epochs, ns.batch_size, ns.latent_dim, ns.image_size, ns.g_lr, ns.beta1, ns.label_smoothing,
            ns.gradient_clip, ns.n_critic) == (200, 64, 100, 64, 2e-4, 0.5, 0.9, None, 1)
    with pytest.raises(SystemExit):
        parse_arguments(["--image_size", "32"])
    cfg = TrainingConfig()
    assert cfg.beta2 == 0.999 and cfg.fixed_noise_samples == 64 and cfg.gradient_clip_value is None
    man = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))["s64"]["layout_A"]
    assert {"epoch", "global_step", "generator_state_dict", "discriminator_state_dict", "g_optimizer_state_dict",
            "d_optimizer_state_dict", "config", "fixed_noise", "best_g_loss"} == set(man)


def test_mode_collapse_detector_and_logger(tmp_path):
    from signature_gan_amd.train_vanilla_gan_signatures import ModeCollapseDetector, RunLogger
    det = ModeCollapseDetector(threshold=0.1, window_size=5)
    for _ in range(4):
        det.update(0.1, 0.5)
    assert det.check_collapse() == (False, "Insufficient data")
    det.update(0.1, 0.5)
    assert det.check_collapse()[0]
    log = RunLogger(str(tmp_path), "vanilla_gan_signatures")
    log.log_metrics(1, 0.5, 1.25, 0.75, 0.25)
    assert log.save_to_csv().name.endswith("_metrics.csv") and log.save_to_json().name.endswith("_log.json")
    payload = json.load(open(log.save_to_json()))
    assert list(payload["metrics"][0]) == ["epoch", "g_loss", "d_loss", "d_real", "d_fake", "timestamp"]


def test_shard_bounds():
    from signature_gan_amd.dp import shard_bounds
    assert [shard_bounds(512, r, 8) for r in (0, 3, 7)] == [(0, 64), (192, 256), (448, 512)]
    with pytest.raises(ValueError):
        shard_bounds(65, 0, 2)


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from common import oracle_states
    from signature_gan_amd.dp import allreduce_sum_, shard_bounds
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    size, latent, gb = 64, 100, 4
    lo, hi = shard_bounds(gb, rank, world)
    real = torch.from_numpy(I.gen_real(gb, size, SEED["real"]))[lo:hi]
    z = torch.from_numpy(I.gen_z(gb, latent, SEED["z"]))[lo:hi]
    masks = [torch.from_numpy(m)[lo:hi] for m in I.gen_masks(gb, list(O.D_CHAIN[size]) * 2, 9)]
    g_sd, d_sd, _, _ = oracle_states(size, latent, warm=False)
    _, grads, _, _, _ = O.d_grads(g_sd, d_sd, real, z, masks[:4], masks[4:], size)
    spans, total = layout.spans(layout.discriminator_entries(size))
    bucket = torch.zeros(total)
    for k, (o, n, shape) in spans.items():
        bucket[o:o + n] = grads[k].reshape(-1)
    allreduce_sum_(bucket)
    bucket.mul_(1.0 / world)
    if rank == 0:
        torch.save(bucket, out)
    dist.destroy_process_group()


def test_data_parallel_bucket_average_gloo(tmp_path):
    """World-size-2 gloo run of the DP host logic against the single-process emulation of
    SURVEY 8(e): per-shard backward from the same weights, gradients averaged."""
    import torch.multiprocessing as mp
    from common import oracle_states
    out = str(tmp_path / "bucket.pt")
    port = 29650 + os.getpid() % 200
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    size, latent, gb = 64, 100, 4
    real = torch.from_numpy(I.gen_real(gb, size, SEED["real"]))
    z = torch.from_numpy(I.gen_z(gb, latent, SEED["z"]))
    masks = [torch.from_numpy(m) for m in I.gen_masks(gb, list(O.D_CHAIN[size]) * 2, 9)]
    per_rank = []
    for r in range(2):
        g_sd, d_sd, _, _ = oracle_states(size, latent, warm=False)
        sl = slice(2 * r, 2 * r + 2)
        _, grads, _, _, _ = O.d_grads(g_sd, d_sd, real[sl], z[sl], [m[sl] for m in masks[:4]], [m[sl] for m in masks[4:]], size)
        per_rank.append(grads)
    avg = O.average_grads(per_rank)
    spans, _ = layout.spans(layout.discriminator_entries(size))
    for k, (o, n, shape) in spans.items():
        err = float((got[o:o + n].view(shape) - avg[k]).abs().max())
        assert err <= 1e-5 * float(avg[k].abs().max()) + 1e-9, (k, err)


def test_spectral_norm_discriminator_keys_match_the_reference():
    """use_spectral_norm=True: the module carries torch.nn.utils.spectral_norm's keys and shapes (fixture from the
    reference: tests/golden/golden_spectral_norm.npz)."""
    from signature_gan_amd.discriminator_vanilla_gan import Discriminator
    from signature_gan_amd.vanilla_gan_model import VanillaGAN
    f = np.load(os.path.join(GOLDEN, "golden_spectral_norm.npz"))
    for size in (64, 128):
        ref = {k: tuple(s) for k, s in json.loads(str(f[f"s{size}/keys"]))}
        d = Discriminator(input_size=size, use_spectral_norm=True)
        mine = {k: tuple(v.shape) for k, v in d.state_dict().items()}
        assert mine == ref
        assert sum(p.numel() for p in d.parameters()) == sum(p.numel() for p in Discriminator(input_size=size).parameters())
    with pytest.raises(RuntimeError):                 # spectral-norm training is built; like every model it needs the MI355X
        VanillaGAN(use_spectral_norm=True, device="cpu")


def test_mlp_oracle_matches_an_nn_module_restatement():
    """The fully-connected extension has no reference model (parity unpinned): its oracle is at least checked against an
    independent nn.Module / autograd / torch.optim.Adam spelling of the same definition."""
    import torch.nn as nn
    from oracle import mlp_oracle as M
    from oracle.siggan_oracle import AdamState
    size, hidden, B = 28, (64, 96), 6
    gen = torch.Generator().manual_seed(0)
    G = nn.Sequential()
    k = 100
    g_sd = {}
    for i, h in enumerate(hidden):
        lin, bn = nn.Linear(k, h), nn.BatchNorm1d(h)
        G.append(lin); G.append(bn); G.append(nn.ReLU())
        g_sd.update({f"net.{i}.linear.weight": lin.weight, f"net.{i}.linear.bias": lin.bias, f"net.{i}.bn.weight": bn.weight,
                     f"net.{i}.bn.bias": bn.bias, f"net.{i}.bn.running_mean": bn.running_mean, f"net.{i}.bn.running_var": bn.running_var,
                     f"net.{i}.bn.num_batches_tracked": bn.num_batches_tracked})
        k = h
    out = nn.Linear(k, size * size); G.append(out); G.append(nn.Tanh())
    g_sd.update({"out.weight": out.weight, "out.bias": out.bias})
    D = nn.Sequential(nn.Flatten())
    d_sd, k = {}, size * size
    for j, h in enumerate(reversed(hidden)):
        lin = nn.Linear(k, h); D.append(lin); D.append(nn.LeakyReLU(0.2))
        d_sd.update({f"net.{j}.weight": lin.weight, f"net.{j}.bias": lin.bias}); k = h
    lin = nn.Linear(k, 1); D.append(lin); D.append(nn.Sigmoid()); d_sd.update({"out.weight": lin.weight, "out.bias": lin.bias})
    o_g = {k_: v.detach().clone() for k_, v in g_sd.items()}
    o_d = {k_: v.detach().clone() for k_, v in d_sd.items()}
    g_names = [k_ for k_ in o_g if "running" not in k_ and "num_batches" not in k_]
    g_opt, d_opt = AdamState(g_names, o_g), AdamState(list(o_d), o_d)
    z, z2 = torch.randn(B, 100, generator=gen), torch.randn(B, 100, generator=gen)
    real = torch.rand(B, 1, size, size, generator=gen) * 2 - 1
    optD, optG, crit = torch.optim.Adam(D.parameters(), 2e-4, betas=(0.5, 0.999)), torch.optim.Adam(G.parameters(), 2e-4, betas=(0.5, 0.999)), nn.BCELoss()
    D.train(); G.eval(); optD.zero_grad()
    with torch.no_grad():
        fake = G(z).view(-1, 1, size, size)
    loss = crit(D(real), torch.full((B, 1), 0.9)) + crit(D(fake), torch.zeros(B, 1)); loss.backward(); optD.step()
    met, _ = M.d_step(o_g, o_d, d_opt, real, z, hidden, size)
    assert abs(met["d_loss"] - float(loss)) <= 1e-5
    for k_, v in d_sd.items():
        assert float((v.detach() - o_d[k_]).abs().max()) <= 1e-6, k_
    G.train(); optG.zero_grad()
    gl = crit(D(G(z2).view(-1, 1, size, size)), torch.ones(B, 1)); gl.backward(); optG.step()
    met, _ = M.g_step(o_g, o_d, g_opt, z2, hidden, size)
    assert abs(met["g_loss"] - float(gl)) <= 1e-5
    for k_, v in g_sd.items():
        # (a Linear bias in front of BatchNorm has a zero true gradient: Adam moves it by up to lr on rounding noise)
        tol = 2.5 * 2e-4 if k_.endswith("linear.bias") else 1e-5
        assert float((v.detach().float() - o_g[k_].float()).abs().max()) <= tol, k_


def test_bench_gpus_n_launches_n_ranks_without_touching_the_gpu():
    """`python bench.py --gpus N` (the driver's command: no torchrun around it) must start N ranks itself, and the launching
    process must never initialise the GPU -- here: never even import torch.  Run in a fresh interpreter with torch made
    un-importable and subprocess.Popen replaced by a recorder; the real multi-rank run is rehearsed on the GPU box
    (SIGGAN_DIST_BACKEND=gloo, see profiles/)."""
    import subprocess
    script = r'''
import json, os, sys
sys.modules["torch"] = None                     # any `import torch` in the launcher raises ImportError
os.environ.pop("WORLD_SIZE", None); os.environ.pop("RANK", None)
sys.path.insert(0, %r)
import bench
seen = []
class FakeProc:
    def __init__(self, cmd, env=None, stdout=None):
        seen.append({"cmd": cmd, "rank": env["RANK"], "local": env["LOCAL_RANK"], "world": env["WORLD_SIZE"],
                     "addr": env["MASTER_ADDR"], "port": env["MASTER_PORT"], "own_stdout": stdout is None})
        self.rc = 3 if os.environ.get("FAIL_RANK") == env["RANK"] else 0
        self.terminated = False
    def poll(self): return self.rc
    def terminate(self): self.terminated = True
import subprocess
subprocess.Popen = FakeProc
rc = bench.main(["--gpus", "4", "--steps", "7", "--warmup", "2"])
print(json.dumps({"rc": rc, "seen": seen, "torch_imported": sys.modules.get("torch") is not None}))
''' % ROOT
    for fail in (None, "2"):
        env = dict(os.environ, **({"FAIL_RANK": fail} if fail else {}))
        env.pop("WORLD_SIZE", None)
        out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0, out.stderr
        rec = json.loads(out.stdout.strip().splitlines()[-1])
        assert not rec["torch_imported"]
        assert rec["rc"] == (3 if fail else 0)
        assert [r["rank"] for r in rec["seen"]] == ["0", "1", "2", "3"] == [r["local"] for r in rec["seen"]]
        assert {r["world"] for r in rec["seen"]} == {"4"} and {r["addr"] for r in rec["seen"]} == {"127.0.0.1"}
        assert len({r["port"] for r in rec["seen"]}) == 1
        assert [r["own_stdout"] for r in rec["seen"]] == [True, False, False, False]     # ONE JSON line: rank 0's
        for r in rec["seen"]:
            assert r["cmd"][1].endswith("bench.py") and r["cmd"][2:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_bench_states_the_host_core_count():
    """bench.py's cpu_baseline carries the box's physical core count, logical CPUs and this job's share (north star: "core count
    stated"); the probe must give sane numbers wherever it runs and never touch torch or the GPU."""
    import bench
    phys, logical, share = bench.host_cores()
    assert logical >= 1 and 1 <= share <= logical
    assert phys is None or 1 <= phys <= logical
