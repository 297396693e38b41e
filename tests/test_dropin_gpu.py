"""The drop-in modules on the MI355X: VanillaGAN / GANTrainer steps equal the engine's (and
therefore the oracle's), checkpoints keep the reference's two layouts (checked against the
manifest produced from the reference), a real torch.optim.Adam can load the optimiser state,
and the trainer's epoch loop writes the artefacts the reference's UI looks for."""
import json
import os

import numpy as np
import pytest
import torch

from common import GOLDEN, I, O, SEED, assert_close, oracle_states

pytestmark = pytest.mark.gpu


def _manifest(obj):
    if isinstance(obj, torch.Tensor):
        return {"tensor": list(obj.shape), "dtype": str(obj.dtype).replace("torch.", "")}
    if isinstance(obj, dict):
        return {str(k): _manifest(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        if len(obj) > 8 and all(isinstance(x, (int, float)) for x in obj):
            return {"list_of": type(obj[0]).__name__, "len": len(obj)}
        return [_manifest(v) for v in obj]
    return type(obj).__name__


def _load_inputs_into(model, size, latent):
    gs, ds = O.g_state_specs(latent, size), O.d_state_specs(size)
    model.generator.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in I.gen_state(gs, SEED["state_g"]).items()})
    model.discriminator.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in I.gen_state(ds, SEED["state_d"]).items()})


def test_vanilla_gan_steps_match_oracle_and_checkpoint_roundtrip(tmp_path):
    from signature_gan_amd.vanilla_gan_model import VanillaGAN
    size, latent, B = 64, 100, 8
    m = VanillaGAN(latent_dim=latent, image_size=size, device="cuda:0", max_batch=B)
    _load_inputs_into(m, size, latent)
    z = torch.from_numpy(I.gen_z(B, latent, SEED["z"]))
    g_sd, d_sd, g_opt, d_opt = oracle_states(size, latent, warm=False)

    # generation == oracle eval forward; D forward eval == oracle
    img = m.generate(B, noise=z)
    assert img.shape == (B, 1, size, size) and img.is_cuda
    assert_close(img.cpu().numpy(), O.g_forward(g_sd, z, False, size).numpy(), 2e-4, 2e-5, "generate")
    x = torch.from_numpy(I.gen_real(B, size, SEED["real"]))
    m.discriminator.eval()
    assert_close(m.discriminator(x.cuda()).cpu().numpy(), O.d_forward(d_sd, x, size).numpy(), 2e-4, 2e-6, "D forward")
    assert m.discriminator.forward_features(x.cuda()).shape == (B, 8192)

    # G step with injected noise == oracle (no dropout in the G step, so fully deterministic)
    gm = m.train_generator_step(B, noise=z)
    om, _ = O.g_step(g_sd, d_sd, g_opt, z, size)
    assert_close(gm["g_loss"], om["g_loss"], 2e-4, 2e-6, "g_loss")
    assert_close(gm["g_fake_mean"], om["g_fake_mean"], 2e-4, 2e-6, "g_fake_mean")
    for k, v in m.generator.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert_close(v.float().cpu().numpy(), g_sd[k].float().numpy(), 2e-4, 1e-6, k)
    dm = m.train_discriminator_step(x)            # library RNG for z / masks: just sanity
    assert set(dm) >= {"d_loss", "d_loss_real", "d_loss_fake", "d_real_acc", "d_fake_acc", "d_real_mean", "d_fake_mean"}
    assert all(np.isfinite(v) for v in dm.values() if v is not None) and m.global_step == 1
    step = m.train_step(x, n_critic=2)
    assert "g_loss" in step and "d_loss" in step and m.global_step == 3
    step = m.train_step(x)                        # n_critic == 1: the pipelined engine step
    assert set(step) >= {"d_loss", "d_real_acc", "d_fake_mean", "g_loss", "g_fake_mean"} and m.global_step == 4
    assert all(np.isfinite(v) for v in step.values()) and not m.discriminator.training and m.generator.training

    # ---- layout B checkpoint: structure equals the reference's, safe loader accepts it ----------
    man = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))["s64"]
    m.save(tmp_path / "ck")
    ck = torch.load(tmp_path / "ck.pt", map_location="cpu", weights_only=True)
    got, ref = _manifest(ck), man["layout_B"]
    # the reference's keys plus ONE extra primitive entry its loaders ignore: the library RNG position (seed, offset)
    assert set(got) - {"engine_rng_state"} == set(ref)
    assert ck["engine_rng_state"] == list(m.engine.rng_state()) and all(isinstance(v, int) for v in ck["engine_rng_state"])
    for key in ("generator_state_dict", "discriminator_state_dict"):
        assert got[key] == ref[key], key
    for key in ("g_optimizer_state_dict", "d_optimizer_state_dict"):
        assert got[key]["state"] == ref[key]["state"], key
        assert set(got[key]["param_groups"][0]) == set(ref[key]["param_groups"][0])
    assert sorted(json.load(open(tmp_path / "ck_config.json"))) == man["layout_B_config_json_keys"]

    # a stock torch.optim.Adam accepts the optimiser state (reference-side resume)
    params = [torch.nn.Parameter(torch.zeros_like(p)) for p in m.generator.parameters()]
    stock = torch.optim.Adam(params, lr=2e-4, betas=(0.5, 0.999))
    stock.load_state_dict(ck["g_optimizer_state_dict"])
    assert float(stock.state[params[0]]["step"]) == 3.0   # three G updates so far

    # round trip into a fresh model: identical weights, moments, next step identical
    m2 = VanillaGAN.from_checkpoint(tmp_path / "ck", device="cuda:0")
    for a, b in zip(m.generator.state_dict().values(), m2.generator.state_dict().values()):
        assert torch.equal(a, b)
    assert m2.engine.rng_state() == m.engine.rng_state()      # whatever mix of D-only / G-only / joint steps came before
    assert torch.equal(m.engine.d_exp_avg_sq, m2.engine.d_exp_avg_sq) and torch.equal(m.engine.g_adam_steps, m2.engine.g_adam_steps)
    z2 = torch.from_numpy(I.gen_z(B, latent, 77))
    a, b = m.train_generator_step(B, noise=z2), m2.train_generator_step(B, noise=z2)
    assert a["g_loss"] == b["g_loss"] and torch.equal(m.engine.g_params, m2.engine.g_params)


def test_standalone_modules_and_reference_checkpoint_layouts(tmp_path):
    """Generator alone (the generation callers' path): .to('cuda'), load a layout-A style dict."""
    from signature_gan_amd.generator_vanilla_gan import Generator
    size, latent, B = 128, 128, 3
    g = Generator(latent_dim=latent, output_size=size).to("cuda:0")
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in I.gen_state(O.g_state_specs(latent, size), SEED["state_g"]).items()}
    g.load_state_dict(sd)
    g.eval()
    z = torch.from_numpy(I.gen_z(B, latent, 5))
    g_sd = {k: v.clone() for k, v in sd.items()}
    assert_close(g(z.cuda()).cpu().numpy(), O.g_forward(g_sd, z, False, size).numpy(), 2e-4, 2e-5, "standalone G")
    u8 = O.to_uint8(g(z.cuda()).cpu())
    assert u8.dtype == torch.uint8


def test_trainer_loop_artifacts(tmp_path):
    from signature_gan_amd.train_vanilla_gan_signatures import GANTrainer, TrainingConfig
    cfg = TrainingConfig(batch_size=8, epochs=2, sample_interval=1, checkpoint_interval=1, gradient_clip_value=1.0,
                         checkpoint_dir=str(tmp_path / "checkpoints"), sample_dir=str(tmp_path / "samples"),
                         log_dir=str(tmp_path / "logs"), fixed_noise_samples=16)
    tr = GANTrainer(cfg, device="cuda:0", stop_file=str(tmp_path / "stop.request"))
    data = [torch.rand(8, 1, 64, 64) * 2 - 1 for _ in range(3)]
    d = tr._train_discriminator(data[0])
    assert set(d) == {"d_loss", "d_loss_real", "d_loss_fake", "d_real_mean", "d_fake_mean", "d_grad_norm"} and d["d_grad_norm"] > 0
    g = tr._train_generator(8)
    assert set(g) == {"g_loss", "g_fake_mean", "g_grad_norm"} and g["g_grad_norm"] > 0
    summary = tr.train(data_loader=data)
    assert summary["total_epochs"] == 2
    for name in ("epoch_0000.png", "epoch_0001.png", "epoch_0002.png"):
        assert (tmp_path / "samples" / name).exists()
    for name in ("checkpoint_epoch_0001.pt", "checkpoint_epoch_0002.pt", "checkpoint_latest.pt", "checkpoint_best.pt"):
        assert (tmp_path / "checkpoints" / name).exists()
    assert list((tmp_path / "logs").glob("*_metrics.csv")) and list((tmp_path / "logs").glob("*_log.json"))
    man = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))["s64"]["layout_A"]
    ck = torch.load(tmp_path / "checkpoints" / "checkpoint_latest.pt", map_location="cpu", weights_only=True)
    got = _manifest(ck)
    assert set(got) - {"engine_rng_state"} == set(man)
    assert got["generator_state_dict"] == man["generator_state_dict"]
    assert got["d_optimizer_state_dict"]["state"] == man["d_optimizer_state_dict"]["state"]
    # resume
    tr2 = GANTrainer(cfg, device="cuda:0")
    assert tr2.load_checkpoint() == 2 + 1 - 0 and tr2.global_step == ck["global_step"]
    assert list(tr2.model.engine.rng_state()) == ck["engine_rng_state"]
    assert torch.equal(tr2.model.engine.g_params, tr.model.engine.g_params)
    # cooperative stop
    (tmp_path / "stop.request").write_text("x")
    tr.start_epoch = 0
    assert tr._stop_requested()


def test_generation_cli_and_loaders(tmp_path):
    """checkpoint in -> PNGs out.  All three checkpoint layouts + a bare state_dict load;
    --seed makes the run reproducible; pixels follow the reference's uint8 truncation rule applied to
    the oracle's images for the same z."""
    from PIL import Image
    from signature_gan_amd import generate_signatures as cli
    from signature_gan_amd.utils.inference import infer_architecture_from_state_dict, load_generator, tensor_to_uint8
    size, latent = 64, 100
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in I.gen_state(O.g_state_specs(latent, size), SEED["state_g"]).items()}
    paths = {"A": tmp_path / "a.pt", "sd": tmp_path / "sd.pt", "bare": tmp_path / "bare.pt"}
    torch.save({"epoch": 1, "generator_state_dict": sd, "config": {"latent_dim": latent, "image_size": size}}, paths["A"])
    torch.save({"state_dict": sd}, paths["sd"])
    torch.save(sd, paths["bare"])
    assert infer_architecture_from_state_dict(sd) == (latent, size)
    sd128 = I.gen_state(O.g_state_specs(128, 128), 1)
    assert infer_architecture_from_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd128.items()}) == (128, 128)
    dev = torch.device("cuda:0")
    z = torch.from_numpy(I.gen_z(5, latent, 9))
    want = O.to_uint8(O.g_forward({k: v.clone() for k, v in sd.items()}, z, False, size))[:, 0].numpy()
    for name, p in paths.items():
        g, ld = load_generator(str(p), dev)
        assert ld == latent and not g.training
        got = tensor_to_uint8(g(z.cuda()))
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1, name      # truncation boundary only
        assert (got != want).mean() < 1e-3
    out1, out2 = tmp_path / "o1", tmp_path / "o2"
    cli.main(["--checkpoint", str(paths["A"]), "--n_samples", "7", "--batch_size", "4", "--seed", "3", "--output_dir", str(out1)])
    cli.main(["--checkpoint", str(paths["A"]), "--n_samples", "7", "--batch_size", "4", "--seed", "3", "--output_dir", str(out2)])
    names = sorted(os.listdir(out1))
    assert names == [f"signature_{i:06d}.png" for i in range(1, 8)]
    for n in names:
        a, b = np.asarray(Image.open(out1 / n)), np.asarray(Image.open(out2 / n))
        assert a.shape == (size, size) and a.dtype == np.uint8 and np.array_equal(a, b)
    cli.main(["--checkpoint", str(paths["A"]), "--info"])


@pytest.mark.parametrize("size", [64, 128])
def test_spectral_norm_discriminator_scores_like_the_reference(size):
    """Inference with a spectral-norm checkpoint: eval-mode probabilities and features against the reference's
    (golden_spectral_norm.npz), re-normalisation when u / v change, and the power iteration of a train()-mode forward."""
    import json
    import os
    from common import GOLDEN, I, O, SEED
    from signature_gan_amd.discriminator_vanilla_gan import Discriminator
    f = np.load(os.path.join(GOLDEN, "golden_spectral_norm.npz"))
    state = I.gen_sn_state(O.d_state_specs(size), SEED["state_d"])
    d = Discriminator(input_size=size, use_spectral_norm=True).to("cuda").eval()
    d.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    x = torch.from_numpy(I.gen_real(4, size, SEED["real"])).cuda()
    p = d(x).cpu().reshape(-1).numpy()
    feat = d.forward_features(x).cpu().reshape(-1).numpy()
    assert np.abs(p - f[f"s{size}/probs"]).max() <= 2e-4 * np.abs(f[f"s{size}/probs"]).max()
    want = f[f"s{size}/feat_probe"]
    got = feat[I.probe_idx(feat.size, "feat", 256)]
    assert np.abs(got - want).max() <= 2e-4 * np.abs(want).max()
    sd = d.state_dict()
    assert [k for k in sd] == [k for k, _ in json.loads(str(f[f"s{size}/keys"]))]
    for k, v in state.items():
        assert torch.equal(sd[k].cpu(), torch.from_numpy(v)), k
    with torch.no_grad():                                       # u scaled by 2 -> sigma doubles -> the first block's output halves ...
        d.conv_blocks._modules["0"].block._modules["0"].weight_u.mul_(2.0)
    feat2 = d.forward_features(x).cpu().reshape(-1).numpy()      # ... so the features move: the weights were re-normalised
    assert np.abs(feat2 - feat).max() > 1e-2 * np.abs(feat).max()
    d.train()                                                  # a train()-mode forward runs torch's power iteration: u, v move
    v_before = d.state_dict()["classifier.0.weight_v"].clone()
    p_train = d(x)
    assert p_train.shape == (4, 1) and not torch.equal(v_before, d.state_dict()["classifier.0.weight_v"])
    for k in ("classifier.0.weight_v", "conv_blocks.1.block.0.weight_u"):
        assert abs(float(d.state_dict()[k].norm()) - 1.0) < 1e-4
