#!/usr/bin/env python3
"""Secondary figures ("generation throughput reported as a secondary figure"): tracked so that every
number the record quotes is one command away.

    python profiles/secondary.py [--out profiles/r04_secondary.json] [--no-trainer]

  * generation -- Generator.forward in eval mode (siggan_g_forward; reference: utils/inference.py:136-194,
    vanilla_gan_model.py:338-371) at several batch sizes / both image sizes: us per batch, images/s, achieved TFLOP/s on
    87.06 / 414.19 MFLOP per image, and the per-kernel breakdown of the batch-64 case (MFMA launches stamped with
    hipExtLaunchKernelGGL events, one kernel at a time)
  * trainer -- GANTrainer.train end to end (reference: train_vanilla_gan_signatures.py:486-635) on a folder of synthetic PNGs:
    decode cache + device loader with augmentation + pipelined step with the next batch staged + metrics read one step late +
    tqdm / logs; images/s over whole epochs
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G_FLOP = {64: 87.06e6, 128: 414.19e6}


def generation(torch, Engine):
    rows = []
    for B, S, Z in ((64, 64, 100), (256, 64, 100), (1024, 64, 100), (64, 128, 128), (256, 128, 128)):
        eng = Engine(latent_dim=Z, image_size=S, max_batch=B, device="cuda:0", seed=1)
        eng.init_reference(0)
        z = torch.randn(B, Z, device="cuda:0")
        for _ in range(10):
            eng.g_forward(z, training=False)
        torch.cuda.synchronize(); t0 = time.perf_counter(); n = 200
        for _ in range(n):
            eng.g_forward(z, training=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        row = {"image_size": S, "latent": Z, "batch": B, "us_per_batch": round(dt * 1e6, 1), "images_per_s": round(B / dt, 0),
               "tflops": round(G_FLOP[S] * B / dt / 1e12, 1)}
        if B == 64:
            eng.set_mode(graph=False, overlap=False)
            eng.prof_enable(True)
            for _ in range(50):
                eng.g_forward(z, training=False)
            recs = eng.prof_read()
            eng.prof_enable(False)
            row["mfma_kernels_per_batch"] = {r["name"]: {"launches": r["launches"] / 50, "us": round(1e3 * r["ms"] / 50, 2),
                                                         "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1)} for r in recs}
            row["mfma_us_per_batch"] = round(sum(1e3 * r["ms"] for r in recs) / 50, 1)
        rows.append(row)
        eng.close()
    return rows


def trainer(torch):
    import numpy as np
    from PIL import Image
    from signature_gan_amd.data_loader_signatures import create_data_loader
    from signature_gan_amd.train_vanilla_gan_signatures import GANTrainer, TrainingConfig
    d, run = tempfile.mkdtemp(), tempfile.mkdtemp()
    rng, n, epochs = np.random.default_rng(0), 6400, 10
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (72, 96), dtype=np.uint8), "L").save(os.path.join(d, f"s{i:05d}.png"))
    cfg = TrainingConfig(data_dir=d, epochs=epochs, batch_size=64, checkpoint_dir=run + "/ck", sample_dir=run + "/s", log_dir=run + "/l",
                         checkpoint_interval=100, sample_interval=100)
    tr = GANTrainer(cfg, device="cuda")
    t0 = time.perf_counter()
    loader = create_data_loader(d, batch_size=64, num_workers=4, image_size=64, device="cuda")
    t_load = time.perf_counter() - t0
    t0 = time.perf_counter()
    tr.train(data_loader=loader)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"images": n, "epochs": epochs, "steps_per_epoch": n // 64, "loader_build_s": round(t_load, 2), "train_s": round(dt, 2),
            "images_per_s_end_to_end": round(epochs * (n // 64) * 64 / dt, 0),
            "includes": "decode cache + device loader with augmentation + pipelined step (next batch staged) + metrics read one step late + tqdm / logs"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-trainer", action="store_true")
    a = ap.parse_args()
    import torch
    import signature_gan_amd  # noqa: F401
    from signature_gan_amd.engine import Engine
    out = {"what": "secondary figures, 1 x MI355X, fp32", "command": "python profiles/secondary.py", "generation": generation(torch, Engine)}
    if not a.no_trainer:
        out["trainer"] = trainer(torch)
    text = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
