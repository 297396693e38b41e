"""A/B of library builds inside ONE GPU-box call:  python scratch/ab_step.py [--dtype f32] [--reps 3] name=path.so ...
Each rep runs every build in turn (fresh process per build+rep: the library is loaded once per process) -- the default bench
loop (pipelined step, next batch staged), 60 warm-up + 300 timed steps."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time, torch
sys.path.insert(0, %r)
import signature_gan_amd
from signature_gan_amd.engine import Engine
dtype, size, latent, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
eng = Engine(latent_dim=latent, image_size=size, max_batch=B, device="cuda:0", seed=2, dtype=dtype)
eng.init_reference(0)
import os
real = (torch.rand(B, 1, size, size, device="cuda") * 2 - 1)
for _ in range(60): eng.train_step(real, sync=False, next_real=real)
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): eng.train_step(real, sync=False, next_real=real)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 100)
ts.sort(); print("%%.4f" %% (1e3 * ts[len(ts) // 2]))
z = torch.randn(B, latent, device="cuda")
for _ in range(20): eng.g_forward(z, training=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): eng.g_forward(z, training=False)
torch.cuda.synchronize(); print("%%.2f" %% (1e6 * (time.perf_counter() - t0) / 200))
''' % ROOT
args = sys.argv[1:]
dtype, reps, size, latent, batch = "f32", 3, 64, 100, 64
builds = []
i = 0
while i < len(args):
    if args[i] == "--dtype": dtype = args[i + 1]; i += 2
    elif args[i] == "--reps": reps = int(args[i + 1]); i += 2
    elif args[i] == "--size": size = int(args[i + 1]); i += 2
    elif args[i] == "--latent": latent = int(args[i + 1]); i += 2
    elif args[i] == "--batch": batch = int(args[i + 1]); i += 2
    else: builds.append(args[i].split("=", 1)); i += 1
res = {n: [] for n, _ in builds}
gen = {n: [] for n, _ in builds}
for r in range(reps):
    for n, p in builds:
        env = dict(os.environ, SIGGAN_LIB_PATH=os.path.abspath(p.split("@")[0]))
        if p.endswith("@defer0"): env["SIGGAN_AB_DEFER"] = "0"
        out = subprocess.run([sys.executable, "-c", CHILD, dtype, str(size), str(latent), str(batch)], env=env, capture_output=True, text=True, timeout=300)
        if out.returncode != 0:
            print(n, "FAILED", out.stderr[-600:], flush=True); continue
        a, b = out.stdout.split()[-2:]
        res[n].append(float(a)); gen[n].append(float(b))
        print(f"rep {r} {n:24s} {dtype} step {a} ms   gen {b} us", flush=True)
print(json.dumps({"dtype": dtype, "size": size, "batch": batch, "step_ms": {n: sorted(v) for n, v in res.items()}, "gen_us": {n: sorted(v) for n, v in gen.items()}}))
