#!/usr/bin/env python3
"""Print the DESIGN.md section-5 table from the tracked bench lines (profiles/r04_bench_*.json), so the document quotes the files."""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
rows = [("**configs[1]** 64×64 z = 100 batch 64 fp32 (the bench line)", "default"),
        ("configs[2] shard: same, **bf16**", "bf16_s64_b64"),
        ("configs[3] batch 128 fp32", "f32_s64_b128"),
        ("configs[4] shard: 128×128 z = 128 batch 32 **fp16**", "f16_s128_b32"),
        ("128×128 z = 128 batch 32 fp32", "f32_s128_b32"),
        ("configs[1] with the library communicator (`--dist`, world 1, RCCL all-reduce of both arenas every step)", "f32_dist_world1"),
        ("batch 256 fp32", "f32_s64_b256"),
        ("bf16 batch 128", "bf16_s64_b128"), ("bf16 batch 256", "bf16_s64_b256"),
        ("fp16 128×128 batch 64", "f16_s128_b64"),
        ("configs[0] fully-connected extension (**parity unpinned**)", "mlp_s28_b32")]
print("| workload | `value` | ms/step (median; p10–p90) | whole-step fraction of MFMA peak | dominant kernel | file |")
print("|---|---|---|---|---|---|")
for label, tag in rows:
    d = json.load(open(os.path.join(HERE, f"r04_bench_{tag}.json")))
    t, r = d.get("timing", {}), d.get("roofline")
    kern = "—" if not r else f"`{r['kernel']}` {r['achieved']:.1f} TFLOP/s = {r['frac']:.3f}, {r['avg_launch_us']:.1f} µs × {r['launches_per_step']:.0f}/step"
    print(f"| {label} | {d['value'] / 1e3:.1f} k images/s | {d['ms_per_step']:.3f} ({t.get('ms_per_step_p10', 0):.3f}–{t.get('ms_per_step_p90', 0):.3f}) | "
          f"{d['frac_of_mfma_peak_whole_step']:.3f} | {kern} | `r04_bench_{tag}.json` |")
