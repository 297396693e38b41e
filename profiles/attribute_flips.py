#!/usr/bin/env python3
"""Which gradient tensor carries the deviation a handful of differing activation-sign decisions produce?  CPU only.

    python profiles/attribute_flips.py [profiles/r03_parity_margins.json] > profiles/r03_flip_attribution.txt

For a G-step case of the margins file (the decisions of the HIP path that differ from the REFERENCE run's, each listed there by
activation layer and flat index) the oracle is run with the reference's decisions (fixture census) and with exactly those
elements flipped -- all of them, then one at a time -- and the per-parameter change of the gradient norm is printed.  This is
oracle(HIP signs) - oracle(reference signs), the quantity tests/test_engine_gpu.py::_reference_chain holds HIP - reference to."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch  # noqa: E402
from common import I, O, SEED, census_signs, load_golden, oracle_states  # noqa: E402

m = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_parity_margins.json")))
for case, (size, latent, batch) in (("s128_b32", (128, 128, 32)), ("s64_b64", (64, 100, 64)), ("s64_b128", (64, 100, 128))):
    row = m[f"{case}/warm/g"]
    f, _ = load_golden(size, batch)
    z2 = torch.from_numpy(I.gen_z(batch, latent, SEED["z"] + 1))
    ref_signs = census_signs(f, "gstep")

    def run(signs):
        g_sd, d_sd, _, _ = oracle_states(size, latent, warm=True)
        return O.g_grads(dict(g_sd), d_sd, z2, size, signs=signs)[1]

    def flipped(flips):
        out = [(i.clone(), p.clone()) for i, p in ref_signs]
        for fl in flips:
            idx, pos = out[fl["layer"]]
            k = int((idx == fl["index"]).nonzero())
            pos[k] = ~pos[k]
        return out
    layer = ["Generator fc"] + [f"Generator block {i}" for i in range(len(O.G_CHAIN[size]) - 1)] + \
            [f"Discriminator block {i}" for i in range(len(O.D_CHAIN[size]))]
    gref = run(ref_signs)
    print(f"== {case} G step (warm): {len(row['flips'])} decision(s) differ from the reference run; HIP vs reference measured on the MI355X: "
          f"gradient norm {row['grad_norm_vs_reference']:.2e}, probes {row['grad_probe_vs_reference']:.2e}")
    for fl in row["flips"]:
        print(f"   {layer[fl['layer']]:24s} element {fl['index']:9d}   reference pre-activation = {fl['reference_value_over_layer_max']:+.1e} x layer max")
    ghip = run(flipped(row["flips"]))
    print("   per parameter, all decisions flipped (gradient-norm change; largest element change / tensor max):")
    for k in gref:
        a, b = float(ghip[k].norm()), float(gref[k].norm())
        d = abs(a - b) / b if b > 1e-12 else 0.0
        if d > 1e-4:
            print(f"      {k:38s} {d:.2e}   {float((ghip[k] - gref[k]).abs().max() / gref[k].abs().max()):.2e}"
                  + ("   (true gradient zero: rounding noise only)" if k == "fc.0.bias" else ""))
    live = [k for k in gref if k != "fc.0.bias"]
    for fl in row["flips"]:
        g1 = run(flipped([fl]))
        w = max(live, key=lambda n: abs(float(g1[n].norm()) - float(gref[n].norm())) / max(float(gref[n].norm()), 1e-12))
        print(f"   one decision alone, {layer[fl['layer']]}[{fl['index']}]: worst live parameter {w} "
              f"{abs(float(g1[w].norm()) - float(gref[w].norm())) / float(gref[w].norm()):.2e}")
