#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into HBM bytes per launch and kernel.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/pmc_fetch -- python3 bench.py --serialize --no-cpu --no-roofline --steps 5 --warmup 2
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/pmc_write -- python3 bench.py --serialize --no-cpu --no-roofline --steps 5 --warmup 2
    python3 profiles/make_pmc_traffic.py OUT/pmc_fetch OUT/pmc_write STEPS COMMIT > profiles/r02_pmc_traffic.json
(STEPS = warm-up + timed steps of the profiled command, for the per-step totals; add --dtype bf16 etc. to both passes for the
other workloads: profiles/r02_pmc_traffic_<dtype>_s<size>_b<batch>.json)

Counter units are KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (MI355X_MICROARCH.md, HBM
section), so read bytes = 2 * FETCH_SIZE * 1024.  Cross-check built in: k_adam reads 16 B and writes 12 B per parameter."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    """kernel-slot name as bench.py prints it: k_gconv<BM,BN> / k_wgrad<BM,BN> (fp32 and 16-bit operand kernels alike), other
    kernels by their bare name.  rocprofv3 leaves some template instantiations mangled in the counter CSV."""
    if name.startswith("_Z"):
        m = re.search(r"(\d+)(k_\w+)", name)
        base = m.group(2)[:int(m.group(1))] if m else name
        ints = re.findall(r"Li(\d+)E", name)
        n = base + ("<" + ",".join(ints) + ">" if ints else "")
    else:
        n = re.sub(r"^void ", "", name).replace("siggan::", "")
        n = re.sub(r"\(.*$", "", n).replace(" ", "")
        n = re.sub(r"<(float|__bf16|_Float16|__hip_bfloat16)>", "", n)
        n = re.sub(r"<(float|__bf16|_Float16),", "<", n)
    m = re.match(r"(k_gconv|k_wgrad)(?:16)?<(\d+),(\d+)", n)
    if m:
        return f"{m.group(1)}<{m.group(2)},{m.group(3)}>"
    return re.sub(r"<.*$", "", n) if n.startswith("k_") and not n.startswith(("k_gconv_up4", "k_bn_")) else n


def collect(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = agg[short(r["Kernel_Name"])]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
out = {"how": __doc__.split("\n\n")[1].strip() + "  " + __doc__.split("\n\n")[2].strip().replace("\n", " "),
       "commit": sys.argv[4] if len(sys.argv) > 4 else None, "profiled_steps": steps, "per_launch_bytes": {}}
for k in sorted(set(fetch) | set(write)):
    n = max(fetch.get(k, [0])[0], write.get(k, [0])[0]) or 1
    rd = 2.0 * fetch.get(k, [0, 0.0])[1] * 1024 / n
    wr = write.get(k, [0, 0.0])[1] * 1024 / n
    out["per_launch_bytes"][k] = {"launches": n, "read": round(rd), "write": round(wr), "total": round(rd + wr)}
if steps:
    tot = sum(v["total"] * v["launches"] for k, v in out["per_launch_bytes"].items() if k.startswith("k_"))
    out["per_step_bytes_all_library_kernels"] = round(tot / steps)
json.dump(out, sys.stdout, indent=1)
print()
